// Implicit-GEMM convolution on MFMA for the CNN encoder/decoder (NHWC, no im2col buffer in HBM).
//
// One kernel covers every convolution of the v3 generator:
//   head 5x5 (v3:123-127, common.py:11-14), ResBlock / tail 3x3 (common.py:41-62, v3:351), tail 7x7 behind
//   ReflectionPad2d(3) (v3:354-355), stride-2 3x3 down-convs (v3:292-298), 1x1 fuse convs over a channel
//   concat of two or three maps (v3:255-284, 329-338, crs_gd4:327-330: the cat is never materialised, each source is a "tap"), and
//   ConvTranspose2d(k4,s2,p1) (v3:301-322) as four output-parity phases of 2x2 taps.
// The epilogue applies a per-channel affine (conv bias and ActNorm2d y=(x+b)*exp(w) folded together,
// models/actnorm.py:39-42), ReLU/tanh, up to two residual adds, and writes NHWC T or NCHW fp32.
//
// GEMM view: out^T[n][pixel] = sum_k W[n][k] * patch[pixel][k], k = tap*Cin + c.  Weights are the MFMA A
// operand, gathered input pixels the B operand (one 16-byte channel vector per lane), so a lane ends up
// with 4 consecutive output channels of one pixel (vector NHWC store) and, for NCHW, 16 consecutive
// pixels per channel across lanes.
#include "cfen_common.hpp"
#include "cfen_conv.hpp"
#include "cfen_internal.hpp"

namespace {

// bytes per LDS weight row: = 32 (mod 64), the pitch k_mlp2 measured conflict-free for 16-byte fragment reads (row r16, piece h)
template <typename T>
__host__ __device__ inline int conv_wl_pitch(int Kpad) {
  const int row = Kpad * (int)sizeof(T), m = row & 63;
  return row + (m <= 32 ? 32 - m : 96 - m);
}

// WL: the phase's whole weight matrix is staged in LDS once per workgroup (NW waves) and the A fragments come from there.  Without it
// every wave streams all Cout_pad x Kpad weights through the vector cache for its TM x 16 pixels: 83 KB per wave in ds_conv_e03 -- more
// than the L1 holds, so 2048 waves pull 176 MB out of L2 for a convolution whose maps are 19 MB (ds_conv_e03 36 -> 23 us, the gather-conv
// class 0.98 -> 0.93 ms).  Tried on top and dropped: K split over the four waves of a workgroup (the chain of dependent round trips is NOT
// what bounds these kernels: ds_conv_e03 36.7 -> 35.3 us, lgcat_conv_d03 44.6 -> 52.2) and, with the weights in LDS, the pixel vectors
// fetched branch-free two chunks ahead (small maps +-0, lgcat_conv_d01 99.7 -> 117.5 us).
// UP (with WL): source 1 is GViT's low-resolution map (ConvDesc::up4).  The workgroup's NW * TM * 16 pixels are a run inside one row, or whole rows
// of one 4-row group (host-checked), so their x4 bilinear values come from 3 low-resolution rows x (run / 4 + 2) columns: those are staged (edges
// clamped, as k_upsample4 clamps its indices), every (pixel, channel vector) is interpolated like k_upsample4 does it -- horizontal pass per
// neighbourhood row, then the vertical one, fp32, rounded once to T (the compiler contracts the two kernels' sums differently: last-bit differences) -- into an LDS tile, and the K loop takes source 1 from that tile: the
// full-resolution copy of the GViT output (written by k_upsample4, read back here: 2 x 0.18 GB per forward) and six launches are gone.
template <typename T>
__host__ __device__ inline int conv_up_pitch(int Cg) { return ((Cg / Mma<T>::EPL) | 1) * 16; }   // bytes per pixel of the tile: an odd number of 16-byte slots (conflict-free column reads)

template <typename T, int TN, int TM, bool WL = false, int NW = 4, bool UP = false>
__global__ __launch_bounds__(NW * 64) void k_conv(Grouped<ConvDesc> dg) {
  const ConvDesc& dref = dg.g[blockIdx.z];
  const ConvK d = conv_k(dref, blockIdx.y);
  constexpr int KC = Mma<T>::KC, EPL = Mma<T>::EPL;
  typedef typename Mma<T>::frag frag;
  __shared__ int taps_l[CFEN_MAX_TAPS];
  extern __shared__ __attribute__((aligned(16))) unsigned char wl[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, h = lane >> 4;
  const int phase = blockIdx.y;
  if (tid < d.ntaps) {
    ConvTap t = dref.taps[phase * d.ntaps + tid];
    taps_l[tid] = ((int)(unsigned char)t.src << 16) | ((int)(unsigned char)t.dy << 8) | (int)(unsigned char)t.dx;
  }
  const int wpitch = conv_wl_pitch<T>(d.Kpad);
  if (WL) {
    const int vpr = d.Kpad * (int)sizeof(T) / 16;   // 16-byte pieces per weight row
    const unsigned char* wsrc = (const unsigned char*)d.weight + (size_t)phase * d.Cout_pad * d.Kpad * sizeof(T);
    for (int i = tid; i < d.Cout_pad * vpr; i += NW * 64) {
      const int row = i / vpr, pc = i - row * vpr;
      *reinterpret_cast<uint4*>(wl + row * wpitch + pc * 16) = *reinterpret_cast<const uint4*>(wsrc + (size_t)i * 16);
    }
  }
  const long long total = (long long)d.B * d.Hb * d.Wb;
  const unsigned char* upt = nullptr;
  int up_pitch = 0;
  // UP: the source-0 vectors of the lane's pixels are requested BEFORE the interpolation prologue (its two barriers and LDS passes would otherwise sit
  // in front of the first global load of the K loop: lgcat_conv_d01 72 -> 87 us measured with the loads behind it).  A 1x1 over two maps: the lane's
  // k slot of chunk kc is channel c of source t = 0 / 1 (t >= 2: zero padding), no halo, no taps table.
  constexpr int UP_MAXCH = 6;
  frag upre[UP ? UP_MAXCH : 1][TM];
  if constexpr (UP) {
    const long long qw = ((long long)xcd_chunked_block(blockIdx.x, gridDim.x) * NW + wave) * (TM * 16);
    int t0 = (h * EPL) / d.Cin, c0 = h * EPL - t0 * d.Cin;
#pragma unroll
    for (int kc = 0; kc < UP_MAXCH; ++kc) {
      if (kc * KC < d.Kpad) {
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          long long q = qw + j * 16 + r16;
          if (q >= total) q = total - 1;
          upre[kc][j] = t0 == 0 ? load_frag<T>((const T*)d.src0 + (size_t)q * d.cs_in + c0) : Mma<T>::zero();
        }
        c0 += KC;
        while (c0 >= d.Cin) { c0 -= d.Cin; ++t0; }
      }
    }
  }
  if constexpr (UP) {
    static_assert(WL, "the upsampling variant stages its weights");
    constexpr int P = NW * TM * 16;
    const int cvg = d.Cin / EPL;
    up_pitch = conv_up_pitch<T>(d.Cin);
    const int Pw = P < d.Wb ? P : d.Wb, lowW = Pw / 4 + 2;
    unsigned char* lowt = wl + (d.Cout_pad * wpitch + 15) / 16 * 16;
    unsigned char* upw = lowt + 3 * lowW * cvg * 16;
    upt = upw;
    const long long q0 = (long long)xcd_chunked_block(blockIdx.x, gridDim.x) * P;
    if (q0 < total) {       // (total is a multiple of P: a workgroup is whole or absent)
      const int x0 = (int)(q0 % d.Wb), y0 = (int)((q0 / d.Wb) % d.Hb), b = (int)(q0 / ((long long)d.Wb * d.Hb));
      const int kx0 = x0 >> 2, ky = y0 >> 2;
      const T* low = (const T*)d.src1 + (size_t)b * d.up_h * d.up_w * d.up_cs;
      for (int i = tid; i < 3 * lowW * cvg; i += NW * 64) {
        const int v = i % cvg, j = (i / cvg) % lowW, a = i / (cvg * lowW);
        const int yy = min(max(ky - 1 + a, 0), d.up_h - 1), xx = min(max(kx0 - 1 + j, 0), d.up_w - 1);
        *reinterpret_cast<frag*>(lowt + (size_t)i * 16) = load_frag<T>(low + ((size_t)yy * d.up_w + xx) * d.up_cs + v * EPL);
      }
      __syncthreads();
      const float w0 = 0.375f, w1 = 0.1875f, w2 = 0.0625f;
      for (int i = tid; i < P * cvg; i += NW * 64) {
        const int v = i % cvg, pl = i / cvg;
        const int x = x0 + (P <= d.Wb ? pl : pl % d.Wb), y = y0 + (P <= d.Wb ? 0 : pl / d.Wb);
        const int rx = x & 3, ry = y & 3, jl = (x >> 2) - kx0;       // staged column 0 = low-resolution column kx0 - 1: the left neighbour of kx sits at jl
        const float wx[3] = {rx == 0 ? w0 : rx == 1 ? w1 : rx == 2 ? w2 : 0.f, rx == 0 || rx == 3 ? 0.625f : 0.75f, rx == 3 ? w0 : rx == 2 ? w1 : rx == 1 ? w2 : 0.f};
        const float wy[3] = {ry == 0 ? w0 : ry == 1 ? w1 : ry == 2 ? w2 : 0.f, ry == 0 || ry == 3 ? 0.625f : 0.75f, ry == 3 ? w0 : ry == 2 ? w1 : ry == 1 ? w2 : 0.f};
        float acc[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          float row[EPL];
#pragma unroll
          for (int e = 0; e < EPL; ++e) row[e] = 0.f;
#pragma unroll
          for (int bb = 0; bb < 3; ++bb) {
            const frag pv = *reinterpret_cast<const frag*>(lowt + ((size_t)(a * lowW + jl + bb) * cvg + v) * 16);
#pragma unroll
            for (int e = 0; e < EPL; ++e) row[e] += wx[bb] * (float)pv[e];
          }
          if (wy[a] != 0.f) {     // k_upsample4 leaves a zero-weight row out of the vertical sum
#pragma unroll
            for (int e = 0; e < EPL; ++e) acc[e] += wy[a] * row[e];
          }
        }
        frag o;
#pragma unroll
        for (int e = 0; e < EPL; ++e) o[e] = (T)acc[e];
        *reinterpret_cast<frag*>(upw + (size_t)pl * up_pitch + v * 16) = o;
      }
    }
  }
  __syncthreads();
  const long long q_wave = ((long long)xcd_chunked_block(blockIdx.x, gridDim.x) * NW + wave) * (TM * 16);
  if (q_wave >= total) return;

  int pb[TM], py[TM], px[TM];
  bool pv[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    long long q = q_wave + j * 16 + r16;
    pv[j] = q < total;
    if (!pv[j]) q = total - 1;
    px[j] = (int)(q % d.Wb);
    py[j] = (int)((q / d.Wb) % d.Hb);
    pb[j] = (int)(q / ((long long)d.Wb * d.Hb));
  }
  floatx4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

  const T* wbase = (const T*)d.weight + ((size_t)phase * d.Cout_pad + r16) * d.Kpad + h * EPL;
  int t = (h * EPL) / d.Cin;
  int c = h * EPL - t * d.Cin;
  const int nch = d.Kpad / KC;
  if constexpr (UP) {
#pragma unroll
    for (int kc = 0; kc < UP_MAXCH; ++kc) {
      if (kc < nch) {
        frag bf[TM];
#pragma unroll
        for (int j = 0; j < TM; ++j)
          bf[j] = t == 1 ? *reinterpret_cast<const frag*>(upt + (size_t)((wave * TM + j) * 16 + r16) * up_pitch + c * (int)sizeof(T)) : upre[kc][j];
        frag af[TN];
#pragma unroll
        for (int i = 0; i < TN; ++i) af[i] = *reinterpret_cast<const frag*>(wl + (i * 16 + r16) * wpitch + (kc * KC + h * EPL) * (int)sizeof(T));
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j) acc[i][j] = Mma<T>::mma(af[i], bf[j], acc[i][j]);
        c += KC;
        while (c >= d.Cin) { c -= d.Cin; ++t; }
      }
    }
  } else
  for (int kc = 0; kc < nch; ++kc) {
    frag bf[TM];
    if (t < d.ntaps) {
      const int tp = taps_l[t];
      const int dy = (int)(signed char)((tp >> 8) & 0xff), dx = (int)(signed char)(tp & 0xff);
      const int si = (tp >> 16) & 3;
      const T* sp = (const T*)(si == 0 ? d.src0 : si == 1 ? d.src1 : d.src2);
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        int iy = py[j] * d.in_stride + dy, ix = px[j] * d.in_stride + dx;
        bool ok = true;
        if (d.pad_reflect) {
          iy = iy < 0 ? -iy : (iy >= d.Hin ? 2 * d.Hin - 2 - iy : iy);
          ix = ix < 0 ? -ix : (ix >= d.Win ? 2 * d.Win - 2 - ix : ix);
        } else {
          ok = (iy >= 0) & (iy < d.Hin) & (ix >= 0) & (ix < d.Win);
        }
        bf[j] = ok ? load_frag<T>(sp + (((size_t)pb[j] * d.Hin + iy) * d.Win + ix) * d.cs_in + c) : Mma<T>::zero();
      }
    } else {
#pragma unroll
      for (int j = 0; j < TM; ++j) bf[j] = Mma<T>::zero();
    }
    frag af[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i)
      af[i] = WL ? *reinterpret_cast<const frag*>(wl + (i * 16 + r16) * wpitch + (kc * KC + h * EPL) * (int)sizeof(T))
                 : load_frag<T>(wbase + (size_t)i * 16 * d.Kpad + kc * KC);
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] = Mma<T>::mma(af[i], bf[j], acc[i][j]);
    c += KC;
    while (c >= d.Cin) {
      c -= d.Cin;
      ++t;
    }
  }

  // Epilogue in four sweeps -- affine vectors, residual vectors (unconditional loads at the lane's own or, past the end, the last pixel),
  // arithmetic, stores: written as one loop (load scale, load residual, add, store, next tile) every load was followed by its own
  // s_waitcnt vmcnt(0), which also waited for the store before it: 12 serial memory round trips per wave behind a K loop of 2-3.
  const int oy_off = d.oy_off, ox_off = d.ox_off;
  size_t opix[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) opix[j] = ((size_t)pb[j] * d.Hout + py[j] * d.out_stride + oy_off) * d.Wout + px[j] * d.out_stride + ox_off;
  if (d.out_nchw_f32) {   // (rare: the 7x7 tails run on k_conv7_tz) one tile at a time, nothing batched
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      if (!pv[j]) continue;
      const int oy = py[j] * d.out_stride + oy_off, ox = px[j] * d.out_stride + ox_off;
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int n = i * 16 + 4 * h;
        floatx4 v = acc[i][j] * *reinterpret_cast<const floatx4*>(d.scale + n) + *reinterpret_cast<const floatx4*>(d.shift + n);
        if (d.act == 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        } else if (d.act == 2) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = tanhf(v[r]);
        }
        float* o = (float*)d.out;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < d.Cout) o[(((size_t)pb[j] * d.Cout + n + r) * d.Hout + oy) * d.Wout + ox] = v[r];
      }
    }
    return;
  }
  typedef typename Mma<T>::out4 out4;
  constexpr int IG = TN <= 3 ? TN : TN % 2 == 0 ? 2 : 1;          // feature tiles per sweep group: bounds the live affine + residual registers
#pragma unroll
  for (int i0 = 0; i0 < TN; i0 += IG) {
    out4 r0[IG][TM], r1[IG][TM];
    floatx4 gsc[IG], gsf[IG];
#pragma unroll
    for (int i = 0; i < IG; ++i) {
      gsc[i] = *reinterpret_cast<const floatx4*>(d.scale + (i0 + i) * 16 + 4 * h);
      gsf[i] = *reinterpret_cast<const floatx4*>(d.shift + (i0 + i) * 16 + 4 * h);
    }
    int nres[IG];
#pragma unroll
    for (int i = 0; i < IG; ++i) nres[i] = (i0 + i) * 16 + 4 * h < d.cs_out ? (i0 + i) * 16 + 4 * h : 0;      // lanes past the channel stride re-read vector 0
    if (d.res0) {
#pragma unroll
      for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int i = 0; i < IG; ++i) r0[i][j] = *reinterpret_cast<const out4*>((const T*)d.res0 + opix[j] * d.cs_res + nres[i]);
    }
    if (d.res1) {
#pragma unroll
      for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int i = 0; i < IG; ++i) r1[i][j] = *reinterpret_cast<const out4*>((const T*)d.res1 + opix[j] * d.cs_res + nres[i]);
    }
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int i = 0; i < IG; ++i) {
        floatx4 v = acc[i0 + i][j] * gsc[i] + gsf[i];
        if (d.act == 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        } else if (d.act == 2) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = tanhf(v[r]);
        }
        if (d.res0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += (float)r0[i][j][r];
        }
        if (d.res1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += (float)r1[i][j][r];
        }
        acc[i0 + i][j] = v;
      }
    __builtin_amdgcn_sched_barrier(0);       // keep the next group's loads out of this one: its registers are what the grouping saves
  }
  if constexpr (sizeof(T) == 2 && TN >= 2) {
    if (d.cs_out % 8 == 0) {   // (wave-uniform) pairs of adjacent channel tiles leave as one 16-byte store per lane (cfen_common.hpp pair_tiles16): a 24-channel pixel in ONE instruction
#pragma unroll
      for (int j = 0; j < TM; ++j) {
#pragma unroll
        for (int i = 0; i + 1 < TN; i += 2) {
          const int f = i * 16 + 16 * (h & 1) + 8 * (h >> 1);
          const uint4 v = pair_tiles16(acc[i][j], acc[i + 1][j]);     // every lane takes part in the exchange
          if (pv[j] && f < d.cs_out) *reinterpret_cast<uint4*>((T*)d.out + opix[j] * d.cs_out + f) = v;
        }
        if constexpr (TN % 2 == 1) {
          if (pv[j] && (TN - 1) * 16 + 4 * h < d.cs_out) store4<T>((T*)d.out + opix[j] * d.cs_out + (TN - 1) * 16 + 4 * h, acc[TN - 1][j]);
        }
      }
      return;
    }
  }
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    if (!pv[j]) continue;
#pragma unroll
    for (int i = 0; i < TN; ++i)
      if (i * 16 + 4 * h < d.cs_out) store4<T>((T*)d.out + opix[j] * d.cs_out + i * 16 + 4 * h, acc[i][j]);
  }
}

template <typename T, int TN, int TM>
int launch_conv_t(int ng, const ConvDesc* dp, hipStream_t s) {
  const ConvDesc& d = dp[0];
  Grouped<ConvDesc> dg;
  for (int g = 0; g < CFEN_MAX_GROUPS; ++g) dg.g[g] = dp[g < ng ? g : 0];
  const long long total = (long long)d.B * d.Hb * d.Wb;
  const long long per_block = 4 * TM * 16;
  dim3 grid(cfen_grid8((total + per_block - 1) / per_block), d.nphase, ng);   // padded blocks exit on q_wave >= total
  CFEN_LAUNCH((k_conv<T, TN, TM>), grid, dim3(256), 0, s, dg);
  CFEN_CHECK_LAUNCH("conv");
  return CFEN_OK;
}

template <typename T, int NW, int TM>
bool conv_up4_geometry_ok(const ConvDesc& d) {
  constexpr int P = NW * TM * 16;
  const long long total = (long long)d.B * d.Hb * d.Wb;
  const bool run = d.Wb % P == 0, rows = P % d.Wb == 0 && P / d.Wb <= 4 && 4 % (P / d.Wb) == 0 && d.Hb % (P / d.Wb) == 0;
  return total % P == 0 && (run || rows) && d.Wb % 4 == 0 && d.Hb % 4 == 0 && d.up_h * 4 == d.Hb && d.up_w * 4 == d.Wb && d.Cin % Mma<T>::EPL == 0 &&
         d.up_cs % Mma<T>::EPL == 0 && d.Cin <= d.up_cs;
}

template <typename T, int TN, int TM, int NW, bool UP = false>
int launch_conv_wl_t(int ng, const ConvDesc* dp, hipStream_t s) {
  const ConvDesc& d = dp[0];
  Grouped<ConvDesc> dg;
  for (int g = 0; g < CFEN_MAX_GROUPS; ++g) dg.g[g] = dp[g < ng ? g : 0];
  const long long total = (long long)d.B * d.Hb * d.Wb;
  const long long per_block = (long long)NW * TM * 16;
  size_t smem = (size_t)d.Cout_pad * conv_wl_pitch<T>(d.Kpad);
  if (UP) {
    CFEN_CHECK_ARG(d.Kpad <= 6 * Mma<T>::KC, "conv (x4 source): at most %d input channels per map", 3 * Mma<T>::KC);
    CFEN_CHECK_ARG((conv_up4_geometry_ok<T, NW, TM>(d)), "conv (x4 source): %d x %d map / %d x %d low-resolution map do not tile into runs of %d pixels", d.Hb, d.Wb,
                   d.up_h, d.up_w, NW * TM * 16);
    const int P = NW * TM * 16, Pw = P < d.Wb ? P : d.Wb;
    smem = (smem + 15) / 16 * 16 + (size_t)3 * (Pw / 4 + 2) * (d.Cin / Mma<T>::EPL) * 16 + (size_t)P * conv_up_pitch<T>(d.Cin);
    CFEN_CHECK_ARG(smem <= 152 * 1024, "conv (x4 source): %zu bytes of LDS", smem);
  }
  static bool attr_set[64] = {};
  if (cfen_first_use_on_device(attr_set)) {
    if (hipFuncSetAttribute((const void*)k_conv<T, TN, TM, true, NW, UP>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess) {
      cfen_set_error("conv: cannot reserve LDS for the staged weights");
      return CFEN_ERR_HIP;
    }
  }
  dim3 grid(cfen_grid8((total + per_block - 1) / per_block), d.nphase, ng);
  CFEN_LAUNCH((k_conv<T, TN, TM, true, NW, UP>), grid, dim3(NW * 64), smem, s, dg);
  CFEN_CHECK_LAUNCH("conv");
  return CFEN_OK;
}

template <typename T>
int check_conv(const ConvDesc& d) {
  constexpr int KC = Mma<T>::KC, EPL = Mma<T>::EPL;
  CFEN_CHECK_ARG(d.B > 0 && d.Hb > 0 && d.Wb > 0, "conv: empty problem");
  CFEN_CHECK_ARG(d.Wb % 16 == 0, "conv: output width (%d) must be a multiple of 16", d.Wb);
  CFEN_CHECK_ARG(d.Cin % EPL == 0 && d.cs_in % EPL == 0 && d.Cin <= d.cs_in, "conv: Cin=%d cs_in=%d must be multiples of %d", d.Cin,
                 d.cs_in, EPL);
  CFEN_CHECK_ARG(d.ntaps > 0 && d.nphase > 0 && d.ntaps * d.nphase <= CFEN_MAX_TAPS, "conv: too many taps");
  CFEN_CHECK_ARG(d.Kpad % KC == 0 && d.Kpad >= d.ntaps * d.Cin, "conv: Kpad=%d must be a multiple of %d and >= taps*Cin", d.Kpad, KC);
  CFEN_CHECK_ARG(d.Cout_pad % 16 == 0 && d.Cout_pad >= d.Cout && d.Cout > 0, "conv: bad Cout/Cout_pad");
  CFEN_CHECK_ARG(d.out_nchw_f32 || (d.cs_out % 4 == 0 && d.cs_out <= d.Cout_pad), "conv: cs_out=%d must be a multiple of 4 and <= Cout_pad",
                 d.cs_out);
  CFEN_CHECK_ARG(!(d.out_nchw_f32 && (d.res[0] || d.res[1])), "conv: residuals unsupported with NCHW output");
  CFEN_CHECK_ARG(cfen_aligned16(d.src[0]) && cfen_aligned16(d.src[1]) && cfen_aligned16(d.src[2]) && cfen_aligned16(d.weight) && cfen_aligned16(d.out) &&
                 cfen_aligned16(d.scale) && cfen_aligned16(d.shift) && cfen_aligned16(d.res[0]) && cfen_aligned16(d.res[1]),
                 "conv: pointers must be 16-byte aligned");
  CFEN_CHECK_ARG(d.src[0] && d.weight && d.out && d.scale && d.shift, "conv: null pointer");
  for (int i = 0; i < d.ntaps * d.nphase; ++i)
    CFEN_CHECK_ARG(d.taps[i].src >= 0 && d.taps[i].src < 3 && d.src[d.taps[i].src], "conv: tap %d reads source %d, which is missing", i, (int)d.taps[i].src);
  if (d.pad_reflect) {
    int maxd = 0;
    for (int i = 0; i < d.ntaps * d.nphase; ++i) {
      maxd = max(maxd, abs((int)d.taps[i].dy));
      maxd = max(maxd, abs((int)d.taps[i].dx));
    }
    CFEN_CHECK_ARG(maxd < d.Hin && maxd < d.Win, "conv: reflection pad larger than the image");
  }
  return CFEN_OK;
}

}  // namespace
int& cfen_tune_conv_wlds() {
  static int v = 2;
  return v;
}
int& cfen_tune_conv_wlds_maxlog() {   // staged weights only for launches below 2^this pixels
  static int v = 40;
  return v;
}
namespace {

template <typename T>
int launch_conv(int ng, const ConvDesc* dp, hipStream_t s) {
  CFEN_CHECK_ARG(ng >= 1 && ng <= CFEN_MAX_GROUPS && dp, "conv: 1..%d problems per launch", CFEN_MAX_GROUPS);
  for (int g = 0; g < ng; ++g) {
    int rc = check_conv<T>(dp[g]);
    if (rc) return rc;
    CFEN_CHECK_ARG(dp[g].B == dp[0].B && dp[g].Hb == dp[0].Hb && dp[g].Wb == dp[0].Wb && dp[g].Cout_pad == dp[0].Cout_pad &&
                   dp[g].Kpad == dp[0].Kpad && dp[g].nphase == dp[0].nphase && dp[g].ntaps == dp[0].ntaps,
                   "conv: grouped problems must have the same geometry");
  }
  const ConvDesc& d = dp[0];
  // Wave tile = (Cout_pad / 16) x TM MFMA tiles.  The gather loop is a chain of dependent global loads, so what hides its
  // latency is the number of waves in flight: shrink TM until the launch has a few thousand waves (tools/profile_launches.py).
  const long long px = (long long)ng * d.B * d.Hb * d.Wb * d.nphase;
  const int shrink = px >= (1 << 20) ? 0 : px >= (1 << 18) ? 1 : 2;   // halve TM once / twice for small maps
  // weights staged in LDS (conv.wlds: 0 never, 1 where the matrix fits 60 KB with 4-wave workgroups, 2 also up to 150 KB with 8 waves)
  const int wlm = cfen_tune_conv_wlds();
  const size_t wbytes = (size_t)d.Cout_pad * conv_wl_pitch<T>(d.Kpad);
  if (d.up4) {   // 1x1 over [LViT map ; x4 of GViT's low-resolution map]: the staged-weight shapes below, source 1 interpolated in LDS
    if constexpr (sizeof(T) == 2) {
      for (int g = 0; g < ng; ++g)
        CFEN_CHECK_ARG(dp[g].up4 && dp[g].ntaps == 2 && dp[g].nphase == 1 && dp[g].in_stride == 1 && dp[g].taps[0].src == 0 && dp[g].taps[1].src == 1 &&
                       dp[g].taps[0].dy == 0 && dp[g].taps[0].dx == 0 && dp[g].taps[1].dy == 0 && dp[g].taps[1].dx == 0 && dp[g].up_h == d.up_h &&
                       dp[g].up_w == d.up_w && dp[g].up_cs == d.up_cs && dp[g].Cin == d.Cin,
                       "conv (x4 source): needs a 1x1 convolution over two maps, the same geometry in every grouped problem");
      const bool fat = wbytes >= 16 * 1024;
      switch (d.Cout_pad / 16) {
        case 2: return fat ? launch_conv_wl_t<T, 2, 2, 8, true>(ng, dp, s) : shrink >= 2 ? launch_conv_wl_t<T, 2, 1, 4, true>(ng, dp, s) : launch_conv_wl_t<T, 2, 2, 4, true>(ng, dp, s);
        case 3: return fat ? launch_conv_wl_t<T, 3, 2, 8, true>(ng, dp, s) : shrink >= 2 ? launch_conv_wl_t<T, 3, 1, 4, true>(ng, dp, s) : launch_conv_wl_t<T, 3, 2, 4, true>(ng, dp, s);
        case 4: return launch_conv_wl_t<T, 4, 1, 8, true>(ng, dp, s);
        case 6: return launch_conv_wl_t<T, 6, 1, 8, true>(ng, dp, s);
        default: break;
      }
    }
    cfen_set_error("conv (x4 source): fp16 and 32 / 48 / 64 / 96 output channels only");
    return CFEN_ERR_ARG;
  }
  // workgroup shape (MI355X, batch 8, per-launch times): 96 / 64 output channels and weight matrices >= 16 KB run 8 waves per staged copy
  // (lgcat_conv_d03 41.7 -> 28.0 us, ds_conv_e02 24.6 -> 20.2), the small 1x1 matrices stay on 4-wave workgroups (lgcat_conv_d02 +4 us with 8)
  if (wlm > 0 && wbytes <= (wlm > 1 ? 150 : 60) * 1024 && px < (1ll << cfen_tune_conv_wlds_maxlog())) {
    const bool fat = wbytes >= 16 * 1024;
    switch (d.Cout_pad / 16) {
      case 2: return fat ? launch_conv_wl_t<T, 2, 2, 8>(ng, dp, s) : shrink >= 2 ? launch_conv_wl_t<T, 2, 1, 4>(ng, dp, s) : launch_conv_wl_t<T, 2, 2, 4>(ng, dp, s);
      case 3: return fat ? launch_conv_wl_t<T, 3, 2, 8>(ng, dp, s) : shrink >= 2 ? launch_conv_wl_t<T, 3, 1, 4>(ng, dp, s) : launch_conv_wl_t<T, 3, 2, 4>(ng, dp, s);
      case 4: return launch_conv_wl_t<T, 4, 1, 8>(ng, dp, s);
      case 6: return launch_conv_wl_t<T, 6, 1, 8>(ng, dp, s);
      default: break;
    }
  }
  switch (d.Cout_pad / 16) {
    case 1: return launch_conv_t<T, 1, 2>(ng, dp, s);
    case 2: return shrink >= 2 ? launch_conv_t<T, 2, 1>(ng, dp, s) : launch_conv_t<T, 2, 2>(ng, dp, s);
    case 3: return shrink >= 2 ? launch_conv_t<T, 3, 1>(ng, dp, s) : shrink == 1 ? launch_conv_t<T, 3, 2>(ng, dp, s) : launch_conv_t<T, 3, 4>(ng, dp, s);
    case 4: return shrink >= 1 ? launch_conv_t<T, 4, 1>(ng, dp, s) : launch_conv_t<T, 4, 2>(ng, dp, s);
    case 6: return shrink >= 1 ? launch_conv_t<T, 6, 1>(ng, dp, s) : launch_conv_t<T, 6, 2>(ng, dp, s);
    case 8: return launch_conv_t<T, 8, 1>(ng, dp, s);
    default:
      cfen_set_error("conv: Cout_pad=%d unsupported (16,32,48,64,96,128)", d.Cout_pad);
      return CFEN_ERR_ARG;
  }
}

// ---------------------------------------------------------------------------------------------
// Per-(image, channel) statistics of x0 (+x1 +x2): sum, sum of squares, max -- partial results per
// pixel chunk, finished by the consumer.  Serves InstanceNorm2d (v3:292-302) and CFSM2G's global
// average / max pools (v3:1503-1505).
constexpr int ST_CHUNKS = 64;

template <typename T>
__global__ __launch_bounds__(256) void k_chan_stats(const T* __restrict__ x0, const T* __restrict__ x1, const T* __restrict__ x2,
                                                    float* __restrict__ part, int HW, int C, int cs) {
  constexpr int EPL = Vec16<T>::N;
  __shared__ float red[3][256][EPL + 1];
  const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  const int cv = C / EPL;                       // channel vectors
  const int lanes_per_cv = 256 / cv;            // threads sharing one channel vector
  const int my_cv = tid % cv, my_p = tid / cv;
  const int per = (HW + ST_CHUNKS - 1) / ST_CHUNKS;
  const int p0 = chunk * per, p1 = min(HW, p0 + per);
  float s[EPL], q[EPL], m[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) { s[e] = 0.f; q[e] = 0.f; m[e] = -3.0e38f; }
  if (my_p < lanes_per_cv) {
    // four pixels per pass, their 4 (or 12) vector loads issued before the first add (clamped to the last pixel past the end of the chunk)
    typedef typename Mma<T>::frag frag;
    constexpr int U = 4;
    for (int pb = p0 + my_p; pb < p1; pb += U * lanes_per_cv) {
      frag f0[U], f1[U], f2[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int p = min(pb + u * lanes_per_cv, p1 - 1);
        f0[u] = load_frag<T>(x0 + ((size_t)b * HW + p) * cs + my_cv * EPL);
      }
      if (x1) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int p = min(pb + u * lanes_per_cv, p1 - 1);
          f1[u] = load_frag<T>(x1 + ((size_t)b * HW + p) * cs + my_cv * EPL);
          f2[u] = load_frag<T>(x2 + ((size_t)b * HW + p) * cs + my_cv * EPL);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (pb + u * lanes_per_cv >= p1) continue;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
          float v = (float)f0[u][e];
          if (x1) v = (v + (float)f1[u][e]) + (float)f2[u][e];
          s[e] += v; q[e] += v * v; m[e] = fmaxf(m[e], v);
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < EPL; ++e) { red[0][tid][e] = s[e]; red[1][tid][e] = q[e]; red[2][tid][e] = m[e]; }
  __syncthreads();
  if (tid < C) {
    const int vcol = tid / EPL, e = tid % EPL;
    float ss = 0.f, qq = 0.f, mm = -3.0e38f;
    for (int k = 0; k < lanes_per_cv; ++k) {
      int src = k * cv + vcol;
      ss += red[0][src][e]; qq += red[1][src][e]; mm = fmaxf(mm, red[2][src][e]);
    }
    float* o = part + (((size_t)b * ST_CHUNKS + chunk) * 3) * C;
    o[tid] = ss; o[C + tid] = qq; o[2 * C + tid] = mm;
  }
}

// InstanceNorm2d(affine=False, eps) + ReLU, in place (biased variance).
template <typename T>
__global__ __launch_bounds__(256) void k_instnorm_relu(T* __restrict__ x, const float* __restrict__ part, int HW, int C, int cs, float eps,
                                                       long long nvec_per_img) {
  constexpr int EPL = Vec16<T>::N;
  __shared__ float mean_l[256], rstd_l[256];
  __shared__ double ps_l[256], pq_l[256];
  const int b = blockIdx.y, tid = threadIdx.x;
  // every block finishes the statistics itself; the 64 partials of a channel are split over 256 / C threads (each a fixed subset, combined in
  // a fixed order: deterministic) -- one thread per channel walked them as a chain of 64 dependent round trips, ~8 us in front of the pass
  const int tpc = max(1, 256 / C), sub = tid / C, cc = tid % C;
  if (sub < tpc) {
    double s = 0.0, q = 0.0;
#pragma unroll 4
    for (int k = sub; k < ST_CHUNKS; k += tpc) {
      const float* o = part + (((size_t)b * ST_CHUNKS + k) * 3) * C;
      s += (double)o[cc]; q += (double)o[C + cc];
    }
    ps_l[tid] = s; pq_l[tid] = q;
  }
  __syncthreads();
  if (tid < C) {
    double s = 0.0, q = 0.0;
    for (int k = 0; k < tpc; ++k) { s += ps_l[k * C + tid]; q += pq_l[k * C + tid]; }
    double mean = s / HW;
    double var = q / HW - mean * mean;
    if (var < 0.0) var = 0.0;
    mean_l[tid] = (float)mean;
    rstd_l[tid] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  const int cv = C / EPL;
  // four vectors per thread and pass, loads first: in place the compiler must keep a load behind the store before it, so the plain
  // grid-stride loop was one memory round trip per vector
  constexpr int U = 4;
  const long long stride = (long long)gridDim.x * 256;
  for (long long idx0 = (long long)blockIdx.x * 256 + tid; idx0 < nvec_per_img; idx0 += U * stride) {
    typename Mma<T>::frag q[U];
    T* ptr[U];
    int vv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long idx = idx0 + u * stride;
      const bool ok = idx < nvec_per_img;
      const long long id = ok ? idx : idx0;
      vv[u] = (int)(id % cv);
      ptr[u] = ok ? x + ((size_t)b * HW + id / cv) * cs + vv[u] * EPL : nullptr;
      q[u] = load_frag<T>(x + ((size_t)b * HW + id / cv) * cs + vv[u] * EPL);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!ptr[u]) continue;
      float val[EPL];
#pragma unroll
      for (int e = 0; e < EPL; ++e) val[e] = fmaxf(((float)q[u][e] - mean_l[vv[u] * EPL + e]) * rstd_l[vv[u] * EPL + e], 0.f);
      Vec16<T>::store(ptr[u], val);
    }
  }
}

// CFSM2G (v3:1481-1517): gates from the pooled statistics (four C -> C/4 -> C MLPs, recomputed by
// every block: ~10 kFLOP), then out = x0 + x1*g1 + x2*g2.
template <typename T>
__global__ __launch_bounds__(256) void k_cfsm_apply(const T* __restrict__ x0, const T* __restrict__ x1, const T* __restrict__ x2,
                                                    T* __restrict__ out, const float* __restrict__ part, const float* __restrict__ w,
                                                    int HW, int C, int cs, long long nvec_per_img) {
  constexpr int EPL = Vec16<T>::N;
  __shared__ float avg_l[128], max_l[128], hid[4][32], g_l[2][128];
  __shared__ double ps_l[256];
  __shared__ float pm_l[256];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int bk = C / 4;
  const int tpc = max(1, 256 / C), sub = tid / C, cc = tid % C;      // partials of a channel split over 256 / C threads (see k_instnorm_relu)
  if (sub < tpc) {
    double s = 0.0;
    float m = -3.0e38f;
#pragma unroll 4
    for (int k = sub; k < ST_CHUNKS; k += tpc) {
      const float* o = part + (((size_t)b * ST_CHUNKS + k) * 3) * C;
      s += (double)o[cc];
      m = fmaxf(m, o[2 * C + cc]);
    }
    ps_l[tid] = s; pm_l[tid] = m;
  }
  __syncthreads();
  if (tid < C) {
    double s = 0.0;
    float m = -3.0e38f;
    for (int k = 0; k < tpc; ++k) { s += ps_l[k * C + tid]; m = fmaxf(m, pm_l[k * C + tid]); }
    avg_l[tid] = (float)(s / HW);
    max_l[tid] = m;
  }
  __syncthreads();
  // weights: [avg1, avg2, max1, max2] each {W0 (bk x C), W2 (C x bk)}
  const int per = 2 * bk * C;
  if (tid < 4 * bk) {
    int f = tid / bk, j = tid % bk;
    const float* W0 = w + f * per;
    const float* in = (f < 2) ? avg_l : max_l;
    float a = 0.f;
#pragma unroll 8
    for (int c = 0; c < C; ++c) a += W0[j * C + c] * in[c];
    hid[f][j] = fmaxf(a, 0.f);
  }
  __syncthreads();
  if (tid < 2 * C) {
    int gsel = tid / C, c = tid % C;
    const float* Wa = w + gsel * per + bk * C;          // fc_avg_cf{1,2}.2
    const float* Wm = w + (2 + gsel) * per + bk * C;    // fc_max_cf{1,2}.2
    float a = 0.f;
#pragma unroll 4
    for (int j = 0; j < bk; ++j) a += Wa[c * bk + j] * hid[gsel][j] + Wm[c * bk + j] * hid[2 + gsel][j];
    g_l[gsel][c] = 1.f / (1.f + expf(-a));
  }
  __syncthreads();
  const int cv = C / EPL;
  constexpr int U = 2;      // two pixels' vectors per thread and pass: six loads in flight
  const long long stride = (long long)gridDim.x * 256;
  for (long long idx0 = (long long)blockIdx.x * 256 + tid; idx0 < nvec_per_img; idx0 += U * stride) {
    typename Mma<T>::frag q0[U], q1[U], q2[U];
    size_t off[U];
    int vv[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long idx = idx0 + u * stride;
      ok[u] = idx < nvec_per_img;
      const long long id = ok[u] ? idx : idx0;
      vv[u] = (int)(id % cv);
      off[u] = ((size_t)b * HW + id / cv) * cs + vv[u] * EPL;
      q0[u] = load_frag<T>(x0 + off[u]); q1[u] = load_frag<T>(x1 + off[u]); q2[u] = load_frag<T>(x2 + off[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!ok[u]) continue;
      float a0[EPL];
#pragma unroll
      for (int e = 0; e < EPL; ++e) a0[e] = (float)q0[u][e] + (float)q1[u][e] * g_l[0][vv[u] * EPL + e] + (float)q2[u][e] * g_l[1][vv[u] * EPL + e];
      Vec16<T>::store(out + off[u], a0);
    }
  }
}

// ActNorm2d data-dependent first-call initialisation (models/actnorm.py:25-37) from the statistics of the raw layer output
// x = conv + conv_bias over the WHOLE batch: bias_an = -mean, weight_an = -0.5 log(max(var_unbiased, 0.2)); written as the folded
// epilogue table (scale = exp(weight_an), shift = (conv_bias + bias_an) * scale) and as the raw (weight_an, bias_an) pair.
__global__ __launch_bounds__(128) void k_actnorm_finalize(const float* __restrict__ part, int B, int HW, int C, int Cs, const float* __restrict__ conv_bias,
                                                          float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ an_out,
                                                          int Cpad) {
  const int c = threadIdx.x;
  if (c >= Cpad) return;
  if (c >= C) { scale[c] = 0.f; shift[c] = 0.f; an_out[c] = 0.f; an_out[Cpad + c] = 0.f; return; }
  double s = 0.0, q = 0.0;
  for (int b = 0; b < B; ++b)
    for (int k = 0; k < ST_CHUNKS; ++k) {
      const float* o = part + (((size_t)b * ST_CHUNKS + k) * 3) * Cs;   // statistics were taken over the map's Cs (padded) channels
      s += (double)o[c]; q += (double)o[Cs + c];
    }
  const double n = (double)B * HW;
  const double mean = s / n;
  double var = (q - s * mean) / (n - 1.0);
  if (var < 0.2) var = 0.2;
  const float w = (float)(-0.5 * log(var)), bb = (float)(-mean);
  const float sc = expf(w);
  scale[c] = sc;
  shift[c] = (conv_bias[c] + bb) * sc;
  an_out[c] = w;
  an_out[Cpad + c] = bb;
}

template <typename T>
int run_stats(const void* x0, const void* x1, const void* x2, float* part, int B, int HW, int C, int cs, hipStream_t s) {
  constexpr int EPL = Vec16<T>::N;
  CFEN_CHECK_ARG(B > 0 && HW > 0 && C > 0, "chan_stats: empty problem");
  CFEN_CHECK_ARG(C % EPL == 0 && cs % EPL == 0 && C <= cs && C <= 128, "chan_stats: C=%d cs=%d unsupported", C, cs);
  CFEN_CHECK_ARG((x1 == nullptr) == (x2 == nullptr), "chan_stats: x1 and x2 go together");
  CFEN_CHECK_ARG(cfen_aligned16(x0) && cfen_aligned16(x1) && cfen_aligned16(x2) && part, "chan_stats: bad pointers");
  CFEN_LAUNCH(k_chan_stats<T>, dim3(ST_CHUNKS, B), dim3(256), 0, s, (const T*)x0, (const T*)x1, (const T*)x2, part, HW, C, cs);
  CFEN_CHECK_LAUNCH("chan_stats");
  return CFEN_OK;
}

// blocks per image of the normalise / gate passes.  Every block first rebuilds the per-channel statistics (and, for CFSM2G, the gate
// MLPs) from the partial sums -- a few microseconds of serial work -- so a block must stream enough pixels to amortise it:
// ~8 vectors per thread, at most 128 blocks per image (1024 blocks at batch 8 = 4 per CU): norm class 237 -> 176 us per forward.
inline unsigned grid_img(long long nvec) {
  long long g = (nvec + 2047) / 2048;
  return (unsigned)(g < 1 ? 1 : (g > 128 ? 128 : g));
}

}  // namespace

size_t cfen_stats_workspace_bytes(int B, int C) { return (size_t)B * ST_CHUNKS * 3 * C * sizeof(float); }

int cfen_conv_impl_g(int dtype, int ng, const ConvDesc* d, hipStream_t s) {
  if (dtype == 1) return launch_conv<half_t>(ng, d, s);
  if (dtype == 0) return launch_conv<float>(ng, d, s);
  cfen_set_error("conv: unknown dtype %d", dtype);
  return CFEN_ERR_ARG;
}
int cfen_conv_impl(int dtype, const ConvDesc* d, hipStream_t s) { return cfen_conv_impl_g(dtype, 1, d, s); }

// can a grouped 1x1 fuse conv over [map ; x4 of a low-resolution map] run with ConvDesc::up4?  Mirrors launch_conv's choice of workgroup shape.
bool cfen_conv_up4_supported(int dtype, int ng, int B, int H, int W, int Cin, int cs_low, int Cout_pad, int Kpad) {
  if (dtype != 1 || H % 4 || W % 4) return false;
  ConvDesc d;
  memset(&d, 0, sizeof(d));
  d.B = B; d.Hb = H; d.Wb = W; d.up_h = H / 4; d.up_w = W / 4; d.up_cs = cs_low; d.Cin = Cin;
  const long long px = (long long)ng * B * H * W;
  const int shrink = px >= (1 << 20) ? 0 : px >= (1 << 18) ? 1 : 2;
  const bool fat = (size_t)Cout_pad * conv_wl_pitch<half_t>(Kpad) >= 16 * 1024;
  if (Kpad > 6 * 32) return false;
  switch (Cout_pad / 16) {
    case 2: case 3: return fat ? conv_up4_geometry_ok<half_t, 8, 2>(d) : shrink >= 2 ? conv_up4_geometry_ok<half_t, 4, 1>(d) : conv_up4_geometry_ok<half_t, 4, 2>(d);
    case 4: case 6: return conv_up4_geometry_ok<half_t, 8, 1>(d);
    default: return false;
  }
}

int cfen_instnorm_relu_impl(int dtype, void* x, float* part, int B, int HW, int C, int cs, float eps, hipStream_t s) {
  int rc = dtype == 1 ? run_stats<half_t>(x, nullptr, nullptr, part, B, HW, C, cs, s)
                      : dtype == 0 ? run_stats<float>(x, nullptr, nullptr, part, B, HW, C, cs, s) : CFEN_ERR_ARG;
  if (rc) {
    if (dtype != 0 && dtype != 1) cfen_set_error("instnorm: unknown dtype %d", dtype);
    return rc;
  }
  if (dtype == 1) {
    long long nvec = (long long)HW * (C / 8);
    CFEN_LAUNCH(k_instnorm_relu<half_t>, dim3(grid_img(nvec), B), dim3(256), 0, s, (half_t*)x, part, HW, C, cs, eps, nvec);
  } else {
    long long nvec = (long long)HW * (C / 4);
    CFEN_LAUNCH(k_instnorm_relu<float>, dim3(grid_img(nvec), B), dim3(256), 0, s, (float*)x, part, HW, C, cs, eps, nvec);
  }
  CFEN_CHECK_LAUNCH("instnorm");
  return CFEN_OK;
}

int cfen_actnorm_init_impl(int dtype, const void* x, float* part, int B, int HW, int C, int cs, int Cpad, const float* conv_bias, float* scale,
                           float* shift, float* an_out, hipStream_t s) {
  CFEN_CHECK_ARG(conv_bias && scale && shift && an_out && Cpad <= 128 && C <= Cpad && (long long)B * HW > 1, "actnorm_init: bad arguments");
  int rc = dtype == 1 ? run_stats<half_t>(x, nullptr, nullptr, part, B, HW, cs, cs, s)   // all cs channels: the padded ones hold zeros
                      : dtype == 0 ? run_stats<float>(x, nullptr, nullptr, part, B, HW, cs, cs, s) : CFEN_ERR_ARG;
  if (rc) return rc;
  CFEN_LAUNCH(k_actnorm_finalize, dim3(1), dim3(128), 0, s, (const float*)part, B, HW, C, cs, conv_bias, scale, shift, an_out, Cpad);
  CFEN_CHECK_LAUNCH("actnorm_init");
  return CFEN_OK;
}

int cfen_cfsm2g_impl(int dtype, const void* x0, const void* x1, const void* x2, void* out, const float* w, float* part, int B, int HW,
                     int C, int cs, hipStream_t s) {
  CFEN_CHECK_ARG(C % 4 == 0 && C <= 128 && C / 4 <= 32, "cfsm2g: C=%d unsupported", C);
  CFEN_CHECK_ARG(x1 && x2 && w && out, "cfsm2g: null pointer");
  int rc = dtype == 1 ? run_stats<half_t>(x0, x1, x2, part, B, HW, C, cs, s)
                      : dtype == 0 ? run_stats<float>(x0, x1, x2, part, B, HW, C, cs, s) : CFEN_ERR_ARG;
  if (rc) {
    if (dtype != 0 && dtype != 1) cfen_set_error("cfsm2g: unknown dtype %d", dtype);
    return rc;
  }
  CFEN_CHECK_ARG(cfen_aligned16(out), "cfsm2g: output must be 16-byte aligned");
  if (dtype == 1) {
    long long nvec = (long long)HW * (C / 8);
    CFEN_LAUNCH(k_cfsm_apply<half_t>, dim3(grid_img(nvec), B), dim3(256), 0, s, (const half_t*)x0, (const half_t*)x1,
                       (const half_t*)x2, (half_t*)out, part, w, HW, C, cs, nvec);
  } else {
    long long nvec = (long long)HW * (C / 4);
    CFEN_LAUNCH(k_cfsm_apply<float>, dim3(grid_img(nvec), B), dim3(256), 0, s, (const float*)x0, (const float*)x1,
                       (const float*)x2, (float*)out, part, w, HW, C, cs, nvec);
  }
  CFEN_CHECK_LAUNCH("cfsm2g");
  return CFEN_OK;
}
