// LDS-tiled stride-1 convolution for the full-resolution CNN layers (12 channels at 2N x 2N: head 5x5 + ResBlock 3x3,
// v3:123-127 / common.py:41-62; tail 3x3 + 7x7 behind ReflectionPad2d(3), v3:351-355).  These layers are HBM bound
// (2 * map bytes), but the gather kernel (k_conv.hip) spends its time on per-tap address arithmetic and L1/L2 re-reads.
//
// One workgroup = 64 x R output pixels; its (R+KS-1) x (64+KS-1) input halo is staged ONCE in LDS (zero or reflect
// padding resolved while staging).  A wave owns a 16-pixel-wide column strip of R rows:
//   * NHWC pixels are PIXB = 16/32/64 bytes, so the 64 bytes an MFMA chunk consumes per column are TPC = 64/PIXB
//     horizontally adjacent taps: the B fragment is one ds_read_b128 at  row*RB + (x + c*TPC)*PIXB + 16*h  -- no per-tap
//     address math, no select;
//   * the same input-row fragment serves every (output row r, dy = iy - r) pair, so it is read once and fed to up to KS
//     MFMAs whose weights (all KS*NCR fragments of the layer) sit in registers.
// Weight layout "rows": [Cout_pad=16][KS][NCR*TPC taps][CSI], i.e. the tap-major layout of a KS x (NCR*TPC) kernel whose
// extra columns are zero (packing.pack_conv_weight_rows).
#include "cfen_common.hpp"
#include "cfen_conv.hpp"
#include "cfen_internal.hpp"

namespace {

template <typename T, int PIXB, int KS, int R>
__global__ __launch_bounds__(256) void k_conv_tile(Grouped<ConvDesc> dg, int nblk) {
  const ConvDesc& d = dg.g[blockIdx.z];
  constexpr int SZ = (int)sizeof(T), EPL = Mma<T>::EPL, KC = Mma<T>::KC;
  constexpr int CSI = PIXB / SZ;              // channel stride of the input map
  constexpr int TPC = 64 / PIXB;              // taps per 64-byte chunk
  constexpr int NCR = (KS + TPC - 1) / TPC;   // chunks per kernel row
  constexpr int PAD = KS / 2;
  constexpr int ROWS = R + KS - 1;
  constexpr int WT = 64 + NCR * TPC;          // staged columns: 64 + KS - 1 real, the rest zero (they meet zero weights)
  constexpr int RB = WT * PIXB;
  constexpr int PPP = PIXB / 16;              // 16-byte pieces per pixel
  constexpr int NPIECE = ROWS * WT * PPP;
  constexpr int NIT = (NPIECE + 255) / 256;
  constexpr int KPAD = KS * NCR * KC;
  typedef typename Mma<T>::frag frag;
  static_assert(KC * SZ == 64, "one MFMA chunk is 64 bytes of K per column");
  __shared__ __attribute__((aligned(16))) unsigned char lds[ROWS * RB];

  const int blk = (int)xcd_chunked_block(blockIdx.x, gridDim.x);
  if (blk >= nblk) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, h = lane >> 4;
  const int tiles_x = d.Win / 64, tiles_y = d.Hin / R;
  const int tx = blk % tiles_x, ty = (blk / tiles_x) % tiles_y, b = blk / (tiles_x * tiles_y);
  const int x0 = tx * 64, y0 = ty * R;
  const unsigned char* src = (const unsigned char*)d.src[0] + (size_t)b * d.Hin * d.Win * PIXB;

  // ---- stage the halo tile (all loads in flight before the first LDS store) ----
  frag stg[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int idx = tid + i * 256;
    const int piece = idx % PPP, col = (idx / PPP) % WT, row = idx / (PPP * WT);
    int gy = y0 - PAD + row, gx = x0 - PAD + col;
    bool ok = idx < NPIECE && col < 64 + KS - 1;
    if (d.pad_reflect) {
      gy = gy < 0 ? -gy : (gy >= d.Hin ? 2 * d.Hin - 2 - gy : gy);
      gx = gx < 0 ? -gx : (gx >= d.Win ? 2 * d.Win - 2 - gx : gx);
    } else {
      ok = ok && gy >= 0 && gy < d.Hin && gx >= 0 && gx < d.Win;
    }
    stg[i] = ok ? *reinterpret_cast<const frag*>(src + ((size_t)gy * d.Win + gx) * PIXB + piece * 16) : Mma<T>::zero();
  }
  // ---- the layer's weights: lane (r16 = output feature, h) keeps its 16 bytes of every chunk ----
  frag wf[KS][NCR];
  {
    const T* wp = (const T*)d.weight + (size_t)r16 * KPAD + h * EPL;
#pragma unroll
    for (int dy = 0; dy < KS; ++dy)
#pragma unroll
      for (int c = 0; c < NCR; ++c) wf[dy][c] = load_frag<T>(wp + (dy * NCR + c) * KC);
  }
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int idx = tid + i * 256;
    if (idx < NPIECE) *reinterpret_cast<frag*>(&lds[idx * 16]) = stg[i];   // piece order == LDS order
  }
  __syncthreads();

  floatx4 acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = floatx4{0.f, 0.f, 0.f, 0.f};
  const unsigned char* lp = lds + (wave * 16 + r16) * PIXB + h * 16;
#pragma unroll
  for (int iy = 0; iy < ROWS; ++iy) {
#pragma unroll
    for (int c = 0; c < NCR; ++c) {
      const frag bf = *reinterpret_cast<const frag*>(lp + iy * RB + c * TPC * PIXB);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int dy = iy - r;
        if (dy >= 0 && dy < KS) acc[r] = Mma<T>::mma(wf[dy][c], bf, acc[r]);
      }
    }
  }

  // ---- epilogue: lane owns output features 4h..4h+3 of pixel (y0 + r, x0 + 16*wave + r16) ----
  const int n = 4 * h, ox = x0 + wave * 16 + r16;
  const floatx4 sc = *reinterpret_cast<const floatx4*>(d.scale + n), sh = *reinterpret_cast<const floatx4*>(d.shift + n);
  // the descriptor fields of the row loop, read ONCE: through `d` each row re-read them from the argument block after its store (73 s_loads)
  const int act = d.act, nchw = d.out_nchw_f32, Cout = d.Cout, Hin = d.Hin, Win = d.Win, cs_out = d.cs_out, cs_res = d.cs_res;
  const void* const res0 = d.res[0];
  const void* const res1 = d.res[1];
  void* const outp = d.out;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int oy = y0 + r;
    floatx4 v = acc[r] * sc + sh;
    if (act == 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    } else if (act == 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = tanhf(v[e]);
    }
    if (nchw) {
      float* o = (float*)outp;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n + e < Cout) o[(((size_t)b * Cout + n + e) * Hin + oy) * Win + ox] = v[e];
    } else if (n < cs_out) {
      const size_t opix = ((size_t)b * Hin + oy) * Win + ox;
      if (res0) v += load4<T>((const T*)res0 + opix * cs_res + n);
      if (res1) v += load4<T>((const T*)res1 + opix * cs_res + n);
      store4<T>((T*)outp + opix * cs_out + n, v);
    }
  }
}

constexpr int TILE_R = 8;

template <typename T, int PIXB, int KS>
int launch_tile(int ng, const ConvDesc* dp, hipStream_t s) {
  const ConvDesc& d = dp[0];
  Grouped<ConvDesc> dg;
  for (int g = 0; g < CFEN_MAX_GROUPS; ++g) dg.g[g] = dp[g < ng ? g : 0];
  const long long nblk = (long long)d.B * (d.Hin / TILE_R) * (d.Win / 64);
  CFEN_LAUNCH((k_conv_tile<T, PIXB, KS, TILE_R>), dim3(cfen_grid8(nblk), 1, ng), dim3(256), 0, s, dg, (int)nblk);
  CFEN_CHECK_LAUNCH("conv (tile)");
  return CFEN_OK;
}

template <typename T>
int launch_conv_tile(int ng, const ConvDesc* dp, int k, hipStream_t s) {
  for (int g = 0; g < ng; ++g) {
    const ConvDesc& d = dp[g];
  CFEN_CHECK_ARG(cfen_aligned16(d.src[0]) && cfen_aligned16(d.weight) && cfen_aligned16(d.out) && cfen_aligned16(d.scale) &&
                 cfen_aligned16(d.shift) && cfen_aligned16(d.res[0]) && cfen_aligned16(d.res[1]), "conv: pointers must be 16-byte aligned");
  CFEN_CHECK_ARG(d.src[0] && d.weight && d.out && d.scale && d.shift, "conv: null pointer");
  CFEN_CHECK_ARG(!(d.out_nchw_f32 && (d.res[0] || d.res[1])), "conv: residuals unsupported with NCHW output");
  CFEN_CHECK_ARG(d.out_nchw_f32 || (d.cs_out % 4 == 0 && d.cs_out <= 16), "conv: cs_out=%d must be a multiple of 4 and <= 16", d.cs_out);
    CFEN_CHECK_ARG(!d.pad_reflect || (k / 2 < d.Hin && k / 2 < d.Win), "conv: reflection pad larger than the image");
  }
  const int pixb = dp[0].cs_in * (int)sizeof(T);
  if constexpr (sizeof(T) == 2) {
    if (pixb == 16 && k == 5) return launch_tile<T, 16, 5>(ng, dp, s);
    if (pixb == 32 && k == 3) return launch_tile<T, 32, 3>(ng, dp, s);
    if (pixb == 32 && k == 7) return launch_tile<T, 32, 7>(ng, dp, s);
  } else {
    if (pixb == 32 && k == 5) return launch_tile<T, 32, 5>(ng, dp, s);
    if (pixb == 64 && k == 3) return launch_tile<T, 64, 3>(ng, dp, s);
    if (pixb == 64 && k == 7) return launch_tile<T, 64, 7>(ng, dp, s);
  }
  cfen_set_error("conv (rows layout): kernel %dx%d over %d-byte pixels has no tiled instantiation", k, k, pixb);
  return CFEN_ERR_ARG;
}

}  // namespace

// geometry the tiled kernel covers (mirrored by packing.conv_uses_rows_layout)
bool cfen_conv_tile_supported(int dtype, int kind, int k, int stride, int pad, int nsrc, int cs_in, int Cout_pad, int H, int W) {
  const int pixb = cs_in * (dtype == 1 ? 2 : 4);
  if (kind != 0 || stride != 1 || nsrc != 1 || pad != k / 2 || Cout_pad != 16 || H % TILE_R || W % 64) return false;
  if (dtype == 1) return (pixb == 16 && k == 5) || (pixb == 32 && (k == 3 || k == 7));
  return (pixb == 32 && k == 5) || (pixb == 64 && (k == 3 || k == 7));
}

int cfen_conv_tile_kpad(int dtype, int k, int cs_in) {
  const int pixb = cs_in * (dtype == 1 ? 2 : 4), tpc = 64 / pixb, ncr = (k + tpc - 1) / tpc;
  return k * ncr * (dtype == 1 ? 32 : 16);
}

int cfen_conv_tile_impl(int dtype, const ConvDesc* d, int k, hipStream_t s) { return cfen_conv_tile_impl_g(dtype, 1, d, k, s); }

int cfen_conv_tile_impl_g(int dtype, int ng, const ConvDesc* dp, int k, hipStream_t s) {
  CFEN_CHECK_ARG(ng >= 1 && ng <= CFEN_MAX_GROUPS && dp, "conv (rows layout): 1..%d problems per launch", CFEN_MAX_GROUPS);
  for (int g = 0; g < ng; ++g) {
    const ConvDesc* d = dp + g;
    CFEN_CHECK_ARG(d->B == dp->B && d->Hin == dp->Hin && d->Win == dp->Win && d->cs_in == dp->cs_in, "conv (rows layout): grouped problems must have the same geometry");
  CFEN_CHECK_ARG(cfen_conv_tile_supported(dtype, 0, k, d->in_stride, k / 2, 1, d->cs_in, d->Cout_pad, d->Hin, d->Win) && d->nphase == 1 &&
                 d->Hout == d->Hin && d->Wout == d->Win,
                 "conv (rows layout): unsupported geometry k=%d cs_in=%d Cout_pad=%d %dx%d", k, d->cs_in, d->Cout_pad, d->Hin, d->Win);
  CFEN_CHECK_ARG(d->Kpad == cfen_conv_tile_kpad(dtype, k, d->cs_in), "conv (rows layout): Kpad=%d, expected %d", d->Kpad,
                 cfen_conv_tile_kpad(dtype, k, d->cs_in));
  }
  if (dtype == 1) return launch_conv_tile<half_t>(ng, dp, k, s);
  return launch_conv_tile<float>(ng, dp, k, s);
}

// ---------------------------------------------------------------------------------------------------------------
// ConvTranspose2d(k4, s2, p1) (v3:301-322) on an LDS-staged input tile.  Output pixel (2y+py, 2x+px) gathers a 2x2
// input neighbourhood (cfen_conv.hpp: cfen_desc_convT4), so the four parity phases of one base tile read the SAME
// (RY+2) x (16*NX+2) input halo: it is staged once (channels zero-padded to a whole number of 64-byte chunks, CPT per
// pixel) and each of the 4 waves computes one phase of the whole tile with its phase's weights in registers.  An
// input-row fragment feeds both output rows that use it (taps ty = 0 and 1).  16-byte pieces are XOR-swizzled inside a
// pixel so that the 16 lanes of a ds_read_b128 group hit distinct bank positions.
// Weight layout "rows": [4 phases][Cout_pad][4 taps][PIXB / sizeof(T)] (packing.pack_convT_weight_rows).
namespace {

// A ds_read_b128 is served in four groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS) --
// i.e. k-quarter h of pixels r16 in {0-3, 12-15} together with k-quarter h ^ 1 of pixels {4-11}; each group must land on 16 distinct 16-byte
// slots of the 256-byte bank row for every tap shift of the tile (column base 0, 1, 2).  Round 2's flips assumed groups of one k-quarter and
// were conflict-free at base 0 only (SQ_LDS_BANK_CONFLICT: 52 / 50 / 19 % of the LDS cycles of the 64- / 128- / 192-byte variants); these are
// conflict-free at all three bases (exhaustive check of the bank rule, tools/lds_swizzle_search.py).
template <int PIXB> CFEN_DEV int convt_swz(int col) {
  if (PIXB == 128) return ((col >> 1) & 3) << 1; // 8 pieces per pixel
  return ((col >> 2) & 1) << 1;                  // 64 / 192 bytes per pixel: flip inside each 64-byte chunk
}

template <typename T, int PIXB, int TN, int NX, int RY>
__global__ __launch_bounds__(256) void k_convT_tile(Grouped<ConvDesc> dg, int nblk, int tpw) {   // tpw consecutive tiles per workgroup (see k_conv7_tz)
  const ConvDesc& d = dg.g[blockIdx.z];
  constexpr int SZ = (int)sizeof(T), EPL = Mma<T>::EPL, KC = Mma<T>::KC;
  constexpr int CPT = PIXB / 64;                 // chunks per tap
  constexpr int TW = 16 * NX;                    // base pixels per tile row
  constexpr int WT = TW + 2, ROWS = RY + 2;
  constexpr int RB = WT * PIXB;
  constexpr int PPP = PIXB / 16;
  constexpr int NPIECE = ROWS * WT * PPP;
  constexpr int NIT = (NPIECE + 255) / 256;
  constexpr int KPAD = 4 * CPT * KC;
  typedef typename Mma<T>::frag frag;
  // the halo tile, then (behind a barrier) the output tile: 2 RY rows x 2 TW pixels of at most TN * 16 channels, from where the workgroup stores whole
  // rows in 16-byte pieces -- a wave's own results are every second pixel of a row (one parity phase), 8 bytes a lane: a quarter to a half of each
  // line per store instruction
  constexpr int OUTB = PIXB == 64 ? 2 * RY * 2 * TW * TN * 16 * SZ : 0;     // (the 64-byte-pixel variant only, see the epilogue)
  __shared__ __attribute__((aligned(16))) unsigned char lds[ROWS * RB > OUTB ? ROWS * RB : OUTB];

  constexpr bool MT = PIXB == 128;               // multi-tile only in the 128-byte variant (us_conv_d02: 54.7 -> 47.0 us with two tiles); the loop
                                                 // costs registers: the 192-byte variant drops to one wave per SIMD with it (44 -> 58 us), the
                                                 // 64-byte one has 4 weight fragments and nothing to amortise
  const int ntile = MT ? tpw : 1;
  const int blk0 = (int)xcd_chunked_block(blockIdx.x, gridDim.x) * ntile;
  if (blk0 >= nblk) return;
  const int tid = threadIdx.x, lane = tid & 63, phase = tid >> 6, r16 = lane & 15, h = lane >> 4;
  const int py = phase >> 1, px = phase & 1;
  // this wave's phase: weights of its 4 taps
  frag wf[4][CPT][TN];
  {
    const T* wp = (const T*)d.weight + ((size_t)phase * d.Cout_pad + r16) * KPAD + h * EPL;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int c = 0; c < CPT; ++c)
#pragma unroll
        for (int i = 0; i < TN; ++i) wf[t][c][i] = load_frag<T>(wp + (size_t)i * 16 * KPAD + (t * CPT + c) * KC);
  }
  for (int it = 0; it < ntile; ++it) {
  const int blk = blk0 + it;
  if (blk >= nblk) break;
  if (it) __syncthreads();                        // every wave is done reading the previous tile's halo
  const int tiles_x = d.Win / TW, tiles_y = d.Hin / RY;
  const int tx = blk % tiles_x, ty = (blk / tiles_x) % tiles_y, b = blk / (tiles_x * tiles_y);
  const int x0 = tx * TW, y0 = ty * RY;
  const int src_pixb = d.cs_in * SZ, src_pieces = src_pixb / 16;
  const unsigned char* src = (const unsigned char*)d.src[0] + (size_t)b * d.Hin * d.Win * src_pixb;

  frag stg[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int idx = tid + i * 256;
    const int piece = idx % PPP, col = (idx / PPP) % WT, row = idx / (PPP * WT);
    const int gy = y0 - 1 + row, gx = x0 - 1 + col;
    const bool ok = idx < NPIECE && piece < src_pieces && gy >= 0 && gy < d.Hin && gx >= 0 && gx < d.Win;
    stg[i] = ok ? *reinterpret_cast<const frag*>(src + ((size_t)gy * d.Win + gx) * src_pixb + piece * 16) : Mma<T>::zero();
  }
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int idx = tid + i * 256;
    const int piece = idx % PPP, col = (idx / PPP) % WT, row = idx / (PPP * WT);
    if (idx < NPIECE) *reinterpret_cast<frag*>(&lds[row * RB + col * PIXB + ((piece ^ convt_swz<PIXB>(col)) << 4)]) = stg[i];
  }
  __syncthreads();

  floatx4 acc[RY][NX][TN];
#pragma unroll
  for (int r = 0; r < RY; ++r)
#pragma unroll
    for (int xq = 0; xq < NX; ++xq)
#pragma unroll
      for (int i = 0; i < TN; ++i) acc[r][xq][i] = floatx4{0.f, 0.f, 0.f, 0.f};

  // Parity p reads input offsets {0, -1} (p = 0) or {+1, 0} (p = 1) for taps 0, 1 (cfen_conv.hpp), i.e. tap t of output
  // base row r reads halo row r - t + 1 + p.  Output rows r = q (tap 0) and r = q + 1 (tap 1) share halo row q + 1 + p:
  // loop over q with compile-time accumulator indices; the phase only moves the LDS address.
  const unsigned char* lp = lds + (1 + py) * RB;
#pragma unroll
  for (int q = -1; q < RY; ++q) {
#pragma unroll
    for (int txx = 0; txx < 2; ++txx) {
#pragma unroll
      for (int xq = 0; xq < NX; ++xq) {
        const int col = xq * 16 + r16 + px + 1 - txx;
        const unsigned char* pp = lp + q * RB + col * PIXB;
        const int sw = convt_swz<PIXB>(col);
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
          const frag bf = *reinterpret_cast<const frag*>(pp + (((c * 4 + h) ^ sw) << 4));
          if (q >= 0) {
#pragma unroll
            for (int i = 0; i < TN; ++i) acc[q][xq][i] = Mma<T>::mma(wf[txx][c][i], bf, acc[q][xq][i]);
          }
          if (q + 1 < RY) {
#pragma unroll
            for (int i = 0; i < TN; ++i) acc[q + 1][xq][i] = Mma<T>::mma(wf[2 + txx][c][i], bf, acc[q + 1][xq][i]);
          }
        }
      }
    }
  }

  const int Hout = 2 * d.Hin, Wout = 2 * d.Win;
  if constexpr (PIXB == 64) {
    const int opx = d.cs_out * SZ, orow = 2 * TW * opx;       // bytes of an output pixel / of a tile row
    __syncthreads();                                          // every wave is done reading the halo
  #pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int n = i * 16 + 4 * h;
      if (n >= d.cs_out) continue;
      const floatx4 sc = *reinterpret_cast<const floatx4*>(d.scale + n), sh = *reinterpret_cast<const floatx4*>(d.shift + n);
  #pragma unroll
      for (int r = 0; r < RY; ++r)
  #pragma unroll
        for (int xq = 0; xq < NX; ++xq) {
          floatx4 v = acc[r][xq][i] * sc + sh;
          if (d.act == 1) {
  #pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          store4<T>(reinterpret_cast<T*>(lds + (2 * r + py) * orow + (2 * (xq * 16 + r16) + px) * opx) + n, v);
        }
    }
    __syncthreads();
    {
      const int ppr = orow / 16;                              // 16-byte pieces per tile row
      unsigned char* gout = (unsigned char*)d.out + (((size_t)b * Hout + 2 * y0) * Wout + 2 * x0) * opx;
      for (int idx = tid; idx < 2 * RY * ppr; idx += 256) {
        const int row = idx / ppr, pc = idx - row * ppr;
        *reinterpret_cast<frag*>(gout + (size_t)row * Wout * opx + pc * 16) = *reinterpret_cast<const frag*>(lds + row * orow + pc * 16);
      }
    }
  } else {
    // (the 128- / 192-byte variants keep the direct stores: staged they measured 46 -> 47.6 and 43 -> 43.1 us, the 64-byte one 85.7 -> 79.6)
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int n = i * 16 + 4 * h;
      if (n >= d.cs_out) continue;
      const floatx4 sc = *reinterpret_cast<const floatx4*>(d.scale + n), sh = *reinterpret_cast<const floatx4*>(d.shift + n);
#pragma unroll
      for (int r = 0; r < RY; ++r)
#pragma unroll
        for (int xq = 0; xq < NX; ++xq) {
          const int oy = 2 * (y0 + r) + py, ox = 2 * (x0 + xq * 16 + r16) + px;
          floatx4 v = acc[r][xq][i] * sc + sh;
          if (d.act == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          store4<T>((T*)d.out + (((size_t)b * Hout + oy) * Wout + ox) * d.cs_out + n, v);
        }
    }
  }
  }   // tiles of this workgroup
}

template <typename T, int PIXB, int TN, int NX, int RY>
int launch_convT_tile(int ng, const ConvDesc* dp, hipStream_t s) {
  const ConvDesc& d = dp[0];
  Grouped<ConvDesc> dg;
  for (int g = 0; g < CFEN_MAX_GROUPS; ++g) dg.g[g] = dp[g < ng ? g : 0];
  const long long nblk = (long long)d.B * (d.Hin / RY) * (d.Win / (16 * NX));
  const int tpw = PIXB == 128 ? std::max(1, std::min(cfen_tune_convT_tpw(), (int)(nblk * ng / 1024))) : 1;
  CFEN_LAUNCH((k_convT_tile<T, PIXB, TN, NX, RY>), dim3(cfen_grid8((nblk + tpw - 1) / tpw), 1, ng), dim3(256), 0, s, dg, (int)nblk, tpw);
  CFEN_CHECK_LAUNCH("convT (tile)");
  return CFEN_OK;
}

constexpr int CT_RY = 4;

}  // namespace

static int convt_pixb(int dtype, int cin) { return (cin * (dtype == 1 ? 2 : 4) + 63) / 64 * 64; }

// geometry the tiled ConvTranspose kernel covers (mirrored by packing.convT_uses_rows_layout)
bool cfen_convT_tile_supported(int dtype, int cs_in, int Cout_pad, int Hin, int Win) {
  const int pixb = convt_pixb(dtype, cs_in);
  if (dtype != 1 || Hin % CT_RY || Win % 32) return false;       // fp32 keeps the gather kernel
  return (pixb == 64 && Cout_pad == 16) || (pixb == 128 && Cout_pad == 32) || (pixb == 192 && Cout_pad == 48);
}

int cfen_convT_tile_kpad(int dtype, int cs_in) { return 4 * convt_pixb(dtype, cs_in) / (dtype == 1 ? 2 : 4); }

int cfen_convT_tile_impl(int dtype, const ConvDesc* d, hipStream_t s) { return cfen_convT_tile_impl_g(dtype, 1, d, s); }

int cfen_convT_tile_impl_g(int dtype, int ng, const ConvDesc* dp, hipStream_t s) {
  CFEN_CHECK_ARG(ng >= 1 && ng <= CFEN_MAX_GROUPS && dp, "convT (rows layout): 1..%d problems per launch", CFEN_MAX_GROUPS);
  for (int g = 0; g < ng; ++g) {
    const ConvDesc* d = dp + g;
    CFEN_CHECK_ARG(d->B == dp->B && d->Hin == dp->Hin && d->Win == dp->Win && d->cs_in == dp->cs_in && d->Cout_pad == dp->Cout_pad,
                   "convT (rows layout): grouped problems must have the same geometry");
  CFEN_CHECK_ARG(d->nphase == 4 && cfen_convT_tile_supported(dtype, d->cs_in, d->Cout_pad, d->Hin, d->Win),
                 "convT (rows layout): unsupported geometry cs_in=%d Cout_pad=%d %dx%d dtype=%d", d->cs_in, d->Cout_pad, d->Hin, d->Win, dtype);
  CFEN_CHECK_ARG(d->Kpad == cfen_convT_tile_kpad(dtype, d->cs_in), "convT (rows layout): Kpad=%d, expected %d", d->Kpad,
                 cfen_convT_tile_kpad(dtype, d->cs_in));
  CFEN_CHECK_ARG(cfen_aligned16(d->src[0]) && cfen_aligned16(d->weight) && cfen_aligned16(d->out) && cfen_aligned16(d->scale) &&
                 cfen_aligned16(d->shift) && d->src[0] && d->weight && d->out && d->scale && d->shift, "convT: null or misaligned pointer");
  CFEN_CHECK_ARG(!d->res[0] && !d->res[1] && !d->out_nchw_f32 && d->cs_out % 4 == 0 && d->cs_out <= d->Cout_pad && d->act != 2,
                 "convT (rows layout): residuals / NCHW output / tanh unsupported");
  }
  const int pixb = convt_pixb(dtype, dp->cs_in);
  if (pixb == 64) return launch_convT_tile<half_t, 64, 1, 2, CT_RY>(ng, dp, s);
  if (pixb == 128) return launch_convT_tile<half_t, 128, 2, 2, CT_RY>(ng, dp, s);
  return launch_convT_tile<half_t, 192, 3, 1, CT_RY>(ng, dp, s);
}

// ---------------------------------------------------------------------------------------------------------------
// 7x7 convolution with <= 4 output channels (the tails, v3:354-355: 12 -> 3 / 1 channels behind ReflectionPad2d(3), tanh,
// fp32 NCHW output).  With 3 output channels a 16-row MFMA is 81 % padding; here the 16 rows are (channel co, pixel
// offset dxo) pairs, co < 4, dxo < 4: column j of the MFMA stands for the 4 adjacent output pixels 4j .. 4j+3 and the
// weight matrix is the Toeplitz expansion  Wz[(co, dxo)][dy][kx'][ci] = w[co][dy][kx' - dxo][ci]  (0 <= kx' - dxo < 7,
// kx' < 10 -> 5 chunks of 2 taps per input row).  3.2x fewer MFMAs per pixel than the per-pixel-column kernel above, and
// a lane ends up with 4 CONSECUTIVE pixels of one channel: 16-byte NCHW stores.
// Workgroup = 64 x 16 output pixels, wave w owns rows 4w .. 4w+3 across the full width; the 22 x 70 pixel halo is staged
// once, its 16-byte pieces XOR-swizzled (slot ^= group & 6, group = pixel / 4) so that every ds_read_b128 lane group
// is conflict-free for all five chunk positions (found by exhaustive search over linear swizzles).
namespace {

constexpr int Z_RW = 4, Z_ROWS = 4 * Z_RW + 6, Z_WT = 72, Z_RB = Z_WT * 32, Z_NCH = 5, Z_KPAD = 7 * Z_NCH * 32;

// A workgroup walks `tpw` consecutive tiles: the 35 weight fragments of a lane (35 KB per wave, more than the L1 holds, so every wave of
// every tile pulled them from L2: 0.86 GB per launch of the three tails) are loaded once per workgroup instead of once per tile.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_conv7_tz(Grouped<ConvDesc> dg, int nblk, int tpw) {
  const ConvDesc& d = dg.g[blockIdx.z];
  typedef half_t T;
  typedef Mma<T>::frag frag;
  constexpr int NPIECE = Z_ROWS * Z_WT * 2, NIT = (NPIECE + 255) / 256;
  __shared__ __attribute__((aligned(16))) unsigned char lds[Z_ROWS * Z_RB];

  const int blk0 = (int)xcd_chunked_block(blockIdx.x, gridDim.x) * tpw;
  if (blk0 >= nblk) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, h = lane >> 4;
  frag wf[7][Z_NCH];
  {
    const T* wp = (const T*)d.weight + (size_t)r16 * Z_KPAD + h * 8;
#pragma unroll
    for (int dy = 0; dy < 7; ++dy)
#pragma unroll
      for (int c = 0; c < Z_NCH; ++c) wf[dy][c] = load_frag<T>(wp + (dy * Z_NCH + c) * 32);
  }
  for (int it = 0; it < tpw; ++it) {
  const int blk = blk0 + it;
  if (blk >= nblk) break;
  if (it) __syncthreads();                        // every wave is done reading the previous tile's halo
  const int tiles_x = d.Win / 64, tiles_y = d.Hin / (4 * Z_RW);
  const int tx = blk % tiles_x, ty = (blk / tiles_x) % tiles_y, b = blk / (tiles_x * tiles_y);
  const int x0 = tx * 64, y0 = ty * 4 * Z_RW;
  const unsigned char* src = (const unsigned char*)d.src[0] + (size_t)b * d.Hin * d.Win * 32;

  frag stg[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int idx = tid + i * 256;
    const int q = idx & 1, P = (idx >> 1) % Z_WT, row = (idx >> 1) / Z_WT;
    int gy = y0 - 3 + row, gx = x0 - 3 + P;
    bool ok = idx < NPIECE && P < 70;
    if (d.pad_reflect) {
      gy = gy < 0 ? -gy : (gy >= d.Hin ? 2 * d.Hin - 2 - gy : gy);
      gx = gx < 0 ? -gx : (gx >= d.Win ? 2 * d.Win - 2 - gx : gx);
    } else {
      ok = ok && gy >= 0 && gy < d.Hin && gx >= 0 && gx < d.Win;
    }
    stg[i] = ok ? *reinterpret_cast<const frag*>(src + ((size_t)gy * d.Win + gx) * 32 + q * 16) : Mma<T>::zero();
  }
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int idx = tid + i * 256;
    const int q = idx & 1, P = (idx >> 1) % Z_WT, row = (idx >> 1) / Z_WT;
    const int slot = 2 * P + q, G = P >> 2;
    if (idx < NPIECE) *reinterpret_cast<frag*>(&lds[row * Z_RB + (((slot & ~7) | ((slot & 7) ^ (G & 6))) << 4)]) = stg[i];
  }
  __syncthreads();

  floatx4 acc[Z_RW];
#pragma unroll
  for (int r = 0; r < Z_RW; ++r) acc[r] = floatx4{0.f, 0.f, 0.f, 0.f};
  const unsigned char* lp = lds + wave * Z_RW * Z_RB;
#pragma unroll
  for (int iy = 0; iy < Z_RW + 6; ++iy) {
#pragma unroll
    for (int c = 0; c < Z_NCH; ++c) {
      const int G = r16 + (c >> 1);
      const frag bf = *reinterpret_cast<const frag*>(lp + iy * Z_RB + ((8 * G + ((4 * (c & 1) + h) ^ (G & 6))) << 4));
#pragma unroll
      for (int r = 0; r < Z_RW; ++r) {
        const int dy = iy - r;
        if (dy >= 0 && dy < 7) acc[r] = Mma<T>::mma(wf[dy][c], bf, acc[r]);
      }
    }
  }

  // lane (j = r16, h): channel co = h, output pixels x0 + 4j .. +3 of row y0 + 4*wave + r
  if (d.out_nchw_f32 == 2) {
    // uint8 HWC x 3 output = util.tensor2im of the fp32 result (util/util.py:12-24: (x + 1) / 2 * 255 in fp32, truncating cast, no clamp, a 1-channel
    // map tiled to 3; k_tensor2im_u8's arithmetic) written by the tail itself: the lanes h = 0, 1, 2 of a column hold R, G, B of the same 4 pixels,
    // lane h = 0 collects their bytes (two cross-lane reads) and stores the 12 bytes of its 4 pixels -- 16 lanes = 192 consecutive bytes of a row
    const float sc = d.scale[h], sh = d.shift[h];       // (h = 3 reads a padding entry of the [16] table; its bytes are never stored)
    unsigned char* o = (unsigned char*)d.out + (((size_t)b * d.Hin + y0 + wave * Z_RW) * d.Win + x0 + 4 * r16) * 3;
#pragma unroll
    for (int r = 0; r < Z_RW; ++r) {
      floatx4 v = acc[r] * sc + sh;
      if (d.act == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      } else if (d.act == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = tanhf(v[e]);
      }
      unsigned w = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) w |= (unsigned)(unsigned char)(int)((v[e] + 1.f) / 2.0f * 255.0f) << (8 * e);
      const unsigned wg = d.Cout >= 3 ? (unsigned)__shfl((int)w, r16 + 16, 64) : w;     // all 64 lanes take part in the exchange
      const unsigned wb = d.Cout >= 3 ? (unsigned)__shfl((int)w, r16 + 32, 64) : w;
      if (h == 0) {
        // bytes R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
        const unsigned o0 = (w & 0xffu) | ((wg & 0xffu) << 8) | ((wb & 0xffu) << 16) | ((w & 0xff00u) << 16);
        const unsigned o1 = ((wg >> 8) & 0xffu) | (((wb >> 8) & 0xffu) << 8) | (((w >> 16) & 0xffu) << 16) | (((wg >> 16) & 0xffu) << 24);
        const unsigned o2 = ((wb >> 16) & 0xffu) | (((w >> 24) & 0xffu) << 8) | (((wg >> 24) & 0xffu) << 16) | (((wb >> 24) & 0xffu) << 24);
        unsigned* op = reinterpret_cast<unsigned*>(o + (size_t)r * d.Win * 3);
        op[0] = o0; op[1] = o1; op[2] = o2;
      }
    }
  } else if (h < d.Cout) {
    const float sc = d.scale[h], sh = d.shift[h];
    float* o = (float*)d.out + (((size_t)b * d.Cout + h) * d.Hin + y0 + wave * Z_RW) * d.Win + x0 + 4 * r16;
#pragma unroll
    for (int r = 0; r < Z_RW; ++r) {
      floatx4 v = acc[r] * sc + sh;
      if (d.act == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      } else if (d.act == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = tanhf(v[e]);
      }
      *reinterpret_cast<floatx4*>(o + (size_t)r * d.Win) = v;
    }
  }
  }   // tiles of this workgroup
}

}  // namespace

// geometry of the Toeplitz 7x7 kernel (mirrored by packing.conv_uses_toeplitz7)
bool cfen_conv7_tz_supported(int dtype, int k, int stride, int pad, int nsrc, int cs_in, int Cout, int out_nchw_f32, int H, int W) {
  return dtype == 1 && k == 7 && stride == 1 && pad == 3 && nsrc == 1 && cs_in == 16 && Cout >= 1 && Cout <= 4 && out_nchw_f32 && (out_nchw_f32 != 2 || Cout == 1 || Cout == 3) &&
         H % (4 * Z_RW) == 0 &&
         W % 64 == 0;
}
int cfen_conv7_tz_kpad() { return Z_KPAD; }
int& cfen_tune_convT_tpw() {   // tiles per workgroup of the 128-byte k_convT_tile variant ("convT.tpw")
  static int v = 2;
  return v;
}
int& cfen_tune_conv7_tpw() {   // tiles per workgroup of k_conv7_tz ("conv7.tpw")
  static int v = 4;
  return v;
}

int cfen_conv7_tz_impl_g(int dtype, int ng, const ConvDesc* dp, hipStream_t s) {
  CFEN_CHECK_ARG(ng >= 1 && ng <= CFEN_MAX_GROUPS && dp, "conv7 (toeplitz): 1..%d problems per launch", CFEN_MAX_GROUPS);
  Grouped<ConvDesc> dg;
  for (int g = 0; g < ng; ++g) {
    const ConvDesc& d = dp[g];
    CFEN_CHECK_ARG(cfen_conv7_tz_supported(dtype, 7, d.in_stride, 3, 1, d.cs_in, d.Cout, d.out_nchw_f32, d.Hin, d.Win) && d.nphase == 1 &&
                   d.Hout == d.Hin && d.Wout == d.Win && d.Kpad == Z_KPAD, "conv7 (toeplitz): unsupported geometry or weight layout");
    CFEN_CHECK_ARG(d.B == dp[0].B && d.Hin == dp[0].Hin && d.Win == dp[0].Win, "conv7 (toeplitz): grouped problems must have the same geometry");
    CFEN_CHECK_ARG(d.src[0] && d.weight && d.out && d.scale && d.shift && cfen_aligned16(d.src[0]) && cfen_aligned16(d.weight) && cfen_aligned16(d.out),
                   "conv7 (toeplitz): null or misaligned pointer");
    CFEN_CHECK_ARG(!d.pad_reflect || (3 < d.Hin && 3 < d.Win), "conv7 (toeplitz): reflection pad larger than the image");
  }
  for (int g = 0; g < CFEN_MAX_GROUPS; ++g) dg.g[g] = dp[g < ng ? g : 0];
  const long long nblk = (long long)dp[0].B * (dp[0].Hin / (4 * Z_RW)) * (dp[0].Win / 64);
  const int tpw = std::max(1, std::min(cfen_tune_conv7_tpw(), (int)(nblk * ng / 1024)));      // at least ~2 workgroups per CU stay in the grid
  CFEN_LAUNCH(k_conv7_tz, dim3(cfen_grid8((nblk + tpw - 1) / tpw), 1, ng), dim3(256), 0, s, dg, (int)nblk, tpw);
  CFEN_CHECK_LAUNCH("conv7 (toeplitz)");
  return CFEN_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// head = conv5x5 (3 -> 12) + ResBlock (conv3x3, ReLU, conv3x3, + input) of the generator (v3:123-127, 395; common.py:41-62) in ONE launch
// (round 3).  The three stride-1 convolutions ran at 2.8 - 5.1 TB/s each, i.e. at the HBM rate of their own maps: 33 + 67 MB for the 5x5,
// 67 + 67 for the first 3x3, 3 x 67 for the second -- 400 MB for a block whose input and output are 100 MB.  Here a workgroup (5 waves)
// produces a 64 x 8 output tile from an 87 x 16 input halo: the 5x5 output t1 (80 x 12 pixels) and the first 3x3's output t2 (80 x 10)
// exist only as fp16 tiles in LDS (rounded exactly as the maps the unfused kernels wrote), positions outside the image hold the zero padding
// the next convolution expects.  Intermediate strips are computed 80 pixels wide so that every column the final 64 need -- zero-weight pad
// taps included -- is a finite, correct value.  Per-wave structure as k_conv_tile: a 16-pixel column strip, an input-row fragment feeds every
// (output row, dy) pair, all weight fragments in registers ("rows" layout of the three layers, unchanged).
namespace {

struct HeadFusedArgs {
  const void* in; void* out;                      // NHWC fp16: input 8 channels (16-byte pixels), output 16 channels (12 real)
  const void *w5, *wa, *wb;                       // rows layouts: 5x5 over 16-byte pixels, 3x3 over 32-byte pixels (twice)
  const float *s5, *t5, *sa, *ta, *sb, *tb;       // (scale, shift) epilogue tables of the three layers
  int B, H, W;
};

constexpr int HF_R = 8;                            // output rows per workgroup
constexpr int HF_WIN = 87, HF_RIN = HF_R + 8;      // input halo: columns x0 - 10 .. x0 + 76, rows y0 - 4 .. y0 + R + 3
constexpr int HF_WT = 83;                          // t1 / t2 tile columns: x0 - 9 .. x0 + 73 (computed: x0 - 8 .. x0 + 71)
constexpr int HF_R1 = HF_R + 4, HF_R2 = HF_R + 2;  // t1 rows y0 - 2 .., t2 rows y0 - 1 ..
constexpr int HF_LDS = HF_RIN * HF_WIN * 16 + (HF_R1 + HF_R2) * HF_WT * 32;
static_assert(2 * HF_LDS <= 160 * 1024, "two workgroups a CU");

__global__ __launch_bounds__(320) void k_head_fused(HeadFusedArgs a, int nblk) {
  typedef half_t T;
  typedef half8 frag;
  __shared__ __attribute__((aligned(16))) unsigned char lds[HF_LDS];
  unsigned char* const tin = lds;
  unsigned char* const t1 = lds + HF_RIN * HF_WIN * 16;
  unsigned char* const t2 = t1 + HF_R1 * HF_WT * 32;
  constexpr int RBI = HF_WIN * 16, RBT = HF_WT * 32;

  const int blk = (int)xcd_chunked_block(blockIdx.x, gridDim.x);
  if (blk >= nblk) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, h = lane >> 4;
  const int tiles_x = a.W / 64, tiles_y = a.H / HF_R;
  const int tx = blk % tiles_x, ty = (blk / tiles_x) % tiles_y, b = blk / (tiles_x * tiles_y);
  const int x0 = tx * 64, y0 = ty * HF_R;
  const unsigned char* src = (const unsigned char*)a.in + (size_t)b * a.H * a.W * 16;

  // ---- stage the input halo (zero outside the image), all loads in flight before the first LDS store ----
  constexpr int NPIECE = HF_RIN * HF_WIN, NIT = (NPIECE + 319) / 320;
  frag stg[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int idx = tid + i * 320;
    const int col = idx % HF_WIN, row = idx / HF_WIN;
    const int gy = y0 - 4 + row, gx = x0 - 10 + col;
    const bool ok = idx < NPIECE && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    stg[i] = ok ? *reinterpret_cast<const frag*>(src + ((size_t)gy * a.W + gx) * 16) : Mma<T>::zero();
  }
  // the three layers' weights: lane (r16 = output feature, h) keeps its 16 bytes of every chunk
  frag w5[5][2], wa[3][2], wb[3][2];
  {
    const T* p5 = (const T*)a.w5 + (size_t)r16 * (5 * 2 * 32) + h * 8;
    const T* pa = (const T*)a.wa + (size_t)r16 * (3 * 2 * 32) + h * 8;
    const T* pb = (const T*)a.wb + (size_t)r16 * (3 * 2 * 32) + h * 8;
#pragma unroll
    for (int dy = 0; dy < 5; ++dy)
#pragma unroll
      for (int c = 0; c < 2; ++c) w5[dy][c] = load_frag<T>(p5 + (dy * 2 + c) * 32);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        wa[dy][c] = load_frag<T>(pa + (dy * 2 + c) * 32);
        wb[dy][c] = load_frag<T>(pb + (dy * 2 + c) * 32);
      }
  }
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int idx = tid + i * 320;
    if (idx < NPIECE) *reinterpret_cast<frag*>(tin + idx * 16) = stg[i];
  }
  __syncthreads();

  const int n = 4 * h;                               // the lane's 4 output channels of pixel column r16 of its strip
  const int gcol = x0 - 8 + wave * 16 + r16;         // image column of that pixel in the 80-wide intermediate strips
  const bool colin = gcol >= 0 && gcol < a.W;
  // ---- stage A: t1 = conv5x5(input) + bias on rows y0 - 2 .. y0 + R + 1, columns x0 - 8 .. x0 + 71 ----
  {
    floatx4 acc[HF_R1];
#pragma unroll
    for (int r = 0; r < HF_R1; ++r) acc[r] = floatx4{0.f, 0.f, 0.f, 0.f};
    const unsigned char* lp = tin + (wave * 16 + r16) * 16 + h * 16;     // input column (gcol - 2) - (x0 - 10) = 16 wave + r16; chunk = 4 taps of 16 B
#pragma unroll
    for (int iy = 0; iy < HF_RIN; ++iy)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const frag bf = *reinterpret_cast<const frag*>(lp + iy * RBI + c * 64);
#pragma unroll
        for (int r = 0; r < HF_R1; ++r) {
          const int dy = iy - r;
          if (dy >= 0 && dy < 5) acc[r] = Mma<T>::mma(w5[dy][c], bf, acc[r]);
        }
      }
    const floatx4 sc = *reinterpret_cast<const floatx4*>(a.s5 + n), sh = *reinterpret_cast<const floatx4*>(a.t5 + n);
#pragma unroll
    for (int r = 0; r < HF_R1; ++r) {
      const int gy = y0 - 2 + r;
      floatx4 v = acc[r] * sc + sh;
      if (!(colin && gy >= 0 && gy < a.H)) v = floatx4{0.f, 0.f, 0.f, 0.f};   // zero padding of the next convolution
      const half4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
      *reinterpret_cast<half4*>(t1 + r * RBT + (wave * 16 + r16 + 1) * 32 + n * 2) = o;
    }
  }
  __syncthreads();
  // ---- stage B: t2 = relu(conv3x3(t1) + bias) on rows y0 - 1 .. y0 + R, same 80 columns ----
  {
    floatx4 acc[HF_R2];
#pragma unroll
    for (int r = 0; r < HF_R2; ++r) acc[r] = floatx4{0.f, 0.f, 0.f, 0.f};
    const unsigned char* lp = t1 + (wave * 16 + r16) * 32 + h * 16;      // t1 tile column (gcol - 1) - (x0 - 9) = 16 wave + r16; chunk = 2 taps of 32 B
#pragma unroll
    for (int iy = 0; iy < HF_R1; ++iy)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const frag bf = *reinterpret_cast<const frag*>(lp + iy * RBT + c * 64);
#pragma unroll
        for (int r = 0; r < HF_R2; ++r) {
          const int dy = iy - r;
          if (dy >= 0 && dy < 3) acc[r] = Mma<T>::mma(wa[dy][c], bf, acc[r]);
        }
      }
    const floatx4 sc = *reinterpret_cast<const floatx4*>(a.sa + n), sh = *reinterpret_cast<const floatx4*>(a.ta + n);
#pragma unroll
    for (int r = 0; r < HF_R2; ++r) {
      const int gy = y0 - 1 + r;
      floatx4 v = acc[r] * sc + sh;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      if (!(colin && gy >= 0 && gy < a.H)) v = floatx4{0.f, 0.f, 0.f, 0.f};
      const half4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
      *reinterpret_cast<half4*>(t2 + r * RBT + (wave * 16 + r16 + 1) * 32 + n * 2) = o;
    }
  }
  __syncthreads();
  // ---- stage C: out = conv3x3(t2) + bias + t1 on the 64 x R tile (waves 0..3) ----
  if (wave < 4) {
    floatx4 acc[HF_R];
#pragma unroll
    for (int r = 0; r < HF_R; ++r) acc[r] = floatx4{0.f, 0.f, 0.f, 0.f};
    const unsigned char* lp = t2 + (wave * 16 + r16 + 8) * 32 + h * 16;  // t2 tile column (x0 + 16 wave + r16 - 1) - (x0 - 9)
#pragma unroll
    for (int iy = 0; iy < HF_R2; ++iy)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const frag bf = *reinterpret_cast<const frag*>(lp + iy * RBT + c * 64);
#pragma unroll
        for (int r = 0; r < HF_R; ++r) {
          const int dy = iy - r;
          if (dy >= 0 && dy < 3) acc[r] = Mma<T>::mma(wb[dy][c], bf, acc[r]);
        }
      }
    const floatx4 sc = *reinterpret_cast<const floatx4*>(a.sb + n), sh = *reinterpret_cast<const floatx4*>(a.tb + n);
    const int ox = x0 + wave * 16 + r16;
#pragma unroll
    for (int r = 0; r < HF_R; ++r) {
      floatx4 v = acc[r] * sc + sh;
      const half4 res = *reinterpret_cast<const half4*>(t1 + (r + 2) * RBT + (wave * 16 + r16 + 9) * 32 + n * 2);   // t1 at (y0 + r, ox): the ResBlock's input
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += (float)res[e];
      store4<T>((T*)a.out + (((size_t)b * a.H + y0 + r) * a.W + ox) * 16 + n, v);
    }
  }
}

}  // namespace

bool cfen_head_fused_supported(int dtype, int cs_in, int C, int H, int W) { return dtype == 1 && cs_in == 8 && C == 12 && H % HF_R == 0 && W % 64 == 0; }

// in: (B, H, W, 8) fp16; out: (B, H, W, 16); weights / tables = the "<layer>.wr" / ".scale" / ".shift" entries of head.0.0, head.0.1.body.0, head.0.1.body.2
int cfen_head_fused_impl(int dtype, const void* in, void* out, const void* w5, const float* s5, const float* t5, const void* wa, const float* sa,
                         const float* ta, const void* wb, const float* sb, const float* tb, int B, int H, int W, hipStream_t s) {
  CFEN_CHECK_ARG(cfen_head_fused_supported(dtype, 8, 12, H, W) && B > 0, "head (fused): fp16, image edges multiples of 8 / 64 (got dtype %d, %d x %d)", dtype, H, W);
  CFEN_CHECK_ARG(in && out && w5 && wa && wb && s5 && t5 && sa && ta && sb && tb, "head (fused): null pointer");
  CFEN_CHECK_ARG(cfen_aligned16(in) && cfen_aligned16(out) && cfen_aligned16(w5) && cfen_aligned16(wa) && cfen_aligned16(wb) && cfen_aligned16(s5) &&
                 cfen_aligned16(t5) && cfen_aligned16(sa) && cfen_aligned16(ta) && cfen_aligned16(sb) && cfen_aligned16(tb), "head (fused): pointers must be 16-byte aligned");
  const HeadFusedArgs a{in, out, w5, wa, wb, s5, t5, sa, ta, sb, tb, B, H, W};
  const long long nblk = (long long)B * (H / HF_R) * (W / 64);
  CFEN_CHECK_ARG(nblk < (1ll << 31), "head (fused): grid too large");
  CFEN_LAUNCH(k_head_fused, dim3(cfen_grid8(nblk)), dim3(320), 0, s, a, (int)nblk);
  CFEN_CHECK_LAUNCH("head (fused)");
  return CFEN_OK;
}
