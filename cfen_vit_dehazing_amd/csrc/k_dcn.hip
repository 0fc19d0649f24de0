// Deformable convolution forward (DCNv1 and modulated DCNv2) as ONE gather + MFMA GEMM kernel.
//
// Replaces, for the forward direction, the reference's CUDA extension:
//   deformable_im2col_gpu_kernel            dcn/src/deform_conv_cuda_kernel.cu:189-242  (bilinear :83-114)
//   modulated_deformable_im2col_gpu_kernel  dcn/src/deform_conv_cuda_kernel.cu:569-632
//   deform_conv_forward_cuda                dcn/src/deform_conv_cuda.cpp:151-258  (im2col + per-group addmm_)
//   modulated_deform_conv_cuda_forward      dcn/src/deform_conv_cuda.cpp:486-564  (+ bias)
// The reference materialises the column matrix (C*kh*kw x B*Hout*Wout) in HBM and calls a BLAS GEMM.
// Here a workgroup owns 64 output pixels x up to 128 output channels of one (image, group): it samples
// a 64-pixel x K-slice column tile straight into LDS (bilinear gather with the reference's exact
// border rules), stages the matching weight slice next to it, and contracts both with MFMA; the
// column matrix never exists in HBM.  Tensors are NCHW like the reference API.
#include <algorithm>
#include <type_traits>
#include "cfen_common.hpp"

int& cfen_tune_dcn_tile();
int& cfen_tune_dcn_tps();

namespace {
struct DcnArgs;
void dcn_use_scratch(int dtype, DcnArgs& a, void* columns, size_t columns_bytes);

struct DcnArgs {
  const void* im; const void* offset; const void* mask; const void* weight; const void* bias; void* out;
  int B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, group, dg, Ho, Wo;
  const void* imT;   // fast path: NHWC copy of the input  [B][H][W][C]            (k_dcn_prep, in the caller's `columns` scratch)
  const void* wT;    //            tap-major weights        [Cout][kh*kw][C/group]
  int im_nhwc;       // round 6: `im` already IS [B][H][W][C] (a channels_last tensor's memory): the layout pre-pass only re-orders the weights
};

constexpr int D_PIX = 64, D_CO = 128;

template <typename T>
CFEN_DEV float dcn_bilinear(const T* im, int H, int W, float h, float w) {   // the sampling rule of .cu:83-114, per axis as in k_dcn_lean's dcn_axis
  // a sample at (h, w) blends the four pixels around it; a neighbour outside the map contributes zero (the caller has already rejected h <= -1, h >= H, w <= -1, w >= W)
  const int y0 = (int)floorf(h), x0 = (int)floorf(w);
  const float fy = h - (float)y0, fx = w - (float)x0;
  const bool top = y0 >= 0, bottom = y0 + 1 <= H - 1, left = x0 >= 0, right = x0 + 1 <= W - 1;
  const T* row0 = im + (long long)y0 * W;
  const T* row1 = row0 + W;
  const float a00 = top && left ? (float)row0[x0] : 0.f, a01 = top && right ? (float)row0[x0 + 1] : 0.f;
  const float a10 = bottom && left ? (float)row1[x0] : 0.f, a11 = bottom && right ? (float)row1[x0 + 1] : 0.f;
  const float wy0 = 1.f - fy, wx0 = 1.f - fx;
  return wy0 * wx0 * a00 + wy0 * fx * a01 + fy * wx0 * a10 + fy * fx * a11;       // (same products and order of additions as before: bit for bit)
}

template <typename T>
__global__ __launch_bounds__(256) void k_dcn(DcnArgs a) {
  constexpr int KC = Mma<T>::KC;
  constexpr int KCH = 2 * KC;                    // K slice per stage: 128 bytes per row
  constexpr int ROWB = KCH * (int)sizeof(T) + 16;
  typedef typename Mma<T>::frag frag;
  __shared__ __attribute__((aligned(16))) unsigned char colT[D_PIX * ROWB];
  __shared__ __attribute__((aligned(16))) unsigned char Wl[D_CO * ROWB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, h = lane >> 4;
  const int Cg = a.C / a.group, Cout_g = a.Cout / a.group;
  const int ncb = (Cout_g + D_CO - 1) / D_CO;
  const int g = blockIdx.z / ncb, cb = blockIdx.z % ncb;
  const int b = blockIdx.y;
  const int HWo = a.Ho * a.Wo;
  const int kk = a.kh * a.kw;
  const int Kg = Cg * kk;
  const int cpdg = a.C / a.dg;                   // channels per deformable group

  const T* im = (const T*)a.im + (size_t)b * a.C * a.H * a.W;
  const T* off = (const T*)a.offset + (size_t)b * a.dg * 2 * kk * HWo;
  const T* msk = a.mask ? (const T*)a.mask + (size_t)b * a.dg * kk * HWo : nullptr;
  const T* wgt = (const T*)a.weight + (size_t)(g * Cout_g) * Kg;

  const int pix = tid & 63;
  const int p = blockIdx.x * D_PIX + pix;
  const bool pvalid = p < HWo;
  const int ho = pvalid ? p / a.Wo : 0, wo = pvalid ? p % a.Wo : 0;
  const int h_in = ho * a.sh - a.ph, w_in = wo * a.sw - a.pw;

  floatx4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < Kg; k0 += KCH) {
    // ---- column tile: colT[pix][kq] = sampled (and modulated) input ----
    for (int kq = tid >> 6; kq < KCH; kq += 4) {
      const int k = k0 + kq;
      float val = 0.f;
      if (pvalid && k < Kg) {
        const int c = k / kk, ij = k - c * kk;
        const int i = ij / a.kw, j = ij - i * a.kw;
        const int cim = g * Cg + c;
        const int dgi = cim / cpdg;
        const T* op = off + ((size_t)dgi * 2 * kk + 2 * ij) * HWo + p;
        const float oh = (float)op[0], ow = (float)op[HWo];
        const float h_im = (float)(h_in + i * a.dh) + oh, w_im = (float)(w_in + j * a.dw) + ow;
        if (h_im > -1.f && w_im > -1.f && h_im < (float)a.H && w_im < (float)a.W)
          val = dcn_bilinear<T>(im + (size_t)cim * a.H * a.W, a.H, a.W, h_im, w_im);
        if (msk) val *= (float)msk[((size_t)dgi * kk + ij) * HWo + p];
      }
      *reinterpret_cast<T*>(colT + pix * ROWB + kq * sizeof(T)) = (T)val;
    }
    // ---- weight slice: Wl[co][kq] ----
    for (int idx = tid; idx < D_CO * KCH; idx += 256) {
      const int row = idx / KCH, kq = idx - row * KCH;
      const int co = cb * D_CO + row, k = k0 + kq;
      T v = (co < Cout_g && k < Kg) ? wgt[(size_t)co * Kg + k] : (T)0;
      *reinterpret_cast<T*>(Wl + row * ROWB + kq * sizeof(T)) = v;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      frag af[2], bf[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const frag*>(Wl + (wave * 32 + i * 16 + r16) * ROWB + c * 64 + h * 16);
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const frag*>(colT + (j * 16 + r16) * ROWB + c * 64 + h * 16);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = Mma<T>::mma(af[i], bf[j], acc[i][j]);
    }
    __syncthreads();
  }

  T* out = (T*)a.out + ((size_t)b * a.Cout + g * Cout_g) * HWo;
  const T* bias = a.bias ? (const T*)a.bias + g * Cout_g : nullptr;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int pp = blockIdx.x * D_PIX + j * 16 + r16;
    if (pp >= HWo) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = cb * D_CO + wave * 32 + i * 16 + 4 * h + r;
        if (co < Cout_g) {
          float v = acc[i][j][r] + (bias ? (float)bias[co] : 0.f);
          out[(size_t)co * HWo + pp] = (T)v;
        }
      }
  }
}

// ---- fast path ----------------------------------------------------------------------------------------------------------------
// The kernel above re-reads the tap's offset pair and redoes the bilinear set-up for EVERY input channel and gathers 2-byte
// scalars from the NCHW planes (4 x C x k*k scalar gathers per output pixel): 1-3 % of the HBM rate.  Here a small pre-pass lays the
// input out NHWC and the weights tap-major in the caller's scratch (the reference's `columns` buffer, deform_conv_cuda.cpp:151-156 --
// the column matrix itself still never exists), and the main kernel works on 16-byte channel vectors: one (pixel, tap, channel
// vector) task loads the tap's offsets / mask ONCE per deformable group it touches, sets the four corners up once and gathers four
// 16-byte vectors -- 8x fewer gather instructions (fp16), vector LDS writes, 16-byte weight staging.  K runs tap-major
// (k' = tap * C/g + c) so a channel vector is contiguous in the column tile; the pre-pass permutes the weights to match.
template <typename T>
__global__ __launch_bounds__(256) void k_dcn_prep(const T* __restrict__ im, T* __restrict__ imT, const T* __restrict__ w, T* __restrict__ wT,
                                                  int C, long long HW, long long nim, int Cg, int kk, long long nw) {
  constexpr int VE = 16 / (int)sizeof(T);
  typedef typename Mma<T>::frag vec;
  const int cv = C / VE;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < nim + nw; idx += (long long)gridDim.x * 256) {
    if (idx < nim) {          // idx = (b * cv + v) * HW + p: consecutive lanes read consecutive pixels of one plane
      const long long p = idx % HW, bv = idx / HW;
      const int v = (int)(bv % cv);
      const long long b = bv / cv;
      vec o;
#pragma unroll
      for (int e = 0; e < VE; ++e) o[e] = im[(b * C + v * VE + e) * HW + p];
      *reinterpret_cast<vec*>(imT + (b * HW + p) * C + v * VE) = o;
    } else {                  // wT[(co * kk + ij) * Cg + c] = w[(co * Cg + c) * kk + ij]
      const long long j = idx - nim;
      const int c = (int)(j % Cg);
      const long long t = j / Cg;
      const int ij = (int)(t % kk);
      const long long co = t / kk;
      wT[j] = w[(co * Cg + c) * kk + ij];
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_dcn_nhwc(DcnArgs a) {
  constexpr int KC = Mma<T>::KC, SZ = (int)sizeof(T), VE = 16 / SZ;
  constexpr int KCH = 2 * KC;                    // K' slice per stage: 128 bytes per row
  constexpr int ROWB = KCH * SZ + 32;            // pitch = 32 (mod 64) bytes: conflict-free ds_read_b128
  constexpr int NV = KCH / VE;                   // channel vectors per staged row
  typedef typename Mma<T>::frag frag;
  __shared__ __attribute__((aligned(16))) unsigned char colT[D_PIX * ROWB];
  __shared__ __attribute__((aligned(16))) unsigned char Wl[D_CO * ROWB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, h = lane >> 4;
  const int Cg = a.C / a.group, Cout_g = a.Cout / a.group;
  const int ncb = (Cout_g + D_CO - 1) / D_CO;
  const int g = blockIdx.z / ncb, cb = blockIdx.z % ncb;
  const int b = blockIdx.y;
  const int HWo = a.Ho * a.Wo;
  const int kk = a.kh * a.kw;
  const int Kg = Cg * kk;
  const int cpdg = a.C / a.dg;
  const int rows = min(D_CO, Cout_g - cb * D_CO);          // weight rows this block owns
  const int ntile = (rows + 15) / 16;                      // 16-row MFMA tiles with data

  const T* imT = (const T*)a.imT + (size_t)b * a.H * a.W * a.C;
  const T* off = (const T*)a.offset + (size_t)b * a.dg * 2 * kk * HWo;
  const T* msk = a.mask ? (const T*)a.mask + (size_t)b * a.dg * kk * HWo : nullptr;
  const T* wT = (const T*)a.wT + (size_t)(g * Cout_g + cb * D_CO) * Kg;

  const int pix = tid & 63;
  const int p = blockIdx.x * D_PIX + pix;
  const bool pvalid = p < HWo;
  const int ho = pvalid ? p / a.Wo : 0, wo = pvalid ? p % a.Wo : 0;
  const int h_in = ho * a.sh - a.ph, w_in = wo * a.sw - a.pw;

  floatx4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < Kg; k0 += KCH) {
    // ---- column tile: colT[pix][kq .. kq + VE) = VE consecutive channels of one tap, sampled (and modulated) ----
    if (cpdg % VE == 0) {
      // Fast path (a channel vector never straddles two deformable groups): the thread's two vectors of this K slice are set up
      // together and branch-free -- both offset pairs (and masks) are loaded, then all eight corner vectors (coordinates clamped,
      // validity folded into the four bilinear weights), then the interpolation: two memory round trips per slice instead of four.
      static_assert(NV == 8, "two vectors per thread and slice");
      float oh[2], ow[2], mm[2];
      int cim[2], ti[2], tj[2];
      bool live[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int k = k0 + ((tid >> 6) + 4 * u) * VE;
        live[u] = pvalid && k < Kg;
        const int ks = live[u] ? k : 0;
        const int ij = ks / Cg, c0 = ks - ij * Cg;
        ti[u] = ij / a.kw; tj[u] = ij - ti[u] * a.kw;
        cim[u] = g * Cg + c0;
        const int dgi = cim[u] / cpdg;
        const size_t pp = pvalid ? (size_t)p : 0;
        const T* op = off + ((size_t)dgi * 2 * kk + 2 * ij) * HWo + pp;
        oh[u] = (float)op[0]; ow[u] = (float)op[HWo];
        mm[u] = msk ? (float)msk[((size_t)dgi * kk + ij) * HWo + pp] : 1.f;
      }
      float wgt[2][4];
      frag q[2][4];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const float h_im = (float)(h_in + ti[u] * a.dh) + oh[u], w_im = (float)(w_in + tj[u] * a.dw) + ow[u];
        const bool inside = live[u] && h_im > -1.f && w_im > -1.f && h_im < (float)a.H && w_im < (float)a.W;   // .cu:226-236
        const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im), h_high = h_low + 1, w_high = w_low + 1;   // .cu:83-114
        const float lh = h_im - h_low, lw = w_im - w_low, hh = 1.f - lh, hw = 1.f - lw;
        const float sc = inside ? mm[u] : 0.f;
        wgt[u][0] = (h_low >= 0 && w_low >= 0) ? hh * hw * sc : 0.f;
        wgt[u][1] = (h_low >= 0 && w_high <= a.W - 1) ? hh * lw * sc : 0.f;
        wgt[u][2] = (h_high <= a.H - 1 && w_low >= 0) ? lh * hw * sc : 0.f;
        wgt[u][3] = (h_high <= a.H - 1 && w_high <= a.W - 1) ? lh * lw * sc : 0.f;
        const int yl = min(max(h_low, 0), a.H - 1), yh = min(max(h_high, 0), a.H - 1), xl = min(max(w_low, 0), a.W - 1), xh = min(max(w_high, 0), a.W - 1);
        const T* base = imT + cim[u];
        q[u][0] = load_frag<T>(base + ((size_t)yl * a.W + xl) * a.C); q[u][1] = load_frag<T>(base + ((size_t)yl * a.W + xh) * a.C);
        q[u][2] = load_frag<T>(base + ((size_t)yh * a.W + xl) * a.C); q[u][3] = load_frag<T>(base + ((size_t)yh * a.W + xh) * a.C);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        frag o;
#pragma unroll
        for (int e = 0; e < VE; ++e)
          o[e] = (T)((wgt[u][0] * (float)q[u][0][e] + wgt[u][1] * (float)q[u][1][e] + wgt[u][2] * (float)q[u][2][e] + wgt[u][3] * (float)q[u][3][e]));
        *reinterpret_cast<frag*>(colT + pix * ROWB + ((tid >> 6) + 4 * u) * 16) = o;
      }
    } else
    for (int vq = tid >> 6; vq < NV; vq += 4) {
      const int k = k0 + vq * VE;
      float res[VE];
#pragma unroll
      for (int e = 0; e < VE; ++e) res[e] = 0.f;
      if (pvalid && k < Kg) {
        const int ij = k / Cg, c0 = k - ij * Cg;             // Cg % VE == 0: the vector stays inside one tap
        const int i = ij / a.kw, j = ij - i * a.kw;
        const int cim0 = g * Cg + c0;
        const int dg_first = cim0 / cpdg, dg_last = (cim0 + VE - 1) / cpdg;
        for (int dgi = dg_first; dgi <= dg_last; ++dgi) {     // one pass per deformable group the vector touches (1 when C/dg % VE == 0)
          const T* op = off + ((size_t)dgi * 2 * kk + 2 * ij) * HWo + p;
          const float oh = (float)op[0], ow = (float)op[HWo];
          const float m = msk ? (float)msk[((size_t)dgi * kk + ij) * HWo + p] : 1.f;
          const float h_im = (float)(h_in + i * a.dh) + oh, w_im = (float)(w_in + j * a.dw) + ow;
          if (h_im > -1.f && w_im > -1.f && h_im < (float)a.H && w_im < (float)a.W) {   // .cu:226-236
            const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);            // .cu:83-114
            const int h_high = h_low + 1, w_high = w_low + 1;
            const float lh = h_im - h_low, lw = w_im - w_low, hh = 1.f - lh, hw = 1.f - lw;
            const bool v1 = h_low >= 0 && w_low >= 0, v2 = h_low >= 0 && w_high <= a.W - 1;
            const bool v3 = h_high <= a.H - 1 && w_low >= 0, v4 = h_high <= a.H - 1 && w_high <= a.W - 1;
            const int yl = max(h_low, 0), yh = min(h_high, a.H - 1), xl = max(w_low, 0), xh = min(w_high, a.W - 1);
            const T* base = imT + cim0;
            const frag q1 = load_frag<T>(base + ((size_t)yl * a.W + xl) * a.C), q2 = load_frag<T>(base + ((size_t)yl * a.W + xh) * a.C);
            const frag q3 = load_frag<T>(base + ((size_t)yh * a.W + xl) * a.C), q4 = load_frag<T>(base + ((size_t)yh * a.W + xh) * a.C);
            const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
#pragma unroll
            for (int e = 0; e < VE; ++e) {
              const float val = w1 * (v1 ? (float)q1[e] : 0.f) + w2 * (v2 ? (float)q2[e] : 0.f) + w3 * (v3 ? (float)q3[e] : 0.f) +
                                w4 * (v4 ? (float)q4[e] : 0.f);
              if ((cim0 + e) / cpdg == dgi) res[e] = val * m;
            }
          }
        }
      }
      frag o;
#pragma unroll
      for (int e = 0; e < VE; ++e) o[e] = (T)res[e];
      *reinterpret_cast<frag*>(colT + pix * ROWB + vq * 16) = o;
    }
    // ---- weight slice: Wl[co][kq], 16-byte pieces of the tap-major copy ----
    for (int idx = tid; idx < rows * NV; idx += 256) {
      const int row = idx / NV, vq = idx - row * NV;
      const int k = k0 + vq * VE;
      *reinterpret_cast<frag*>(Wl + row * ROWB + vq * 16) = k < Kg ? load_frag<T>(wT + (size_t)row * Kg + k) : Mma<T>::zero();
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      frag bf[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const frag*>(colT + (j * 16 + r16) * ROWB + c * 64 + h * 16);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (wave * 2 + i < ntile) {       // wave-uniform: tiles past the block's rows hold stale LDS
          const frag af = *reinterpret_cast<const frag*>(Wl + (wave * 32 + i * 16 + r16) * ROWB + c * 64 + h * 16);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = Mma<T>::mma(af, bf[j], acc[i][j]);
        }
      }
    }
    __syncthreads();
  }

  T* out = (T*)a.out + ((size_t)b * a.Cout + g * Cout_g + cb * D_CO) * HWo;
  const T* bias = a.bias ? (const T*)a.bias + g * Cout_g + cb * D_CO : nullptr;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int pp = blockIdx.x * D_PIX + j * 16 + r16;
    if (pp >= HWo) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = wave * 32 + i * 16 + 4 * h + r;
        if (co < rows) out[(size_t)co * HWo + pp] = (T)(acc[i][j][r] + (bias ? (float)bias[co] : 0.f));
      }
  }
}

// ---- lean path (round 3) -------------------------------------------------------------------------------------------------------
// What bounds k_dcn_nhwc is instruction issue, not memory (tools/dcn_pmc.sh on (8,24,256,256), dg 1: L1 hit rate 93 %, texture addresser
// busy 41 %, 181 cycles of L2 latency -- and 1300-3000 VALU + 600 SALU instructions per wave, 65-100 % of the SIMDs' issue slots: 64-bit
// address arithmetic per load, integer divisions per task, a conversion per channel and corner).  Two intermediate kernels that only
// shortened the chain of dependent memory round trips (DMA'd weights, offsets up front, gathers two items ahead) or cut the L1 line
// look-ups by three (the lanes of a pixel side by side) ran in the same 180-190 us as k_dcn_nhwc's 150 + 14.  This kernel is built
// around the instruction count:
//   * a lane owns a (pixel, tap, UNIT of 3 channel vectors) task: offsets / mask loaded once, corners set up once (per-axis weights with
//     the border rules folded in, v_med3 clamps, 24-bit multiplies), 12 gathers; which (tap, unit) a wave works on comes from a table
//     in LDS (no divisions in the loop); ONE task in flight per lane (TB = 1: 96-108 registers, four waves per SIMD, beat two or three in
//     flight at 170-250), the next task's offsets loaded before this task's gathers;
//   * every global access is a buffer load: 32-bit byte offset in a VGPR, the tap's plane / the unit's channel offset in an SGPR;
//   * fp16 is interpolated with packed fp16 FMAs (v_pk_fma_f16: 2 channels per instruction; the reference's half path computes its
//     bilinear sum in half as well, deform_conv_cuda_kernel.cu:83-114 with scalar_t = at::Half); fp32 stays fp32;
//   * a K slice is `tps` whole taps sized so that four workgroups fit a CU (24 channels: 5 + 4 taps), the 16 x 16 output tiles are dealt to
//     the waves as (row tile, pixel tile) pairs;
//   * occupancy is what moved this kernel after the instruction count: every step above that raised the waves per CU gained, nothing that
//     only deepened a lane's memory-level parallelism did (DCN forward round 3);
//   * deformable groups narrower than a vector (C/dg = 3, 6, 12: the generator's dg = 8 layers): the task walks the UNIT / CPDG groups of
//     its unit, each with its own offsets, gathers CPDG channels per corner in ONE load (3 fp16 channels: the 4-byte aligned 8 bytes around
//     them) and still writes whole 16-byte vectors into the column tile -- k_dcn_nhwc re-gathers a full vector per group.
typedef _Float16 dcn_h2 __attribute__((ext_vector_type(2)));
typedef unsigned int dcn_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int dcn_u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned int dcn_u32x4 __attribute__((ext_vector_type(4)));

CFEN_DEV dcn_h2 dcn_pack_h2(float w) {
  const _Float16 hw = (_Float16)w;
  dcn_h2 r = {hw, hw};
  return r;
}

// per-axis bilinear weights with the reference's border rules folded in (.cu:83-114, 226-236): the sample counts only for
// -1 < c < n; its low neighbour only when floor(c) >= 0, its high neighbour only when floor(c) + 1 <= n - 1
struct DcnAxis { float lo, hi; int il, ih; };
CFEN_DEV DcnAxis dcn_axis(float c, float n_f, float nm1_f, int nm1) {
  const float f = floorf(c);
  const float l = c - f, hh = 1.f - l;
  const int lowi = (int)f;
  DcnAxis r;
  r.lo = (c >= 0.f && c < n_f) ? hh : 0.f;
  r.hi = (c > -1.f && c < nm1_f) ? l : 0.f;
  r.il = min(max(lowi, 0), nm1);
  r.ih = min(max(lowi + 1, 0), nm1);
  return r;
}

template <typename T> struct DcnLoad;
template <> struct DcnLoad<half_t> {
  static CFEN_DEV float scalar(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const unsigned short u = __builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0);
    _Float16 hv;
    __builtin_memcpy(&hv, &u, 2);
    return (float)hv;
  }
};
template <> struct DcnLoad<float> {
  static CFEN_DEV float scalar(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
  }
};

// wg[4] (fp32 corner weights) x four 16-byte corner vectors -> one 16-byte vector of the column tile
CFEN_DEV half8 dcn_interp(const float (&wg)[4], const dcn_h2 (&wh)[4], const dcn_u32x4 (&q)[4]) {
  dcn_u32x4 o;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    dcn_h2 c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const unsigned int w = q[k][d]; __builtin_memcpy(&c[k], &w, 4); }
    const dcn_h2 r = wh[0] * c[0] + wh[1] * c[1] + wh[2] * c[2] + wh[3] * c[3];
    unsigned int w;
    __builtin_memcpy(&w, &r, 4);
    o[d] = w;
  }
  half8 out;
  __builtin_memcpy(&out, &o, 16);
  (void)wg;
  return out;
}
CFEN_DEV floatx4 dcn_interp(const float (&wg)[4], const float (&)[4], const dcn_u32x4 (&q)[4]) {
  floatx4 o;
#pragma unroll
  for (int d = 0; d < 4; ++d)
    o[d] = wg[0] * __uint_as_float(q[0][d]) + wg[1] * __uint_as_float(q[1][d]) + wg[2] * __uint_as_float(q[2][d]) + wg[3] * __uint_as_float(q[3][d]);
  return o;
}

constexpr int DCN_MAX_TASKS = 256;     // (tap, unit) table entries
constexpr int DCN_FALLBACK = -100;     // launch_dcn_lean: not this kernel's shape after all

template <typename T, int CPDG, bool MASK, int TB, int NPAIR>
__global__ __launch_bounds__(256) void k_dcn_lean(DcnArgs a, int tps, int pitch) {
  constexpr int SZ = (int)sizeof(T), VE = 16 / SZ, UNIT = 3 * VE;
  constexpr bool ONE = CPDG == 0;                       // the unit lies inside one deformable group
  constexpr int NS = ONE ? 1 : UNIT / CPDG;             // deformable groups a unit walks through
  constexpr bool HALF = SZ == 2;
  typedef typename Mma<T>::frag frag;
  typedef typename std::conditional<HALF, dcn_h2, float>::type wpack;
  extern __shared__ __attribute__((aligned(16))) unsigned char dcn_smem[];
  __shared__ unsigned int task_tab[DCN_MAX_TASKS];      // (ti << 24) | (tj << 16) | unit
  const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, h = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nthr = blockDim.x, nwv = nthr >> 6;          // waves of the workgroup (4; 3 measured slower, see the launcher)
  const int Cg = a.C / a.group, Cout_g = a.Cout / a.group;
  const int ncb = (Cout_g + D_CO - 1) / D_CO;
  const int g = blockIdx.z / ncb, cb = blockIdx.z % ncb;
  const int b = blockIdx.y;
  const int HWo = a.Ho * a.Wo;
  const int kk = a.kh * a.kw;
  const int Kg = Cg * kk;
  const int cpdg = a.C / a.dg;
  const int rows = min(D_CO, Cout_g - cb * D_CO);
  const int ntile = (rows + 15) / 16;
  const int units = Cg / UNIT;
  unsigned char* colT = dcn_smem;
  unsigned char* Wl = dcn_smem + D_PIX * pitch;

  if (tid < kk * units) {
    const int ij = tid / units, un = tid - ij * units;
    const int ti = ij / a.kw, tj = ij - ti * a.kw;
    task_tab[tid] = ((unsigned)ti << 24) | ((unsigned)tj << 16) | (unsigned)un;
  }

  const __amdgpu_buffer_rsrc_t im_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>((const T*)a.imT + (size_t)b * a.H * a.W * a.C), 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t off_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>((const T*)a.offset + (size_t)b * a.dg * 2 * kk * HWo), 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t msk_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(MASK ? (const T*)a.mask + (size_t)b * a.dg * kk * HWo : (const T*)a.offset), 0, -1, 0x00020000);
  const T* wT = (const T*)a.wT + (size_t)(g * Cout_g + cb * D_CO) * Kg;

  const int p = blockIdx.x * D_PIX + lane;
  const bool pvalid = p < HWo;
  const int pp = pvalid ? p : 0;
  const int ho = pp / a.Wo, wo = pp - ho * a.Wo;
  const float hb = (float)(ho * a.sh - a.ph), wb = (float)(wo * a.sw - a.pw);
  const int voff_p = pp * SZ;
  const float H_f = (float)a.H, W_f = (float)a.W, Hm1_f = (float)(a.H - 1), Wm1_f = (float)(a.W - 1);
  const unsigned rowb = (unsigned)(a.W * a.C * SZ), pixb = (unsigned)(a.C * SZ);      // < 2^24 (launcher)
  const int plane = HWo * SZ;                                                           // bytes of one offset / mask plane

  // output tiles (row tile i, pixel tile j) = pair i * 4 + j, dealt round-robin to the waves: pair wave + nwv * n is this wave's n-th
  floatx4 acc[NPAIR];                                 // NPAIR >= ceil(4 * row tiles / waves): 3 serves up to 32 output rows, 11 everything (launcher)
#pragma unroll
  for (int n = 0; n < NPAIR; ++n) acc[n] = floatx4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();                                      // task table

  for (int tap0 = 0; tap0 < kk; tap0 += tps) {
    const int nt = min(tps, kk - tap0);
    const int kb = nt * Cg * SZ, kb64 = (kb + 63) & ~63;          // bytes of a slice row with data / as the MFMA loop walks it
    const int ntask = nt * units, e0 = tap0 * units;
    // ---- column tile: colT[pixel][tap-in-slice * Cg + c] ----
    // (whole-vector groups, one task in flight) the NEXT task's offsets / mask are loaded before this task's gathers: one memory round trip per
    // task instead of two
    float p_oh = 0.f, p_ow = 0.f, p_mm = 1.f;
    auto load_offsets = [&](int t, float& oh_, float& ow_, float& mm_) {
      const unsigned ent = __builtin_amdgcn_readfirstlane(task_tab[e0 + t]);
      const int ti = ent >> 24, tj = (ent >> 16) & 255, un = ent & 0xffff;
      const int ij = ti * a.kw + tj, dgi = (g * Cg + un * UNIT) / cpdg;
      const int so = (dgi * 2 * kk + 2 * ij) * plane;
      oh_ = DcnLoad<T>::scalar(off_rsrc, voff_p, so);
      ow_ = DcnLoad<T>::scalar(off_rsrc, voff_p, so + plane);
      mm_ = MASK ? DcnLoad<T>::scalar(msk_rsrc, voff_p, (dgi * kk + ij) * plane) : 1.f;
    };
    if constexpr (ONE && TB == 1) {
      if (wave < ntask) load_offsets(wave, p_oh, p_ow, p_mm);
    }
    for (int t0 = wave; t0 < ntask; t0 += nwv * TB) {
      if constexpr (ONE) {
        float oh[TB], ow[TB], mm[TB], fi[TB], fj[TB];
        int csoff[TB], dst[TB];
        bool live[TB];
#pragma unroll
        for (int u = 0; u < TB; ++u) {
          const int t = t0 + nwv * u;
          live[u] = t < ntask;                                     // wave-uniform
          const unsigned ent = __builtin_amdgcn_readfirstlane(task_tab[e0 + (live[u] ? t : 0)]);
          const int ti = ent >> 24, tj = (ent >> 16) & 255, un = ent & 0xffff;
          const int ij = ti * a.kw + tj, cim = g * Cg + un * UNIT;
          const int dgi = cim / cpdg;                              // scalar
          fi[u] = (float)(ti * a.dh); fj[u] = (float)(tj * a.dw);
          csoff[u] = cim * SZ;
          dst[u] = ((ij - tap0) * Cg + un * UNIT) * SZ;
          if constexpr (TB == 1) {
            oh[u] = p_oh; ow[u] = p_ow; mm[u] = p_mm;
            if (t0 + nwv < ntask) load_offsets(t0 + nwv, p_oh, p_ow, p_mm);
          } else {
            const int so = (dgi * 2 * kk + 2 * ij) * plane;
            oh[u] = DcnLoad<T>::scalar(off_rsrc, voff_p, so);
            ow[u] = DcnLoad<T>::scalar(off_rsrc, voff_p, so + plane);
            mm[u] = MASK ? DcnLoad<T>::scalar(msk_rsrc, voff_p, (dgi * kk + ij) * plane) : 1.f;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        float wg[TB][4];
        wpack wh[TB][4];
        dcn_u32x4 q[TB][3][4];
#pragma unroll
        for (int u = 0; u < TB; ++u) {
          const DcnAxis ay = dcn_axis(hb + fi[u] + oh[u], H_f, Hm1_f, a.H - 1), ax = dcn_axis(wb + fj[u] + ow[u], W_f, Wm1_f, a.W - 1);
          const float m = (live[u] && pvalid) ? mm[u] : 0.f;
          const float xl = ax.lo * m, xh = ax.hi * m;
          wg[u][0] = ay.lo * xl; wg[u][1] = ay.lo * xh; wg[u][2] = ay.hi * xl; wg[u][3] = ay.hi * xh;
          const unsigned yl = __umul24(ay.il, rowb), yh = __umul24(ay.ih, rowb), xlb = __umul24(ax.il, pixb), xhb = __umul24(ax.ih, pixb);
          const int vo[4] = {(int)(yl + xlb), (int)(yl + xhb), (int)(yh + xlb), (int)(yh + xhb)};
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int v = 0; v < 3; ++v) q[u][v][k] = __builtin_amdgcn_raw_buffer_load_b128(im_rsrc, vo[k] + v * 16, csoff[u], 0);
        }
#pragma unroll
        for (int u = 0; u < TB; ++u) {
          asm volatile("" : "+v"(wg[u][0]), "+v"(wg[u][1]), "+v"(wg[u][2]), "+v"(wg[u][3]));   // no interpolation (and its wait) between the gathers
          if constexpr (HALF) {
#pragma unroll
            for (int k = 0; k < 4; ++k) wh[u][k] = dcn_pack_h2(wg[u][k]);
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) wh[u][k] = wg[u][k];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < TB; ++u) {
          if (!live[u]) continue;
#pragma unroll
          for (int v = 0; v < 3; ++v) *reinterpret_cast<frag*>(colT + lane * pitch + dst[u] + v * 16) = dcn_interp(wg[u], wh[u], q[u][v]);
        }
      } else {
        // one task at a time: NS groups x 4 corners are NS * 4 independent gathers already
        constexpr int SB = CPDG * SZ;                                // bytes of a group's channels at one corner
        constexpr int LW = SB == 6 ? 2 : SB / 4;                     // dwords per load: 6 bytes -> the aligned 8 around them, 12 -> 3, 24 -> 4 + 2
        for (int t = t0; t < ntask && t < t0 + nwv * TB; t += nwv) {
          const unsigned ent = __builtin_amdgcn_readfirstlane(task_tab[e0 + t]);
          const int ti = ent >> 24, tj = (ent >> 16) & 255, un = ent & 0xffff;
          const int ij = ti * a.kw + tj, cim0 = g * Cg + un * UNIT;
          const int dg0 = cim0 / CPDG;
          const float fi = (float)(ti * a.dh), fj = (float)(tj * a.dw);
          unsigned int res[12];                                        // the unit: 3 vectors of 16 bytes
          // groups per batch of gathers: all of them (3-channel groups: 32 gathers in flight at 160 registers measured 231 / 308 us on (8,24,256,256)
          // dg 8, two batches of 16 at 120 registers 243 / 331; with 28 KB slices 231 / 281 against 246 / 339)
          constexpr int SEGB = NS;
#pragma unroll
          for (int s0 = 0; s0 < NS; s0 += SEGB) {
            float oh[SEGB], ow[SEGB], mm[SEGB];
#pragma unroll
            for (int s = 0; s < SEGB; ++s) {
              const int so = ((dg0 + s0 + s) * 2 * kk + 2 * ij) * plane;
              oh[s] = DcnLoad<T>::scalar(off_rsrc, voff_p, so);
              ow[s] = DcnLoad<T>::scalar(off_rsrc, voff_p, so + plane);
              mm[s] = MASK ? DcnLoad<T>::scalar(msk_rsrc, voff_p, ((dg0 + s0 + s) * kk + ij) * plane) : 1.f;
            }
            __builtin_amdgcn_sched_barrier(0);
            float wg[SEGB][4];
            unsigned int raw[SEGB][4][LW == 6 ? 6 : LW];
#pragma unroll
            for (int s = 0; s < SEGB; ++s) {
              const DcnAxis ay = dcn_axis(hb + fi + oh[s], H_f, Hm1_f, a.H - 1), ax = dcn_axis(wb + fj + ow[s], W_f, Wm1_f, a.W - 1);
              const float m = pvalid ? mm[s] : 0.f;
              const float xl = ax.lo * m, xh = ax.hi * m;
              wg[s][0] = ay.lo * xl; wg[s][1] = ay.lo * xh; wg[s][2] = ay.hi * xl; wg[s][3] = ay.hi * xh;
              const unsigned yl = __umul24(ay.il, rowb), yh = __umul24(ay.ih, rowb), xlb = __umul24(ax.il, pixb), xhb = __umul24(ax.ih, pixb);
              const int vo[4] = {(int)(yl + xlb), (int)(yl + xhb), (int)(yh + xlb), (int)(yh + xhb)};
              const int co = ((s0 + s) * CPDG - (SB == 6 ? (s & 1) : 0)) * SZ;          // compile-time after unrolling: folded into the instruction's offset
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                if constexpr (LW == 2) {
                  const dcn_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(im_rsrc, vo[k] + co, cim0 * SZ, 0);
                  raw[s][k][0] = v[0]; raw[s][k][1] = v[1];
                } else if constexpr (LW == 3) {
                  const dcn_u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(im_rsrc, vo[k] + co, cim0 * SZ, 0);
                  raw[s][k][0] = v[0]; raw[s][k][1] = v[1]; raw[s][k][2] = v[2];
                } else {
                  static_assert(LW == 6, "3, 6 or 12 channels");
                  const dcn_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(im_rsrc, vo[k] + co, cim0 * SZ, 0);
                  const dcn_u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(im_rsrc, vo[k] + co + 16, cim0 * SZ, 0);
                  raw[s][k][0] = v[0]; raw[s][k][1] = v[1]; raw[s][k][2] = v[2]; raw[s][k][3] = v[3]; raw[s][k][4] = w[0]; raw[s][k][5] = w[1];
                }
              }
            }
#pragma unroll
            for (int s = 0; s < SEGB; ++s) asm volatile("" : "+v"(wg[s][0]), "+v"(wg[s][1]), "+v"(wg[s][2]), "+v"(wg[s][3]));
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (HALF) {
              constexpr int ND = LW == 6 ? 6 : LW;                       // dwords (channel pairs) of a group's load
              unsigned int part[SEGB][ND];
#pragma unroll
              for (int s = 0; s < SEGB; ++s) {
                dcn_h2 wh[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) wh[k] = dcn_pack_h2(wg[s][k]);
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                  dcn_h2 c[4];
#pragma unroll
                  for (int k = 0; k < 4; ++k) __builtin_memcpy(&c[k], &raw[s][k][d], 4);
                  const dcn_h2 r = wh[0] * c[0] + wh[1] * c[1] + wh[2] * c[2] + wh[3] * c[3];
                  __builtin_memcpy(&part[s][d], &r, 4);
                }
              }
              if constexpr (SB == 6) {
                // even group: [c0 c1 | c2 x], odd group (loaded one channel early): [x c0 | c1 c2] -> six channels = three pairs
#pragma unroll
                for (int s2 = 0; s2 < SEGB / 2; ++s2) {
                  res[(s0 / 2 + s2) * 3 + 0] = part[2 * s2][0];
                  res[(s0 / 2 + s2) * 3 + 1] = (part[2 * s2][1] & 0xffffu) | (part[2 * s2 + 1][0] & 0xffff0000u);
                  res[(s0 / 2 + s2) * 3 + 2] = part[2 * s2 + 1][1];
                }
              } else {
#pragma unroll
                for (int s = 0; s < SEGB; ++s)
#pragma unroll
                  for (int d = 0; d < ND; ++d) res[(s0 + s) * ND + d] = part[s][d];
              }
            } else {
#pragma unroll
              for (int s = 0; s < SEGB; ++s)
#pragma unroll
                for (int e = 0; e < CPDG; ++e)
                  res[(s0 + s) * CPDG + e] = __float_as_uint(wg[s][0] * __uint_as_float(raw[s][0][e]) + wg[s][1] * __uint_as_float(raw[s][1][e]) +
                                                      wg[s][2] * __uint_as_float(raw[s][2][e]) + wg[s][3] * __uint_as_float(raw[s][3][e]));
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          const int dstb = ((ij - tap0) * Cg + un * UNIT) * SZ;
#pragma unroll
          for (int v = 0; v < 3; ++v) {
            const dcn_u32x4 o = {res[v * 4], res[v * 4 + 1], res[v * 4 + 2], res[v * 4 + 3]};
            *reinterpret_cast<dcn_u32x4*>(colT + lane * pitch + dstb + v * 16) = o;
          }
        }
      }
    }
    // ---- weight slice (16-byte pieces of the tap-major copy), zero in the rows past the block's and in the columns [kb, kb64) ----
    const int npc = kb64 / 16;
    for (int idx = tid; idx < ntile * 16 * npc; idx += nthr) {
      const int row = idx / npc, pc = idx - row * npc;
      *reinterpret_cast<frag*>(Wl + row * pitch + pc * 16) =
          (row < rows && pc * 16 < kb) ? load_frag<T>(wT + (size_t)row * Kg + tap0 * Cg + pc * VE) : Mma<T>::zero();
    }
    if (kb64 > kb) {
      const int nz = (kb64 - kb) / 16;
      for (int idx = tid; idx < D_PIX * nz; idx += nthr) {
        const int row = idx / nz, pc = idx - row * nz;
        *reinterpret_cast<frag*>(colT + row * pitch + kb + pc * 16) = Mma<T>::zero();
      }
    }
    __syncthreads();
    const int npair = ntile * 4;
    for (int c = 0; c < kb64 / 64; ++c) {
#pragma unroll
      for (int n = 0; n < NPAIR; ++n) {
        const int pr = wave + nwv * n;                     // wave-uniform
        if (pr < npair) {
          const int i = pr >> 2, j = pr & 3;
          acc[n] = Mma<T>::mma(*reinterpret_cast<const frag*>(Wl + (i * 16 + r16) * pitch + c * 64 + h * 16),
                               *reinterpret_cast<const frag*>(colT + (j * 16 + r16) * pitch + c * 64 + h * 16), acc[n]);
        }
      }
    }
    __syncthreads();
  }

  T* out = (T*)a.out + ((size_t)b * a.Cout + g * Cout_g + cb * D_CO) * HWo;
  const T* bias = a.bias ? (const T*)a.bias + g * Cout_g + cb * D_CO : nullptr;
  const T* bp = bias ? bias : wT;                   // branch-free bias loads (a branch per value is a memory round trip per value)
#pragma unroll
  for (int n = 0; n < NPAIR; ++n) {
    const int pr = wave + nwv * n;
    if (pr >= ntile * 4) break;
    const int i = pr >> 2, j = pr & 3;
    const int po = blockIdx.x * D_PIX + j * 16 + r16;
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = (float)bp[min(i * 16 + 4 * h + r, rows - 1)];
    if (po >= HWo) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = i * 16 + 4 * h + r;
      if (co < rows) out[(size_t)co * HWo + po] = (T)(acc[n][r] + (bias ? bv[r] : 0.f));
    }
  }
}

// which k_dcn_lean instance serves the problem (0 = none): the unit of 3 channel vectors must tile the group's channels and either lie
// inside one deformable group or be cut by the groups into equal parts of 3, 6 or 12 channels
template <typename T>
int dcn_lean_mode(const DcnArgs& a) {
  constexpr int SZ = (int)sizeof(T), VE = 16 / SZ, UNIT = 3 * VE;
  const int Cg = a.C / a.group, cpdg = a.C / a.dg, kk = a.kh * a.kw;
  if (Cg % UNIT || kk * (Cg / UNIT) > DCN_MAX_TASKS || a.kh > 255 || a.kw > 255) return 0;
  // 24-bit multiplies for the corner offsets, 32-bit byte offsets into one image / one image's offset planes
  if ((long long)a.W * a.C * SZ >= (1 << 24) || a.H >= (1 << 24) || (long long)a.H * a.W * a.C * SZ >= (1ll << 31) ||
      (long long)a.dg * 2 * kk * a.Ho * a.Wo * SZ >= (1ll << 31))
    return 0;
  if (cpdg % UNIT == 0) return 1;
  if (UNIT % cpdg == 0 && (cpdg == 3 || cpdg == 6 || (cpdg == 12 && VE == 8))) return cpdg;
  return 0;
}

template <typename T, bool MASK>
int launch_dcn_lean(const DcnArgs& a, int mode, dim3 grid, hipStream_t s) {
  constexpr int SZ = (int)sizeof(T);
  const int Cg = a.C / a.group, Cout_g = a.Cout / a.group, kk = a.kh * a.kw;
  const int rows16 = (std::min(D_CO, Cout_g) + 15) / 16 * 16;
  // taps per slice: as many whole taps as fit the LDS budget, then evened out over the slices.  The budget is occupancy: nine taps of a 24-channel
  // map in one slice (44 KB: three workgroups per CU) take 121 us, 5 + 4 taps (28 KB: four, the register limit) 107 us, 3 + 3 + 3 114 us
  auto pitch_of = [&](int t) { return ((t * Cg * SZ + 63) & ~63) + 32; };   // 32 (mod 64) bytes: conflict-free 16-byte reads and writes
  int tmax = 0;
  for (int t = 1; t <= kk; ++t)
    if ((size_t)(D_PIX + rows16) * pitch_of(t) <= (t == 1 ? 60 : 38) * 1024) tmax = t;   // four workgroups per CU (a one-tap slice may take up to 60 KB)
  if (!tmax) return DCN_FALLBACK;                      // a one-tap slice does not fit: the caller falls back to k_dcn_nhwc
  const int nsl = (kk + tmax - 1) / tmax;
  int tps = (kk + nsl - 1) / nsl;
  if (cfen_tune_dcn_tps() > 0) tps = std::min(tps, cfen_tune_dcn_tps());
  const int pitch = pitch_of(tps);
  const size_t lds = (size_t)(D_PIX + rows16) * pitch;
  // four waves also where a slice's tasks would divide evenly over three (nine taps of a 24-channel map: 3 + 2 + 2 + 2): 119 us against 132 with
  // three waves of three tasks -- the fourth wave's latency hiding is worth more than the balance (the kernel takes either: blockDim.x / 64)
  const dim3 blk(256);
  // one task in flight per lane everywhere: 116 registers (four waves per SIMD) beat two or three tasks in flight at 170-250 (measured, section 4.3)
  const bool small = rows16 <= 32;                     // <= 8 output tiles: three accumulators per wave (whatever the wave count)
#define DCN_LEAN(CP) do { if (small) CFEN_LAUNCH((k_dcn_lean<T, CP, MASK, 1, 3>), grid, blk, lds, s, a, tps, pitch); \
                          else CFEN_LAUNCH((k_dcn_lean<T, CP, MASK, 1, 11>), grid, blk, lds, s, a, tps, pitch); } while (0)
  switch (mode) {
    case 1: DCN_LEAN(0); break;
    case 3: DCN_LEAN(3); break;
    case 6: DCN_LEAN(6); break;
    default:
      if constexpr (SZ == 2) { DCN_LEAN(12); break; }
      cfen_set_error("deform_conv: lean mode %d", mode);
      return CFEN_ERR_ARG;
  }
#undef DCN_LEAN
  CFEN_CHECK_LAUNCH("deform_conv (lean)");
  return CFEN_OK;
}

int launch_dcn(int dtype, const DcnArgs& a, hipStream_t s) {
  CFEN_CHECK_ARG(a.im && a.offset && a.weight && a.out, "deform_conv: null tensor");
  // shape_check (dcn/src/deform_conv_cuda.cpp:61-149)
  CFEN_CHECK_ARG(a.kw > 0 && a.kh > 0, "kernel size should be greater than zero, but got kH: %d kW: %d", a.kh, a.kw);
  CFEN_CHECK_ARG(a.sw > 0 && a.sh > 0, "stride should be greater than zero, but got dH: %d dW: %d", a.sh, a.sw);
  CFEN_CHECK_ARG(a.dw > 0 && a.dh > 0, "dilation should be greater than 0, but got dilationH: %d dilationW: %d", a.dh, a.dw);
  CFEN_CHECK_ARG(a.B > 0 && a.C > 0 && a.Cout > 0 && a.group > 0 && a.dg > 0, "deform_conv: empty problem");
  CFEN_CHECK_ARG(a.C % a.group == 0 && a.Cout % a.group == 0, "deform_conv: channels must be divisible by groups");
  CFEN_CHECK_ARG(a.C % a.dg == 0, "input channels must divide deformable group size");
  CFEN_CHECK_ARG(a.Ho >= 1 && a.Wo >= 1, "Given input size: (%d x %d x %d). Calculated output size: (%d x %d x %d). Output size is too small",
                 a.C, a.H, a.W, a.Cout, a.Ho, a.Wo);
  CFEN_CHECK_ARG(a.H >= a.kh && a.W >= a.kw, "input image is smaller than kernel");
  const int Cout_g = a.Cout / a.group;
  const int ncb = (Cout_g + D_CO - 1) / D_CO;
  const long long HWo = (long long)a.Ho * a.Wo;
  dim3 grid((unsigned)((HWo + D_PIX - 1) / D_PIX), a.B, a.group * ncb);
  CFEN_CHECK_ARG(grid.y <= 65535 && grid.z <= 65535, "deform_conv: batch / groups too large for one launch");
  if (a.imT && (dtype == 0 || dtype == 1)) {   // fast path: NHWC + tap-major copies in the caller's scratch
    const int lean_mode = !cfen_tune_dcn_tile() ? 0 : dtype == 1 ? dcn_lean_mode<half_t>(a) : dcn_lean_mode<float>(a);
    const long long HW = (long long)a.H * a.W, nim = a.im_nhwc ? 0 : (long long)a.B * (a.C / (dtype == 1 ? 8 : 4)) * HW;
    const long long nw = (long long)a.Cout * (a.C / a.group) * a.kh * a.kw;
    const unsigned pg = (unsigned)std::min<long long>((nim + nw + 255) / 256, 8192);
    if (dtype == 1) {
      CFEN_LAUNCH(k_dcn_prep<half_t>, dim3(pg), dim3(256), 0, s, (const half_t*)a.im, (half_t*)a.imT, (const half_t*)a.weight, (half_t*)a.wT, a.C, HW, nim,
                  a.C / a.group, a.kh * a.kw, nw);
      CFEN_CHECK_LAUNCH("deform_conv (layout pre-pass)");
      if (lean_mode) { const int rc = a.mask ? launch_dcn_lean<half_t, true>(a, lean_mode, grid, s) : launch_dcn_lean<half_t, false>(a, lean_mode, grid, s); if (rc != DCN_FALLBACK) return rc; }
      CFEN_LAUNCH(k_dcn_nhwc<half_t>, grid, dim3(256), 0, s, a);
    } else {
      CFEN_LAUNCH(k_dcn_prep<float>, dim3(pg), dim3(256), 0, s, (const float*)a.im, (float*)a.imT, (const float*)a.weight, (float*)a.wT, a.C, HW, nim,
                  a.C / a.group, a.kh * a.kw, nw);
      CFEN_CHECK_LAUNCH("deform_conv (layout pre-pass)");
      if (lean_mode) { const int rc = a.mask ? launch_dcn_lean<float, true>(a, lean_mode, grid, s) : launch_dcn_lean<float, false>(a, lean_mode, grid, s); if (rc != DCN_FALLBACK) return rc; }
      CFEN_LAUNCH(k_dcn_nhwc<float>, grid, dim3(256), 0, s, a);
    }
  } else if (dtype == 1)
    CFEN_LAUNCH(k_dcn<half_t>, grid, dim3(256), 0, s, a);
  else if (dtype == 0)
    CFEN_LAUNCH(k_dcn<float>, grid, dim3(256), 0, s, a);
  else {
    cfen_set_error("deform_conv: dtype %d unsupported (fp32, fp16; the reference's fp64 dispatch is not provided)", dtype);
    return CFEN_ERR_ARG;
  }
  CFEN_CHECK_LAUNCH("deform_conv");
  return CFEN_OK;
}

// the fast path needs C/group a multiple of the 16-byte channel vector and enough 256-byte-aligned scratch for both copies
void dcn_use_scratch(int dtype, DcnArgs& a, void* columns, size_t columns_bytes) {
  if (!columns || (dtype != 0 && dtype != 1) || a.group <= 0 || a.C <= 0 || a.kh <= 0 || a.kw <= 0) return;
  const size_t esz = dtype == 1 ? 2 : 4, ve = 16 / esz;
  if (a.C % a.group || (size_t)(a.C / a.group) % ve || !cfen_aligned16(columns)) return;
  const size_t im_bytes = ((size_t)a.B * a.H * a.W * a.C * esz + 255) / 256 * 256;
  const size_t w_bytes = (size_t)a.Cout * (a.C / a.group) * a.kh * a.kw * esz;
  if (columns_bytes < im_bytes + w_bytes) return;
  a.imT = a.im_nhwc ? a.im : columns;
  a.wT = (unsigned char*)columns + im_bytes;
}

}  // namespace

int& cfen_tune_dcn_tile() {   // 1 (default): k_dcn_lean where its shapes allow; 0: k_dcn_nhwc (round 2)
  static int v = 1;
  return v;
}

int& cfen_tune_dcn_tps() {   // taps per K slice of k_dcn_lean at most this (0: as many as fit 60 KB of LDS) ("dcn.tps")
  static int v = 0;
  return v;
}

extern "C" {

size_t cfen_deform_conv_columns_bytes(int dtype, int B, int Cin, int H, int W, int Cout, int kH, int kW, int group) {
  if ((dtype != 0 && dtype != 1) || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || kH <= 0 || kW <= 0 || group <= 0) return 0;
  const size_t esz = dtype == 1 ? 2 : 4;
  return ((size_t)B * H * W * Cin * esz + 255) / 256 * 256 + (size_t)Cout * (Cin / group) * kH * kW * esz;
}

static int dcn_forward_v1(int nhwc, int dtype, const void* input, const void* weight, const void* offset, void* output, int B, int Cin, int H,
                          int W, int Cout, int kW, int kH, int dW, int dH, int padW, int padH, int dilationW, int dilationH, int group,
                          int deformable_group, int im2col_step, void* columns, size_t columns_bytes, void* stream) {
  CFEN_CHECK_ARG(im2col_step > 0 && B % (im2col_step < B ? im2col_step : B) == 0, "im2col step must divide batchsize");
  DcnArgs a{input, offset, nullptr, weight, nullptr, output, B, Cin, H, W, Cout, kH, kW, dH, dW, padH, padW, dilationH, dilationW,
            group, deformable_group, 0, 0, nullptr, nullptr, nhwc};
  dcn_use_scratch(dtype, a, columns, columns_bytes);
  CFEN_CHECK_ARG(!nhwc || a.imT, "deform_conv (NHWC input): needs the `columns` scratch and C / group a multiple of the 16-byte channel vector");
  if (dH > 0 && dW > 0) {
    a.Ho = (H + 2 * padH - (dilationH * (kH - 1) + 1)) / dH + 1;
    a.Wo = (W + 2 * padW - (dilationW * (kW - 1) + 1)) / dW + 1;
  }
  return launch_dcn(dtype, a, (hipStream_t)stream);
}

int cfen_deform_conv_forward(int dtype, const void* input, const void* weight, const void* offset, void* output, int B, int Cin, int H,
                             int W, int Cout, int kW, int kH, int dW, int dH, int padW, int padH, int dilationW, int dilationH, int group,
                             int deformable_group, int im2col_step, void* columns, size_t columns_bytes, void* stream) {
  return dcn_forward_v1(0, dtype, input, weight, offset, output, B, Cin, H, W, Cout, kW, kH, dW, dH, padW, padH, dilationW, dilationH, group, deformable_group, im2col_step,
                        columns, columns_bytes, stream);
}
int cfen_deform_conv_forward_nhwc(int dtype, const void* input, const void* weight, const void* offset, void* output, int B, int Cin, int H,
                                  int W, int Cout, int kW, int kH, int dW, int dH, int padW, int padH, int dilationW, int dilationH, int group,
                                  int deformable_group, int im2col_step, void* columns, size_t columns_bytes, void* stream) {
  return dcn_forward_v1(1, dtype, input, weight, offset, output, B, Cin, H, W, Cout, kW, kH, dW, dH, padW, padH, dilationW, dilationH, group, deformable_group, im2col_step,
                        columns, columns_bytes, stream);
}

static int dcn_forward_v2(int nhwc, int dtype, const void* input, const void* weight, const void* bias, const void* offset,
                          const void* mask, void* output, int B, int Cin, int H, int W, int Cout, int kernel_h, int kernel_w,
                          int stride_h, int stride_w, int pad_h, int pad_w, int dilation_h, int dilation_w, int group,
                          int deformable_group, int with_bias, void* columns, size_t columns_bytes, void* stream) {
  CFEN_CHECK_ARG(mask != nullptr, "modulated_deform_conv: mask is required");
  CFEN_CHECK_ARG(!with_bias || bias, "modulated_deform_conv: with_bias set but bias is null");
  DcnArgs a{input, offset, mask, weight, with_bias ? bias : nullptr, output, B, Cin, H, W, Cout, kernel_h, kernel_w, stride_h, stride_w,
            pad_h, pad_w, dilation_h, dilation_w, group, deformable_group, 0, 0, nullptr, nullptr, nhwc};
  dcn_use_scratch(dtype, a, columns, columns_bytes);
  CFEN_CHECK_ARG(!nhwc || a.imT, "modulated_deform_conv (NHWC input): needs the `columns` scratch and C / group a multiple of the 16-byte channel vector");
  if (stride_h > 0 && stride_w > 0) {
    a.Ho = (H + 2 * pad_h - (dilation_h * (kernel_h - 1) + 1)) / stride_h + 1;
    a.Wo = (W + 2 * pad_w - (dilation_w * (kernel_w - 1) + 1)) / stride_w + 1;
  }
  return launch_dcn(dtype, a, (hipStream_t)stream);
}
int cfen_modulated_deform_conv_forward(int dtype, const void* input, const void* weight, const void* bias, const void* offset,
                                       const void* mask, void* output, int B, int Cin, int H, int W, int Cout, int kernel_h, int kernel_w,
                                       int stride_h, int stride_w, int pad_h, int pad_w, int dilation_h, int dilation_w, int group,
                                       int deformable_group, int with_bias, void* columns, size_t columns_bytes, void* stream) {
  return dcn_forward_v2(0, dtype, input, weight, bias, offset, mask, output, B, Cin, H, W, Cout, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group,
                        deformable_group, with_bias, columns, columns_bytes, stream);
}
int cfen_modulated_deform_conv_forward_nhwc(int dtype, const void* input, const void* weight, const void* bias, const void* offset,
                                            const void* mask, void* output, int B, int Cin, int H, int W, int Cout, int kernel_h, int kernel_w,
                                            int stride_h, int stride_w, int pad_h, int pad_w, int dilation_h, int dilation_w, int group,
                                            int deformable_group, int with_bias, void* columns, size_t columns_bytes, void* stream) {
  return dcn_forward_v2(1, dtype, input, weight, bias, offset, mask, output, B, Cin, H, W, Cout, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group,
                        deformable_group, with_bias, columns, columns_bytes, stream);
}

}  // extern "C"
