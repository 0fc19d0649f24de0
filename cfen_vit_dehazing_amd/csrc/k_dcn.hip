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
#include "cfen_common.hpp"

namespace {
struct DcnArgs;
void dcn_use_scratch(int dtype, DcnArgs& a, void* columns, size_t columns_bytes);

struct DcnArgs {
  const void* im; const void* offset; const void* mask; const void* weight; const void* bias; void* out;
  int B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, group, dg, Ho, Wo;
  const void* imT;   // fast path: NHWC copy of the input  [B][H][W][C]            (k_dcn_prep, in the caller's `columns` scratch)
  const void* wT;    //            tap-major weights        [Cout][kh*kw][C/group]
};

constexpr int D_PIX = 64, D_CO = 128;

template <typename T>
CFEN_DEV float dcn_bilinear(const T* im, int H, int W, float h, float w) {   // .cu:83-114
  int h_low = (int)floorf(h), w_low = (int)floorf(w);
  int h_high = h_low + 1, w_high = w_low + 1;
  float lh = h - h_low, lw = w - w_low, hh = 1.f - lh, hw = 1.f - lw;
  float v1 = (h_low >= 0 && w_low >= 0) ? (float)im[h_low * W + w_low] : 0.f;
  float v2 = (h_low >= 0 && w_high <= W - 1) ? (float)im[h_low * W + w_high] : 0.f;
  float v3 = (h_high <= H - 1 && w_low >= 0) ? (float)im[h_high * W + w_low] : 0.f;
  float v4 = (h_high <= H - 1 && w_high <= W - 1) ? (float)im[h_high * W + w_high] : 0.f;
  return hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4;
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
    const long long HW = (long long)a.H * a.W, nim = (long long)a.B * (a.C / (dtype == 1 ? 8 : 4)) * HW;
    const long long nw = (long long)a.Cout * (a.C / a.group) * a.kh * a.kw;
    const unsigned pg = (unsigned)std::min<long long>((nim + nw + 255) / 256, 8192);
    if (dtype == 1) {
      CFEN_LAUNCH(k_dcn_prep<half_t>, dim3(pg), dim3(256), 0, s, (const half_t*)a.im, (half_t*)a.imT, (const half_t*)a.weight, (half_t*)a.wT, a.C, HW, nim,
                  a.C / a.group, a.kh * a.kw, nw);
      CFEN_CHECK_LAUNCH("deform_conv (layout pre-pass)");
      CFEN_LAUNCH(k_dcn_nhwc<half_t>, grid, dim3(256), 0, s, a);
    } else {
      CFEN_LAUNCH(k_dcn_prep<float>, dim3(pg), dim3(256), 0, s, (const float*)a.im, (float*)a.imT, (const float*)a.weight, (float*)a.wT, a.C, HW, nim,
                  a.C / a.group, a.kh * a.kw, nw);
      CFEN_CHECK_LAUNCH("deform_conv (layout pre-pass)");
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
  a.imT = columns;
  a.wT = (unsigned char*)columns + im_bytes;
}

}  // namespace

extern "C" {

size_t cfen_deform_conv_columns_bytes(int dtype, int B, int Cin, int H, int W, int Cout, int kH, int kW, int group) {
  if ((dtype != 0 && dtype != 1) || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || kH <= 0 || kW <= 0 || group <= 0) return 0;
  const size_t esz = dtype == 1 ? 2 : 4;
  return ((size_t)B * H * W * Cin * esz + 255) / 256 * 256 + (size_t)Cout * (Cin / group) * kH * kW * esz;
}

int cfen_deform_conv_forward(int dtype, const void* input, const void* weight, const void* offset, void* output, int B, int Cin, int H,
                             int W, int Cout, int kW, int kH, int dW, int dH, int padW, int padH, int dilationW, int dilationH, int group,
                             int deformable_group, int im2col_step, void* columns, size_t columns_bytes, void* stream) {
  CFEN_CHECK_ARG(im2col_step > 0 && B % (im2col_step < B ? im2col_step : B) == 0, "im2col step must divide batchsize");
  DcnArgs a{input, offset, nullptr, weight, nullptr, output, B, Cin, H, W, Cout, kH, kW, dH, dW, padH, padW, dilationH, dilationW,
            group, deformable_group, 0, 0, nullptr, nullptr};
  dcn_use_scratch(dtype, a, columns, columns_bytes);
  if (dH > 0 && dW > 0) {
    a.Ho = (H + 2 * padH - (dilationH * (kH - 1) + 1)) / dH + 1;
    a.Wo = (W + 2 * padW - (dilationW * (kW - 1) + 1)) / dW + 1;
  }
  return launch_dcn(dtype, a, (hipStream_t)stream);
}

int cfen_modulated_deform_conv_forward(int dtype, const void* input, const void* weight, const void* bias, const void* offset,
                                       const void* mask, void* output, int B, int Cin, int H, int W, int Cout, int kernel_h, int kernel_w,
                                       int stride_h, int stride_w, int pad_h, int pad_w, int dilation_h, int dilation_w, int group,
                                       int deformable_group, int with_bias, void* columns, size_t columns_bytes, void* stream) {
  CFEN_CHECK_ARG(mask != nullptr, "modulated_deform_conv: mask is required");
  CFEN_CHECK_ARG(!with_bias || bias, "modulated_deform_conv: with_bias set but bias is null");
  DcnArgs a{input, offset, mask, weight, with_bias ? bias : nullptr, output, B, Cin, H, W, Cout, kernel_h, kernel_w, stride_h, stride_w,
            pad_h, pad_w, dilation_h, dilation_w, group, deformable_group, 0, 0, nullptr, nullptr};
  dcn_use_scratch(dtype, a, columns, columns_bytes);
  if (stride_h > 0 && stride_w > 0) {
    a.Ho = (H + 2 * pad_h - (dilation_h * (kernel_h - 1) + 1)) / stride_h + 1;
    a.Wo = (W + 2 * pad_w - (dilation_w * (kernel_w - 1) + 1)) / stride_w + 1;
  }
  return launch_dcn(dtype, a, (hipStream_t)stream);
}

}  // extern "C"
