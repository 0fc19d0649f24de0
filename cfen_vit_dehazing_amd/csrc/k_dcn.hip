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
#include "cfen_common.hpp"

namespace {

struct DcnArgs {
  const void* im; const void* offset; const void* mask; const void* weight; const void* bias; void* out;
  int B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, group, dg, Ho, Wo;
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
  if (dtype == 1)
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

}  // namespace

extern "C" {

int cfen_deform_conv_forward(int dtype, const void* input, const void* weight, const void* offset, void* output, int B, int Cin, int H,
                             int W, int Cout, int kW, int kH, int dW, int dH, int padW, int padH, int dilationW, int dilationH, int group,
                             int deformable_group, int im2col_step, void* stream) {
  CFEN_CHECK_ARG(im2col_step > 0 && B % (im2col_step < B ? im2col_step : B) == 0, "im2col step must divide batchsize");
  DcnArgs a{input, offset, nullptr, weight, nullptr, output, B, Cin, H, W, Cout, kH, kW, dH, dW, padH, padW, dilationH, dilationW,
            group, deformable_group, 0, 0};
  if (dH > 0 && dW > 0) {
    a.Ho = (H + 2 * padH - (dilationH * (kH - 1) + 1)) / dH + 1;
    a.Wo = (W + 2 * padW - (dilationW * (kW - 1) + 1)) / dW + 1;
  }
  return launch_dcn(dtype, a, (hipStream_t)stream);
}

int cfen_modulated_deform_conv_forward(int dtype, const void* input, const void* weight, const void* bias, const void* offset,
                                       const void* mask, void* output, int B, int Cin, int H, int W, int Cout, int kernel_h, int kernel_w,
                                       int stride_h, int stride_w, int pad_h, int pad_w, int dilation_h, int dilation_w, int group,
                                       int deformable_group, int with_bias, void* stream) {
  CFEN_CHECK_ARG(mask != nullptr, "modulated_deform_conv: mask is required");
  CFEN_CHECK_ARG(!with_bias || bias, "modulated_deform_conv: with_bias set but bias is null");
  DcnArgs a{input, offset, mask, weight, with_bias ? bias : nullptr, output, B, Cin, H, W, Cout, kernel_h, kernel_w, stride_h, stride_w,
            pad_h, pad_w, dilation_h, dilation_w, group, deformable_group, 0, 0};
  if (stride_h > 0 && stride_w > 0) {
    a.Ho = (H + 2 * pad_h - (dilation_h * (kernel_h - 1) + 1)) / stride_h + 1;
    a.Wo = (W + 2 * pad_w - (dilation_w * (kernel_w - 1) + 1)) / stride_w + 1;
  }
  return launch_dcn(dtype, a, (hipStream_t)stream);
}

}  // extern "C"
