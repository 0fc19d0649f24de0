/* CPU oracle for the deformable-convolution forward.   *** TEST INFRASTRUCTURE ONLY ***
 *
 * Plain-C restatement of the reference's CUDA operator, forward direction:
 *   deformable_im2col_bilinear              dcn/src/deform_conv_cuda_kernel.cu:83-114
 *   deformable_im2col_gpu_kernel            dcn/src/deform_conv_cuda_kernel.cu:189-242
 *   modulated_deformable_im2col_gpu_kernel  dcn/src/deform_conv_cuda_kernel.cu:569-632
 *   deform_conv_forward_cuda                dcn/src/deform_conv_cuda.cpp:151-258   (im2col + per-group GEMM)
 *   modulated_deform_conv_cuda_forward      dcn/src/deform_conv_cuda.cpp:486-564   (+ mask, + bias)
 * Sampling is done in float like the reference's scalar_t=float instantiation; the channel/tap sum is
 * accumulated in double (the reference sums in a BLAS GEMM whose order is unspecified).
 *
 * PINNING: the reference extension cannot be built here (CUDA-only sources, THC headers, no nvcc) and
 * the reference ships no DCN tests or vectors, so this oracle is pinned only by the known-answer
 * properties that follow from the kernel code (tests/test_dcn_oracle.py): zero offsets == conv2d,
 * integer offsets == shifted-tap conv, mask == 1 == DCNv1 + bias, DeformConvPack at init == conv2d.
 * Against the reference implementation itself: parity unpinned.
 *
 * Only tests/ and __graft_entry__.smoke() may load this library.
 */
#include <math.h>
#include <stddef.h>

#define DCN_ORACLE_MAX_K 8192   /* channels per group x taps */

static float bilinear(const float* im, int H, int W, float h, float w) {
  int h_low = (int)floorf(h), w_low = (int)floorf(w);
  int h_high = h_low + 1, w_high = w_low + 1;
  float lh = h - h_low, lw = w - w_low, hh = 1 - lh, hw = 1 - lw;
  float v1 = 0, v2 = 0, v3 = 0, v4 = 0;
  if (h_low >= 0 && w_low >= 0) v1 = im[h_low * W + w_low];
  if (h_low >= 0 && w_high <= W - 1) v2 = im[h_low * W + w_high];
  if (h_high <= H - 1 && w_low >= 0) v3 = im[h_high * W + w_low];
  if (h_high <= H - 1 && w_high <= W - 1) v4 = im[h_high * W + w_high];
  float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
  return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
}

/* mask == NULL -> DCNv1; bias == NULL -> no bias.  All tensors NCHW contiguous float.
 * returns 0, or -1 on an invalid shape (mirrors shape_check, deform_conv_cuda.cpp:61-149). */
int dcn_oracle_forward(const float* im, const float* offset, const float* mask, const float* weight, const float* bias, float* out,
                       int B, int C, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int group,
                       int dg) {
  if (kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || group <= 0 || dg <= 0) return -1;
  if (C % group || Cout % group || C % dg) return -1;
  const int Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
  const int Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
  if (Ho < 1 || Wo < 1 || H < kh || W < kw) return -1;
  const int Cg = C / group, Cog = Cout / group, cpdg = C / dg, kk = kh * kw;
  const size_t HWo = (size_t)Ho * Wo;
  const int Kg = Cg * kk;
  if (Kg > DCN_ORACLE_MAX_K) return -1;
  /* One output pixel at a time: its column (the reference's deformable_im2col, one sample per (channel, tap)) is built once
   * and contracted with every output channel's weights (the reference's addmm_), k ascending, in double.  Rows of the
   * output are independent: the optional OpenMP loop changes nothing in the arithmetic. */
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int ho = 0; ho < Ho; ++ho) {
      float col[DCN_ORACLE_MAX_K];
      for (int g = 0; g < group; ++g)
        for (int wo = 0; wo < Wo; ++wo) {
          for (int c = 0; c < Cg; ++c) {
            const int cim = g * Cg + c, dgi = cim / cpdg;
            const float* imp = im + ((size_t)b * C + cim) * H * W;
            const float* offp = offset + ((size_t)b * dg + dgi) * 2 * kk * HWo;
            const float* mp = mask ? mask + ((size_t)b * dg + dgi) * kk * HWo : NULL;
            for (int i = 0; i < kh; ++i)
              for (int j = 0; j < kw; ++j) {
                const int ij = i * kw + j;
                const float oh = offp[(size_t)(2 * ij) * HWo + (size_t)ho * Wo + wo];
                const float ow = offp[(size_t)(2 * ij + 1) * HWo + (size_t)ho * Wo + wo];
                const float h_im = (float)(ho * sh - ph + i * dh) + oh;
                const float w_im = (float)(wo * sw - pw + j * dw) + ow;
                float val = 0.f;
                if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) val = bilinear(imp, H, W, h_im, w_im);
                if (mp) val *= mp[(size_t)ij * HWo + (size_t)ho * Wo + wo];
                col[c * kk + ij] = val;
              }
          }
          for (int co = 0; co < Cog; ++co) {
            const float* wp = weight + ((size_t)(g * Cog + co)) * Kg;
            double acc = 0.0;
            for (int k = 0; k < Kg; ++k) acc += (double)wp[k] * (double)col[k];
            if (bias) acc += bias[g * Cog + co];
            out[(((size_t)b * Cout + g * Cog + co) * Ho + ho) * Wo + wo] = (float)acc;
          }
        }
    }
  return 0;
}
