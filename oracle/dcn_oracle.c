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
#include <stdlib.h>

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

/* ---------------------------------------------------------------------------------------------------------------------------
 * Backward direction.  Plain-C restatement of
 *   get_gradient_weight / get_coordinate_weight          dcn/src/deform_conv_cuda_kernel.cu:116-187
 *   deformable_col2im_gpu_kernel                          .cu:278-328   (grad_input: the four bilinear corners, atomically added)
 *   deformable_col2im_coord_gpu_kernel                    .cu:359-421   (grad_offset)
 *   modulated_deformable_col2im(_coord)_gpu_kernel        .cu:634-766   (+ mask on both, grad_mask)
 *   deform_conv_backward_input_cuda / _parameters_cuda    deform_conv_cuda.cpp:260-484  (columns = W^T grad_out; grad_W += scale grad_out col^T)
 *   modulated_deform_conv_cuda_backward                   deform_conv_cuda.cpp:566-679  (+ grad_bias = sum of grad_out)
 * One (image, output pixel, channel, tap) at a time, double accumulation.  grad_input / grad_offset / grad_mask are overwritten,
 * grad_weight / grad_bias are ACCUMULATED into (the reference's addmm_ with beta = 1 into the caller's zero-initialised tensors).
 *
 * PINNING: as for the forward there is nothing of the reference to run; tests/test_dcn_oracle.py pins this against (1) torch autograd
 * of F.conv2d at zero offsets (grad_input, grad_weight, grad_bias), (2) central finite differences of dcn_oracle_forward in the offsets
 * and the mask (the analytic restatement of the reference's coordinate weights must agree with the numeric derivative of the restated
 * forward wherever the sample is differentiable), (3) linearity in grad_output.  Against the reference itself: parity unpinned. */
static void corner_weights(float h, float w, int H, int W, int* hl, int* wl, float wt[4], int ok[4]) {
  /* get_gradient_weight (.cu:116-142) for the four corners of (h, w): (hl, wl), (hl, wl+1), (hl+1, wl), (hl+1, wl+1); a corner outside
   * the image receives nothing (col2im's bounds test, .cu:316-317) */
  const int h_low = (int)floorf(h), w_low = (int)floorf(w);
  const float lh = h - h_low, lw = w - w_low;
  *hl = h_low; *wl = w_low;
  wt[0] = (1 - lh) * (1 - lw); wt[1] = (1 - lh) * lw; wt[2] = lh * (1 - lw); wt[3] = lh * lw;
  ok[0] = h_low >= 0 && h_low < H && w_low >= 0 && w_low < W;
  ok[1] = h_low >= 0 && h_low < H && w_low + 1 >= 0 && w_low + 1 < W;
  ok[2] = h_low + 1 >= 0 && h_low + 1 < H && w_low >= 0 && w_low < W;
  ok[3] = h_low + 1 >= 0 && h_low + 1 < H && w_low + 1 >= 0 && w_low + 1 < W;
}

int dcn_oracle_backward(const float* im, const float* offset, const float* mask, const float* weight, const float* gout, float* gin,
                        float* goff, float* gmask, float* gweight, float* gbias, float scale, int B, int C, int H, int W, int Cout, int kh,
                        int kw, int sh, int sw, int ph, int pw, int dh, int dw, int group, int dg) {
  if (kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || group <= 0 || dg <= 0) return -1;
  if (C % group || Cout % group || C % dg) return -1;
  const int Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
  const int Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
  if (Ho < 1 || Wo < 1) return -1;
  const int Cg = C / group, Cog = Cout / group, cpdg = C / dg, kk = kh * kw;
  const size_t HWo = (size_t)Ho * Wo, HW = (size_t)H * W;
  if (gin) for (size_t i = 0; i < (size_t)B * C * HW; ++i) gin[i] = 0.f;
  if (goff) for (size_t i = 0; i < (size_t)B * dg * 2 * kk * HWo; ++i) goff[i] = 0.f;
  if (gmask) for (size_t i = 0; i < (size_t)B * dg * kk * HWo; ++i) gmask[i] = 0.f;
  double* gw = NULL;
  if (gweight) {
    gw = (double*)calloc((size_t)Cout * Cg * kk, sizeof(double));
    if (!gw) return -2;
  }
  for (int b = 0; b < B; ++b)
    for (int ho = 0; ho < Ho; ++ho)
      for (int wo = 0; wo < Wo; ++wo) {
        const size_t px = (size_t)ho * Wo + wo;
        for (int cim = 0; cim < C; ++cim) {
          const int g = cim / Cg, c = cim % Cg, dgi = cim / cpdg;
          const float* imp = im + ((size_t)b * C + cim) * HW;
          const float* offp = offset + ((size_t)b * dg + dgi) * 2 * kk * HWo;
          const float* mp = mask ? mask + ((size_t)b * dg + dgi) * kk * HWo : NULL;
          for (int i = 0; i < kh; ++i)
            for (int j = 0; j < kw; ++j) {
              const int ij = i * kw + j;
              const float h_im = (float)(ho * sh - ph + i * dh) + offp[(size_t)(2 * ij) * HWo + px];
              const float w_im = (float)(wo * sw - pw + j * dw) + offp[(size_t)(2 * ij + 1) * HWo + px];
              const float m = mp ? mp[(size_t)ij * HWo + px] : 1.f;
              const int inside = h_im > -1 && w_im > -1 && h_im < H && w_im < W;
              /* column gradient: (W^T grad_out)[c, ij] at this pixel (deform_conv_cuda.cpp:332-337) */
              double gcol = 0.0;
              for (int co = 0; co < Cog; ++co)
                gcol += (double)weight[(((size_t)(g * Cog + co)) * Cg + c) * kk + ij] * (double)gout[((size_t)b * Cout + g * Cog + co) * HWo + px];
              const float val = inside ? bilinear(imp, H, W, h_im, w_im) : 0.f;
              if (gweight)   /* grad_W[o, c, ij] += grad_out[o] * column value (masked sample)   (deform_conv_cuda.cpp:452-457) */
                for (int co = 0; co < Cog; ++co)
                  gw[(((size_t)(g * Cog + co)) * Cg + c) * kk + ij] += (double)gout[((size_t)b * Cout + g * Cog + co) * HWo + px] * (double)(val * m);
              if (!inside) continue;   /* .cu:122, 155, 400-403: outside (-1, H) x (-1, W) nothing flows back */
              int hl, wl, ok[4];
              float wt[4];
              corner_weights(h_im, w_im, H, W, &hl, &wl, wt, ok);
              const float v[4] = {ok[0] ? imp[hl * W + wl] : 0.f, ok[1] ? imp[hl * W + wl + 1] : 0.f, ok[2] ? imp[(hl + 1) * W + wl] : 0.f,
                                  ok[3] ? imp[(hl + 1) * W + wl + 1] : 0.f};
              if (gin) {
                float* gp = gin + ((size_t)b * C + cim) * HW;
                const float t = (float)gcol * m;
                if (ok[0]) gp[hl * W + wl] += wt[0] * t;
                if (ok[1]) gp[hl * W + wl + 1] += wt[1] * t;
                if (ok[2]) gp[(hl + 1) * W + wl] += wt[2] * t;
                if (ok[3]) gp[(hl + 1) * W + wl + 1] += wt[3] * t;
              }
              if (goff) {
                /* get_coordinate_weight (.cu:144-187): d sample / d h and d sample / d w */
                const float lh = h_im - hl, lw = w_im - wl;
                const float dH = -(1 - lw) * v[0] - lw * v[1] + (1 - lw) * v[2] + lw * v[3];
                const float dW = -(1 - lh) * v[0] + (1 - lh) * v[1] - lh * v[2] + lh * v[3];
                float* go = goff + ((size_t)b * dg + dgi) * 2 * kk * HWo;
                go[(size_t)(2 * ij) * HWo + px] += (float)(gcol * dH * m);
                go[(size_t)(2 * ij + 1) * HWo + px] += (float)(gcol * dW * m);
              }
              if (gmask) gmask[(((size_t)b * dg + dgi) * kk + ij) * HWo + px] += (float)(gcol * val);
            }
        }
      }
  if (gweight) {
    for (size_t i = 0; i < (size_t)Cout * Cg * kk; ++i) gweight[i] += scale * (float)gw[i];
    free(gw);
  }
  if (gbias)
    for (int co = 0; co < Cout; ++co) {
      double s = 0.0;
      for (int b = 0; b < B; ++b)
        for (size_t p = 0; p < HWo; ++p) s += gout[((size_t)b * Cout + co) * HWo + p];
      gbias[co] += (float)s;
    }
  return 0;
}
