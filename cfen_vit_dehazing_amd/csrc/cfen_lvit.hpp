// Arguments of the one-workgroup-per-window LViT block kernel (k_lvit.hip).
#pragma once
#include <hip/hip_runtime.h>

struct LvitArgs {
  const void* fmap; void* out;            // NHWC maps (B, H, W), channel strides cs_in / cs_out, C real channels
  int B, H, W, C, cs_in, cs_out, ws, p;   // window edge ws (32), patch p (2): 256 tokens of D = p*p*C = 96 per window
  const void* Ws;                         // every weight matrix of the block as ONE stream of 1 KiB MFMA A fragments in consumption order
                                          // (packing.pack_lvit_window): linear_encoding, K / V rows, per head W_q tiles + out_proj slice,
                                          // linear1 / linear2 and mlp_head in 32-unit hidden slices; k axes kperm32
  const float* be; const void* pos;       // linear_encoding bias, position table [S][D]
  const float* ln1_g; const float* ln1_b;
  const float* ln2_g; const float* ln2_b;
  const float* b1a; const float* b2a;     // biases of linear1 / linear2 and of mlp_head (as MlpArgs)
  const float* b1b; const float* b2b;
  int Hm;                                 // hidden width of both MLPs
  float eps, scale_log2;                  // LayerNorm eps; log2(e) / sqrt(head_dim)
};

bool cfen_lvit_window_supported(int dtype, int D, int heads, int S, int hidden);
int cfen_lvit_window_impl_g(int dtype, int ng, const LvitArgs* a, hipStream_t s);
int& cfen_tune_lvit_shape();
