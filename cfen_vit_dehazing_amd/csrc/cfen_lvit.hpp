// Arguments of the one-workgroup-per-window LViT block kernel (k_lvit.hip).
#pragma once
#include <hip/hip_runtime.h>

struct LvitArgs {
  const void* fmap; void* out;            // NHWC maps (B, H, W), channel strides cs_in / cs_out, C real channels
  int B, H, W, C, cs_in, cs_out, ws, p;   // window edge ws (32), patch p (2): 256 tokens of D = p*p*C = 96 per window
  const void* We; const float* be; const void* pos;    // linear_encoding [D][D] (k axis kperm32), bias, position table [S][D]
  const float* ln1_g; const float* ln1_b;
  const void* Wkv;                        // [2D][D]: K rows of all heads, then V rows (natural feature order), k axis kperm32
  const void* Wq;                         // [heads][32][D]: per head two 16-row tiles laid out so that the accumulator pair packs into
                                          // natural d order (packing.lvit_q_rows), rows of d >= 24 zero; k axis kperm32
  const void* Wp;                         // [heads][D][32]: out_proj columns of the head, slot s <- d = kperm32(s), slots of d >= 24 zero
  const float* ln2_g; const float* ln2_b;
  const void* W1a; const float* b1a; const void* W2a; const float* b2a;   // as MlpArgs (k axis kperm32)
  const void* W1b; const float* b1b; const void* W2b; const float* b2b;
  int Hm;                                 // hidden width of both MLPs
  float eps, scale_log2;                  // LayerNorm eps; log2(e) / sqrt(head_dim)
};

bool cfen_lvit_window_supported(int dtype, int D, int heads, int S, int hidden);
int cfen_lvit_window_impl_g(int dtype, int ng, const LvitArgs* a, hipStream_t s);
int& cfen_tune_lvit_shape();
