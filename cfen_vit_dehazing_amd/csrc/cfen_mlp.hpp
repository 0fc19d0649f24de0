// Arguments of the fused token-MLP kernel (k_mlp.hip).
#pragma once
#include <hip/hip_runtime.h>

struct MlpArgs {
  const void* X;        // [M][D] tokens in
  const void* A; const void* Wp;   // optional prologue: x <- x + Wp A^T-rows, i.e. x[m] += Wp A[m]  (A [M][D] = attention output, Wp [D][D]
                                   // natural k order: out_proj + residual of the attention block, v3:1386)
  void* Y;              // [M][D] tokens out (may alias X), or null when fmap is set
  void* fmap;           // optional: fold the result into an NHWC map (unpatchify fused)
  const float* ln_g;    // LayerNorm before stage a (null = none)
  const float* ln_b;
  const void* W1a; const float* b1a; const void* W2a; const float* b2a;   // stage a: y1 = x + W2a relu(W1a LN(x) + b1a) + b2a
  const void* W1b; const float* b1b; const void* W2b; const float* b2b;   // stage b (optional): y2 = y1 + W2b relu(W1b y1 + b1b) + b2b
  long long M;
  int D, H;
  float eps;
  int mapH, mapW, C, cs, ws, p;   // fold geometry when fmap != null
};

bool cfen_mlp_supported(int D, int H, int dtype);
int cfen_mlp_impl(int dtype, const MlpArgs* a, hipStream_t s);
int cfen_mlp_impl_g(int dtype, int ng, const MlpArgs* a, hipStream_t s);   // ng problems of the same shape, one launch

// Arguments of the fragment-stream token-MLP kernel (k_stream.hip: k_mlp3).  Same chain as MlpArgs; the matrices come as FRAGMENT STREAMS
// (packing.pack_stream_pair / pack_stream_sq): Wa / Wb = [H / 32][2 phases: W1 slice, W2 slice][D / 16 fragments][1 KiB], Wp = [D / 32][D / 16][1 KiB].
struct Mlp3Args {
  const void* X;        // [M][D] tokens in
  const void* A; const void* Wp;   // optional prologue x <- x + Wp A (out_proj + residual, v3:1386)
  void* Y;              // [M][D] tokens out, or null when fmap is set
  void* fmap;           // optional: fold the result into an NHWC map
  const float* ln_g; const float* ln_b;
  const void* Wa; const float* b1a; const float* b2a;   // stage a: y1 = x + W2a relu(W1a LN(x) + b1a) + b2a
  const void* Wb; const float* b1b; const float* b2b;   // stage b (optional): y2 = y1 + W2b relu(W1b y1 + b1b) + b2b
  long long M;
  int D, H;
  float eps;
  int mapH, mapW, C, cs, ws, p;
};
bool cfen_mlp3_supported(int dtype, int D, int H);
int cfen_mlp3_impl_g(int dtype, int ng, const Mlp3Args* a, hipStream_t s);
