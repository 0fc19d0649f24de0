// Token GEMM with fused epilogue:  Y[m][n] = act(sum_k X[m][k] W[n][k] + bias[n]) + R[m][n] + P[m % period][n]
//
// Replaces the reference's nn.Linear calls inside LViT/GViT (linear_encoding v3:1143, in_proj /
// out_proj of nn.MultiheadAttention v3:1364, linear1/linear2 v3:1388, mlp_head v3:1173) together with
// the elementwise ops that follow them (bias, ReLU, residual add, learned position add v3:1166).
//
// Both operands are k-contiguous ("NT"), which is exactly what the 16-byte MFMA fragments want.
// The weight tile is the MFMA A operand (rows = n) and the token tile the B operand (cols = m), so
// each lane ends up with 4 CONSECUTIVE n for one token m: the epilogue reads/writes 8-byte (fp16) or
// 16-byte (fp32) vectors along the row instead of 2-byte scatters.
//
// Block tile 128 tokens x 96 features, 4 waves as 2(n) x 2(m), each 48 x 64 = 3 x 4 MFMA tiles.
// K is staged through LDS in 128-byte row slices (+16 B pad against bank conflicts), double
// buffered with register prefetch: one barrier per K step.
#include "cfen_common.hpp"

namespace {

template <typename T> struct GemmArgs {
  const T* X; const T* W; const float* bias; const T* R; const T* P; T* Y;
  int M, N, K, ldx, ldw, ldr, ldy, period, relu;
};

constexpr int G_BN = 96, G_BM = 128, G_ROWS = G_BN + G_BM;
constexpr int G_BKB = 128;             // bytes of K per row per stage
constexpr int G_ROWB = G_BKB + 16;     // padded LDS row (a 32-byte pad would be conflict-free but overflows 64 KB of static LDS)
constexpr int G_PIECES = G_BKB / 16;   // 16-byte pieces per row
constexpr int G_LOADS = G_ROWS * G_PIECES / 256;   // 7 pieces per thread

template <typename T>
__global__ __launch_bounds__(256) void k_gemm_nt(GemmArgs<T> a) {
  constexpr int EPL = Mma<T>::EPL;
  constexpr int BK = G_BKB / (int)sizeof(T);
  constexpr int NCH = BK / Mma<T>::KC;   // fragment chunks per stage (2)
  typedef typename Mma<T>::frag frag;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][G_ROWS * G_ROWB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, h = lane >> 4;
  const int n0 = blockIdx.x * G_BN, m0 = blockIdx.y * G_BM;
  const int wn = wave & 1, wm = wave >> 1;

  // per-thread staging assignment
  const T* gptr[G_LOADS];
  int koff[G_LOADS], loff[G_LOADS];
#pragma unroll
  for (int i = 0; i < G_LOADS; ++i) {
    int id = tid + i * 256;
    int row = id / G_PIECES, pc = id % G_PIECES;
    if (row < G_BN) {
      int n = min(n0 + row, a.N - 1);
      gptr[i] = a.W + (size_t)n * a.ldw;
    } else {
      int m = min(m0 + row - G_BN, a.M - 1);
      gptr[i] = a.X + (size_t)m * a.ldx;
    }
    koff[i] = pc * EPL;
    loff[i] = row * G_ROWB + pc * 16;
  }

  floatx4 acc[3][4];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

  const int nk = (a.K + BK - 1) / BK;
  frag stage[G_LOADS];
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < G_LOADS; ++i) {
      int k = kt * BK + koff[i];
      stage[i] = (k < a.K) ? load_frag<T>(gptr[i] + k) : Mma<T>::zero();
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < G_LOADS; ++i) *reinterpret_cast<frag*>(&lds[buf][loff[i]]) = stage[i];
  };

  gload(0);
  lstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      frag af[3], bf[4];
#pragma unroll
      for (int i = 0; i < 3; ++i)
        af[i] = *reinterpret_cast<const frag*>(&lds[buf][(wn * 48 + i * 16 + r16) * G_ROWB + c * 64 + h * 16]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        bf[j] = *reinterpret_cast<const frag*>(&lds[buf][(G_BN + wm * 64 + j * 16 + r16) * G_ROWB + c * 64 + h * 16]);
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = Mma<T>::mma(af[i], bf[j], acc[i][j]);
    }
    if (kt + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }

  // epilogue: lane owns Y[m][n..n+3], m = tile col, n = tile row block 4h
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + wm * 64 + j * 16 + r16;
    if (m >= a.M) continue;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int n = n0 + wn * 48 + i * 16 + 4 * h;
      if (n >= a.N) continue;
      floatx4 v = acc[i][j];
      if (a.bias) {
        floatx4 b = *reinterpret_cast<const floatx4*>(a.bias + n);
        v += b;
      }
      if (a.relu) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
      }
      if (a.R) v += load4<T>(a.R + (size_t)m * a.ldr + n);
      if (a.P) v += load4<T>(a.P + (size_t)(m % a.period) * a.N + n);
      store4<T>(a.Y + (size_t)m * a.ldy + n, v);
    }
  }
}

// Small-M variant (GViT: 128..2048 tokens per batch against weight matrices of up to 6144 x 1536): the
// problem is weight-bandwidth bound, so the grid is cut for parallelism instead of reuse -- one workgroup
// per 16 output features x 64 tokens, its 4 waves split K and reduce through LDS; operands go straight
// from global memory into MFMA fragments (32 contiguous bytes per lane per step), no staging, no barrier
// in the loop.  Same epilogue as k_gemm_nt.
template <typename T>
__global__ __launch_bounds__(256) void k_gemm_skinny(GemmArgs<T> a) {
  constexpr int EPL = Mma<T>::EPL, KS = 2 * Mma<T>::KC;
  typedef typename Mma<T>::frag frag;
  __shared__ floatx4 red[4][4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, h = lane >> 4;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 64;
  const T* wp = a.W + (size_t)min(n0 + r16, a.N - 1) * a.ldw + h * 2 * EPL;
  const T* xp[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) xp[j] = a.X + (size_t)min(m0 + j * 16 + r16, a.M - 1) * a.ldx + h * 2 * EPL;
  floatx4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
  for (int k0 = wave * KS; k0 < a.K; k0 += 4 * KS) {
    const frag a0 = load_frag<T>(wp + k0), a1 = load_frag<T>(wp + k0 + EPL);
    frag b0[4], b1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      b0[j] = load_frag<T>(xp[j] + k0);
      b1[j] = load_frag<T>(xp[j] + k0 + EPL);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[j] = Mma<T>::mma(a0, b0[j], acc[j]);
      acc[j] = Mma<T>::mma(a1, b1[j], acc[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) red[wave][j][lane] = acc[j];
  __syncthreads();
  const int j = wave;                       // wave w finishes token tile w
  floatx4 v = red[0][j][lane] + red[1][j][lane] + red[2][j][lane] + red[3][j][lane];
  const int m = m0 + j * 16 + r16, n = n0 + 4 * h;
  if (m >= a.M || n >= a.N) return;
  if (a.bias) v += *reinterpret_cast<const floatx4*>(a.bias + n);
  if (a.relu) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
  }
  if (a.R) v += load4<T>(a.R + (size_t)m * a.ldr + n);
  if (a.P) v += load4<T>(a.P + (size_t)(m % a.period) * a.N + n);
  store4<T>(a.Y + (size_t)m * a.ldy + n, v);
}

template <typename T>
int launch_gemm(const void* X, int ldx, const void* W, int ldw, const float* bias, const void* R, int ldr,
                const void* P, int period, void* Y, int ldy, int M, int N, int K, int relu, hipStream_t s) {
  constexpr int EPL = Mma<T>::EPL;
  CFEN_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm: empty problem M=%d N=%d K=%d", M, N, K);
  CFEN_CHECK_ARG(N % 4 == 0 && K % EPL == 0, "gemm: N (%d) must be a multiple of 4 and K (%d) of %d", N, K, EPL);
  CFEN_CHECK_ARG(ldx % EPL == 0 && ldw % EPL == 0 && ldy % 4 == 0 && (!R || ldr % 4 == 0), "gemm: misaligned leading dimension");
  CFEN_CHECK_ARG(ldx >= K && ldw >= K && ldy >= N && (!R || ldr >= N), "gemm: leading dimension smaller than row");
  CFEN_CHECK_ARG(cfen_aligned16(X) && cfen_aligned16(W) && cfen_aligned16(Y) && cfen_aligned16(R) && cfen_aligned16(P) &&
                 cfen_aligned16(bias), "gemm: pointers must be 16-byte aligned");
  CFEN_CHECK_ARG(!P || period > 0, "gemm: position table needs a period");
  GemmArgs<T> a{(const T*)X, (const T*)W, bias, (const T*)R, (const T*)P, (T*)Y, M, N, K, ldx, ldw, ldr, ldy, period, relu};
  if (M <= 2048 && K % (2 * Mma<T>::KC) == 0) {
    CFEN_LAUNCH(k_gemm_skinny<T>, dim3((N + 15) / 16, (M + 63) / 64), dim3(256), 0, s, a);
    CFEN_CHECK_LAUNCH("gemm");
    return CFEN_OK;
  }
  dim3 grid((N + G_BN - 1) / G_BN, (M + G_BM - 1) / G_BM);
  CFEN_CHECK_ARG(grid.y <= 65535, "gemm: M too large for one launch");
  CFEN_LAUNCH(k_gemm_nt<T>, grid, dim3(256), 0, s, a);
  CFEN_CHECK_LAUNCH("gemm");
  return CFEN_OK;
}

}  // namespace

int cfen_gemm_impl(int dtype, const void* X, int ldx, const void* W, int ldw, const float* bias, const void* R, int ldr,
                   const void* P, int period, void* Y, int ldy, int M, int N, int K, int relu, hipStream_t s) {
  if (dtype == 1) return launch_gemm<half_t>(X, ldx, W, ldw, bias, R, ldr, P, period, Y, ldy, M, N, K, relu, s);
  if (dtype == 0) return launch_gemm<float>(X, ldx, W, ldw, bias, R, ldr, P, period, Y, ldy, M, N, K, relu, s);
  cfen_set_error("gemm: unknown dtype %d", dtype);
  return CFEN_ERR_ARG;
}
