// LViT front half in one launch (levels 1 and 2, D = 96 / 192):
//     x   = patch tokens gathered from the NHWC map               (Crop2x2 + F.unfold, v3:1025-1056, 1140)
//     y   = W_e x + b_e + x + pos[token % S]                      (linear_encoding + residual + position, v3:1143, 1166)   -> X1
//     qkv = W_qkv LayerNorm(y)                                    (norm1 + in_proj of nn.MultiheadAttention, v3:1364-1371) -> QKV
// replacing patchify + embedding GEMM + LayerNorm + qkv GEMM and their three intermediate tensors.  Same structure as
// k_mlp.hip: a wave owns TM*16 tokens whose features live in fp32 MFMA accumulators (rows = features, columns = tokens);
// an accumulator tile pair IS the B operand of the next GEMM (weights' k axis pre-permuted on the host, packing.kperm32),
// LayerNorm reduces over a lane's registers + two lane swaps.  The weights (18 + 55 KB at D = 96) are read as MFMA A
// fragments straight from L1/L2 -- every wave of the chip reads the same few kilobytes.
#include <type_traits>
#include "cfen_common.hpp"
#include "cfen_internal.hpp"

namespace {

template <typename T> struct PackB;
template <> struct PackB<half_t> {
  static constexpr int NPC = 2;   // accumulator n-tiles per K chunk
  static CFEN_DEV half8 make(const floatx4* t) {
    half8 f = {(half_t)t[0][0], (half_t)t[0][1], (half_t)t[0][2], (half_t)t[0][3],
               (half_t)t[1][0], (half_t)t[1][1], (half_t)t[1][2], (half_t)t[1][3]};
    return f;
  }
};
template <> struct PackB<float> {
  static constexpr int NPC = 1;
  static CFEN_DEV floatx4 make(const floatx4* t) { return t[0]; }
};


// element offset of feature f (a multiple of 4) of a qkv row inside its (window, head) triple for the head-major layout:
// [(window * heads + head) * 3 + part][S][dh], part = f / D; relative to the window's first element, without the token term
// (head_dim is 24 wherever this layout is used -- k_attention_hm is built for it -- so every division here is by a constant)
constexpr int HM_DH = 24;
template <int D>
CFEN_DEV int hm_feature_off(int f, int S) {
  const int part = f / D, fd = f - part * D, hd = fd / HM_DH, d = fd - hd * HM_DH;
  return ((hd * 3 + part) * S) * HM_DH + d;
}

template <typename T, int ND, int TM, int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_embed_qkv(Grouped<CfenEmbedQkvArgs> ga) {
  const CfenEmbedQkvArgs& a = ga.g[blockIdx.z];
  constexpr int KC = Mma<T>::KC, EPL = Mma<T>::EPL;
  constexpr int NPC = PackB<T>::NPC;
  constexpr int D = ND * 16;
  constexpr int NCH = ND / NPC;
  typedef typename Mma<T>::frag frag;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, h = lane >> 4;
  const long long tok0 = ((long long)blockIdx.x * NW + wave) * (TM * 16);
  if (tok0 >= a.M) return;

  // ---- gather x^T into accumulator layout: acc[i][j][r] = tok[token j*16 + r16][feature i*16 + 4h + r] ----
  const int tw = a.ws / a.p, S = tw * tw, nwx = a.W / a.ws, nwy = a.H / a.ws;
  floatx4 acc[ND][TM];
  long long tk[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    long long t = tok0 + j * 16 + r16;
    if (t >= a.M) t = a.M - 1;
    tk[j] = t;
    const int tt = (int)(t % S);
    const long long wi = t / S;
    const int wx = (int)(wi % nwx), wy = (int)((wi / nwx) % nwy);
    const long long b = wi / ((long long)nwx * nwy);
    const int y0 = wy * a.ws + (tt / tw) * a.p, x0 = wx * a.ws + (tt % tw) * a.p;
    const T* pix = (const T*)a.fmap + ((b * a.H + y0) * a.W + x0) * a.cs;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int f = i * 16 + 4 * h;
      const int ij = f / a.C, c = f - ij * a.C;
      acc[i][j] = load4<T>(pix + ((ij / a.p) * a.W + (ij % a.p)) * a.cs + c);
    }
  }
  frag xb[NCH][TM];
#pragma unroll
  for (int j = 0; j < TM; ++j)
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      floatx4 t[NPC];
#pragma unroll
      for (int u = 0; u < NPC; ++u) t[u] = acc[c * NPC + u][j];
      xb[c][j] = PackB<T>::make(t);
    }

  // ---- y = W_e x + b_e + x + pos ----
  const T* We = (const T*)a.We + (size_t)r16 * D + h * EPL;
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    const floatx4 bb = *reinterpret_cast<const floatx4*>(a.be + i * 16 + 4 * h);
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] += bb + load4<T>((const T*)a.pos + (size_t)(tk[j] % S) * D + i * 16 + 4 * h);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const frag af = load_frag<T>(We + (size_t)i * 16 * D + c * KC);
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] = Mma<T>::mma(af, xb[c][j], acc[i][j]);
    }
  }
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    if (tok0 + j * 16 + r16 >= a.M) continue;
    T* yp = (T*)a.X1 + tk[j] * D + 4 * h;
#pragma unroll
    for (int i = 0; i < ND; ++i) store4<T>(yp + i * 16, acc[i][j]);
  }

  // ---- LayerNorm(y) -> B fragments (two passes over registers, statistics in fp32) ----
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ND; ++i) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    s = col_sum(s);
    const float mean = s * (1.f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < ND; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = acc[i][j][r] - mean;
        q += d * d;
      }
    q = col_sum(q);
    const float rstd = rsqrtf(q * (1.f / D) + a.eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      floatx4 t[NPC];
#pragma unroll
      for (int u = 0; u < NPC; ++u) {
        const int i = c * NPC + u;
        const floatx4 g = *reinterpret_cast<const floatx4*>(a.ln_g + i * 16 + 4 * h);
        const floatx4 b = *reinterpret_cast<const floatx4*>(a.ln_b + i * 16 + 4 * h);
        t[u] = (acc[i][j] - mean) * rstd * g + b;
      }
      xb[c][j] = PackB<T>::make(t);
    }
  }

  // element offset of each token's qkv row: row-major [M][3D], or the token term of the head-major layout
  long long qrow[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j)
    qrow[j] = a.hm_heads ? (tk[j] / S) * (3LL * S * D) + (tk[j] % S) * HM_DH : tk[j] * (3LL * D);
  // ---- qkv = W_qkv LN(y): 3D output features, one 16-feature tile at a time, straight to HBM ----
  const T* Wq = (const T*)a.Wqkv + (size_t)r16 * D + h * EPL;
#pragma unroll 2
  for (int i = 0; i < 3 * ND; ++i) {
    floatx4 q[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) q[j] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const frag af = load_frag<T>(Wq + (size_t)i * 16 * D + c * KC);
#pragma unroll
      for (int j = 0; j < TM; ++j) q[j] = Mma<T>::mma(af, xb[c][j], q[j]);
    }
    const long long fo = a.hm_heads ? hm_feature_off<D>(i * 16 + 4 * h, S) : i * 16 + 4 * h;
#pragma unroll
    for (int j = 0; j < TM; ++j)
      if (tok0 + j * 16 + r16 < a.M) store4<T>((T*)a.QKV + qrow[j] + fo, q[j]);
  }
}

// Variant with the weights staged through LDS: 8 waves (TM*128 tokens) share each 16-feature weight tile instead of every
// wave pulling all of W_e / W_qkv from L2 on its own -- at D = 192 that is 288 KB per 32 tokens.  NG tiles per stage,
// double buffered, the next stage's global loads in flight during the MFMAs (as k_mlp.hip).
template <typename T, int ND, int TM, int NG>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_embed_qkv_lds(Grouped<CfenEmbedQkvArgs> ga) {
  const CfenEmbedQkvArgs& a = ga.g[blockIdx.z];
  constexpr int KC = Mma<T>::KC, EPL = Mma<T>::EPL, SZ = (int)sizeof(T);
  constexpr int NPC = PackB<T>::NPC, NW = 8, NT = NW * 64;
  constexpr int D = ND * 16, NCH = ND / NPC;
  constexpr int ROWB = D * SZ + 32;                    // stride = 32 (mod 64) bytes: conflict-free ds_read_b128
  constexpr int STAGE = NG * 16 * ROWB;
  constexpr int PPR = D * SZ / 16, NP = NG * 16 * PPR; // 16-byte pieces per row / per stage
  constexpr int NPF = (NP + NT - 1) / NT;
  constexpr int NES = ND / NG, NQS = 3 * ND / NG;      // embedding / qkv stages
  static_assert(ND % NG == 0 && 2 * STAGE <= 65536, "stage geometry");
  typedef typename Mma<T>::frag frag;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, h = lane >> 4;
  const long long tok0 = ((long long)blockIdx.x * NW + wave) * (TM * 16);

  frag pf[NPF];
  auto prefetch = [&](int st) {   // stage st: rows [st*NG*16, +NG*16) of W_e, then of W_qkv
    const T* W = st < NES ? (const T*)a.We + (size_t)st * NG * 16 * D : (const T*)a.Wqkv + (size_t)(st - NES) * NG * 16 * D;
#pragma unroll
    for (int u = 0; u < NPF; ++u) {
      const int id = tid + u * NT;
      if (id < NP) pf[u] = load_frag<T>(W + (size_t)(id / PPR) * D + (id % PPR) * EPL);
    }
  };
  auto commit = [&](unsigned char* buf) {
#pragma unroll
    for (int u = 0; u < NPF; ++u) {
      const int id = tid + u * NT;
      if (id < NP) *reinterpret_cast<frag*>(buf + (id / PPR) * ROWB + (id % PPR) * 16) = pf[u];
    }
  };
  prefetch(0);

  // ---- gather x^T into accumulator layout (tokens past M are clamped: the whole workgroup keeps hitting the barriers) ----
  const int tw = a.ws / a.p, S = tw * tw, nwx = a.W / a.ws, nwy = a.H / a.ws;
  floatx4 acc[ND][TM];
  long long tk[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    long long t = tok0 + j * 16 + r16;
    if (t >= a.M) t = a.M - 1;
    tk[j] = t;
    const int tt = (int)(t % S);
    const long long wi = t / S;
    const int wx = (int)(wi % nwx), wy = (int)((wi / nwx) % nwy);
    const long long b = wi / ((long long)nwx * nwy);
    const int y0 = wy * a.ws + (tt / tw) * a.p, x0 = wx * a.ws + (tt % tw) * a.p;
    const T* pix = (const T*)a.fmap + ((b * a.H + y0) * a.W + x0) * a.cs;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int f = i * 16 + 4 * h;
      const int ij = f / a.C, c = f - ij * a.C;
      acc[i][j] = load4<T>(pix + ((ij / a.p) * a.W + (ij % a.p)) * a.cs + c);
    }
  }
  frag xb[NCH][TM];
#pragma unroll
  for (int j = 0; j < TM; ++j)
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      floatx4 t[NPC];
#pragma unroll
      for (int u = 0; u < NPC; ++u) t[u] = acc[c * NPC + u][j];
      xb[c][j] = PackB<T>::make(t);
    }
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    const floatx4 bb = *reinterpret_cast<const floatx4*>(a.be + i * 16 + 4 * h);
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] += bb + load4<T>((const T*)a.pos + (size_t)(tk[j] % S) * D + i * 16 + 4 * h);
  }
  commit(lds);
  __syncthreads();

  // ---- y = W_e x + (b_e + x + pos): embedding stages, accumulator indices are compile-time ----
  const unsigned char* ap = lds + r16 * ROWB + h * 16;
#pragma unroll
  for (int st = 0; st < NES; ++st) {
    prefetch(st + 1);                       // NES < NES + NQS: there is always a next stage
#pragma unroll
    for (int u = 0; u < NG; ++u)
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const frag af = *reinterpret_cast<const frag*>(ap + (st & 1) * STAGE + u * 16 * ROWB + c * 64);
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[st * NG + u][j] = Mma<T>::mma(af, xb[c][j], acc[st * NG + u][j]);
      }
    commit(lds + ((st + 1) & 1) * STAGE);
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    if (tok0 + j * 16 + r16 >= a.M) continue;
    T* yp = (T*)a.X1 + tk[j] * D + 4 * h;
#pragma unroll
    for (int i = 0; i < ND; ++i) store4<T>(yp + i * 16, acc[i][j]);
  }
  // ---- LayerNorm(y) -> B fragments ----
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ND; ++i) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    s = col_sum(s);
    const float mean = s * (1.f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < ND; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = acc[i][j][r] - mean;
        q += d * d;
      }
    q = col_sum(q);
    const float rstd = rsqrtf(q * (1.f / D) + a.eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      floatx4 t[NPC];
#pragma unroll
      for (int u = 0; u < NPC; ++u) {
        const int i = c * NPC + u;
        const floatx4 g = *reinterpret_cast<const floatx4*>(a.ln_g + i * 16 + 4 * h);
        const floatx4 b = *reinterpret_cast<const floatx4*>(a.ln_b + i * 16 + 4 * h);
        t[u] = (acc[i][j] - mean) * rstd * g + b;
      }
      xb[c][j] = PackB<T>::make(t);
    }
  }
  // element offset of each token's qkv row: row-major [M][3D], or the token term of the head-major layout
  long long qrow[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j)
    qrow[j] = a.hm_heads ? (tk[j] / S) * (3LL * S * D) + (tk[j] % S) * HM_DH : tk[j] * (3LL * D);
  // ---- qkv stages: every tile goes straight to HBM ----
#pragma unroll 1
  for (int sq = 0; sq < NQS; ++sq) {
    const int st = NES + sq;
    if (sq + 1 < NQS) prefetch(st + 1);
#pragma unroll
    for (int u = 0; u < NG; ++u) {
      floatx4 q[TM];
#pragma unroll
      for (int j = 0; j < TM; ++j) q[j] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const frag af = *reinterpret_cast<const frag*>(ap + (st & 1) * STAGE + u * 16 * ROWB + c * 64);
#pragma unroll
        for (int j = 0; j < TM; ++j) q[j] = Mma<T>::mma(af, xb[c][j], q[j]);
      }
      {
        const long long fo = a.hm_heads ? hm_feature_off<D>((sq * NG + u) * 16 + 4 * h, S) : (sq * NG + u) * 16 + 4 * h;
#pragma unroll
        for (int j = 0; j < TM; ++j)
          if (tok0 + j * 16 + r16 < a.M) store4<T>((T*)a.QKV + qrow[j] + fo, q[j]);
      }
    }
    if (sq + 1 < NQS) commit(lds + ((st + 1) & 1) * STAGE);
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// k_embed_qkv2 (fp16): the same front half with the weight stream rebuilt like k_mlp2 (k_mlp.hip).  rocprofv3 showed the waves of
// both variants above parked on s_waitcnt / barriers for 54-58 % of their cycles: the D = 96 kernel loads every weight fragment from
// L2 right before the MFMAs that use it, the LDS variant stages through registers with two barriers per 64 weight rows.  Here the
// 4 D rows of [W_e ; W_qkv] stream through a two-stage LDS ring by LDS-DMA, RS rows per chunk, one raw s_barrier per chunk with the
// next chunk's DMA in flight; fragments are read in groups of three with the next group already loading (software pipeline pinned
// with sched_barrier); rows padded to a 32 (mod 64)-byte pitch.  The qkv tiles are stored while the next chunk's DMA is in flight, so
// the landing wait is a COUNTED vmcnt (the stores issued after the DMA may stay outstanding; CDNA4 counts loads and stores together).
template <int I, int N, class F>
CFEN_DEV void eq_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    eq_static_for<I + 1, N>(f);
  }
}

// (a __device__ helper, not a call inside the lambda: hipcc's host pass drops the stub of a kernel whose lambda calls this builtin)
CFEN_DEV void eq_dma16(const void* g, unsigned char* l) {
  __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n <= 31 (the instruction takes an immediate)
#define CFEN_EQ_VMCASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
CFEN_DEV void eq_wait_vmcnt(int n) {
  switch (n) {
    CFEN_EQ_VMCASE(0) CFEN_EQ_VMCASE(1) CFEN_EQ_VMCASE(2) CFEN_EQ_VMCASE(3) CFEN_EQ_VMCASE(4) CFEN_EQ_VMCASE(5) CFEN_EQ_VMCASE(6) CFEN_EQ_VMCASE(7)
    CFEN_EQ_VMCASE(8) CFEN_EQ_VMCASE(9) CFEN_EQ_VMCASE(10) CFEN_EQ_VMCASE(11) CFEN_EQ_VMCASE(12) CFEN_EQ_VMCASE(13) CFEN_EQ_VMCASE(14) CFEN_EQ_VMCASE(15)
    CFEN_EQ_VMCASE(16) CFEN_EQ_VMCASE(17) CFEN_EQ_VMCASE(18) CFEN_EQ_VMCASE(19) CFEN_EQ_VMCASE(20) CFEN_EQ_VMCASE(21) CFEN_EQ_VMCASE(22) CFEN_EQ_VMCASE(23)
    CFEN_EQ_VMCASE(24) CFEN_EQ_VMCASE(25) CFEN_EQ_VMCASE(26) CFEN_EQ_VMCASE(27) CFEN_EQ_VMCASE(28) CFEN_EQ_VMCASE(29) CFEN_EQ_VMCASE(30) CFEN_EQ_VMCASE(31)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;      // (never taken for the shapes below; a full drain is always correct)
  }
}
#undef CFEN_EQ_VMCASE

// NS (round 5): ring stages.  With NS = 2 (rounds 2-4) the DMA of chunk c + 1 went out at the head of chunk c, whose 24 MFMAs per wave are over in ~400 cycles:
// every chunk waited a whole LDS-DMA issue -> landed latency (~1.1 us), 24 chunks a workgroup.  NS - 1 chunks are in flight now; the landing wait is a COUNTED
// vmcnt that leaves the younger chunks' DMAs (and, in the qkv loop, the tile stores issued since) outstanding.
template <int ND, int TM, int NW, int RS, int NS = 2>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_embed_qkv2(Grouped<CfenEmbedQkvArgs> ga, int defer) {    // defer (round 6, "embed.defer_refill"): a chunk's refill goes out behind its first fragment reads
  typedef half_t T;
  const CfenEmbedQkvArgs a = ga.g[blockIdx.z];
  constexpr int KC = 32, EPL = 8;
  constexpr int D = ND * 16, NCH = ND / 2, RT = RS / 16;
  constexpr int P1 = D * 2 + 32, PP1 = P1 / 16;
  constexpr int NINS = RS * PP1 / 64;                 // DMA wave-instructions per chunk
  constexpr int NI = (NINS + NW - 1) / NW;
  constexpr int STAGE = NINS * 1024;
  constexpr int NEC = D / RS, NQC = 3 * D / RS;       // embedding / qkv chunks
  constexpr int GPT = NCH / 3;                        // fragment groups per row tile
  constexpr int NG = RT * GPT;                        // ... per chunk
  static_assert(RS % 16 == 0 && RS * PP1 % 64 == 0 && D % RS == 0 && NCH % 3 == 0, "chunk geometry");
  constexpr int NC = NEC + NQC, PF = NS - 1;          // chunks; chunks in flight beyond the one being multiplied
  constexpr bool PAIR = RT % 2 == 0 && ND % 2 == 0;   // adjacent feature tiles leave as one 16-byte store per lane (eq_pair_tiles)
  constexpr int QST = PAIR ? RT / 2 * TM : RT * TM;   // QKV store instructions per chunk and wave
  static_assert(NS >= 2 && NS * STAGE <= 80 * 1024 && PF <= NEC && (PF - 1) * NI + PF * QST <= 31, "ring geometry");
  typedef half8 frag;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[NS * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, h = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long tok0 = ((long long)blockIdx.x * NW + wave) * (TM * 16);   // the launcher guarantees M % (NW * TM * 16) == 0

  unsigned off[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int id = (i * NW + wave) * 64 + lane, row = id / PP1, col = min(id % PP1, PP1 - 3);
    off[i] = (unsigned)(row * D * 2 + col * 16);
  }
  auto issue = [&](int c, int buf) {   // chunk c: rows [c * RS, +RS) of [W_e ; W_qkv]
    const unsigned char* W = c < NEC ? (const unsigned char*)a.We + (size_t)c * RS * D * 2 : (const unsigned char*)a.Wqkv + (size_t)(c - NEC) * RS * D * 2;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int blk = i * NW + wave;
      if (blk < NINS) eq_dma16(W + off[i], lds + buf * STAGE + blk * 1024);
    }
  };
#pragma unroll
  for (int c = 0; c < PF; ++c) issue(c, c);
  const int nd = (NINS - wave + NW - 1) / NW;         // DMA instructions THIS wave issues per chunk (wave-uniform)

  // ---- gather x^T into accumulator layout ----
  const int tw = a.ws / a.p, S = tw * tw, nwx = a.W / a.ws, nwy = a.H / a.ws;
  floatx4 acc[ND][TM];
  long long tk[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const long long t = tok0 + j * 16 + r16;
    tk[j] = t;
    const int tt = (int)(t % S);
    const long long wi = t / S;
    const int wx = (int)(wi % nwx), wy = (int)((wi / nwx) % nwy);
    const long long b = wi / ((long long)nwx * nwy);
    const int y0 = wy * a.ws + (tt / tw) * a.p, x0 = wx * a.ws + (tt % tw) * a.p;
    const T* pix = (const T*)a.fmap + ((b * a.H + y0) * a.W + x0) * a.cs;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int f = i * 16 + 4 * h;
      const int ij = f / a.C, c = f - ij * a.C;
      acc[i][j] = load4<T>(pix + ((ij / a.p) * a.W + (ij % a.p)) * a.cs + c);
    }
  }
  frag xb[NCH][TM];
#pragma unroll
  for (int j = 0; j < TM; ++j)
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      floatx4 t[2] = {acc[c * 2][j], acc[c * 2 + 1][j]};
      xb[c][j] = PackB<T>::make(t);
    }
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    const floatx4 bb = *reinterpret_cast<const floatx4*>(a.be + i * 16 + 4 * h);
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] += bb + load4<T>((const T*)a.pos + (size_t)(tk[j] % S) * D + i * 16 + 4 * h);
  }

  const int a1 = r16 * P1 + h * 16;
  auto load_g = [&](const unsigned char* buf, auto gc, frag (&f)[3]) {
    constexpr int g = decltype(gc)::value, u = g / GPT, cg = g % GPT;
#pragma unroll
    for (int k = 0; k < 3; ++k) f[k] = *reinterpret_cast<const frag*>(buf + a1 + (u * 16) * P1 + (cg * 3 + k) * 64);
  };

  // ---- y = W_e x + (b_e + x + pos): embedding chunks (accumulator indices are compile-time) ----
  eq_static_for<0, NEC>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    // chunk c has landed once at most the DMAs of the chunks issued after it are outstanding (chunk 0: the token gathers and bias / position loads of the
    // prologue sit behind the ring's first DMAs -- one full drain)
    if constexpr (c == 0 || NS == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else eq_wait_vmcnt((PF - 1 < NC - 1 - c ? PF - 1 : NC - 1 - c) * nd);
    __builtin_amdgcn_s_barrier();
    if constexpr (c + PF < NC) {
      if (!defer) issue(c + PF, (c + PF) % NS);      // into the slot of chunk c - 1, which every wave has left
    }
    const unsigned char* buf = lds + (c % NS) * STAGE;
    frag F[2][3];
    load_g(buf, std::integral_constant<int, 0>{}, F[0]);
    if constexpr (c + PF < NC) {
      if (defer) issue(c + PF, (c + PF) % NS);
    }
    eq_static_for<0, NG>([&](auto gc) {
      constexpr int g = decltype(gc)::value, u = g / GPT, cg = g % GPT;
      if constexpr (g + 1 < NG) load_g(buf, std::integral_constant<int, g + 1>{}, F[(g + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[c * RT + u][j] = Mma<T>::mma(F[g & 1][k], xb[cg * 3 + k][j], acc[c * RT + u][j]);
      __builtin_amdgcn_sched_barrier(0);
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  });
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    if constexpr (PAIR) {
      T* yp = (T*)a.X1 + tk[j] * D + 16 * (h & 1) + 8 * (h >> 1);
#pragma unroll
      for (int i = 0; i < ND; i += 2) *reinterpret_cast<uint4*>(yp + i * 16) = pair_tiles16(acc[i][j], acc[i + 1][j]);
    } else {
      T* yp = (T*)a.X1 + tk[j] * D + 4 * h;
#pragma unroll
      for (int i = 0; i < ND; ++i) store4<T>(yp + i * 16, acc[i][j]);
    }
  }
  // ---- LayerNorm(y) -> B fragments ----
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < ND; ++i) sm += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    sm = col_sum(sm);
    const float mean = sm * (1.f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < ND; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = acc[i][j][r] - mean;
        q += d * d;
      }
    q = col_sum(q);
    const float rstd = rsqrtf(q * (1.f / D) + a.eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      floatx4 t[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int i = c * 2 + u;
        const floatx4 g = *reinterpret_cast<const floatx4*>(a.ln_g + i * 16 + 4 * h);
        const floatx4 b = *reinterpret_cast<const floatx4*>(a.ln_b + i * 16 + 4 * h);
        t[u] = (acc[i][j] - mean) * rstd * g + b;
      }
      xb[c][j] = PackB<T>::make(t);
    }
  }
  long long qrow[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j)
    qrow[j] = a.hm_heads ? (tk[j] / S) * (3LL * S * D) + (tk[j] % S) * HM_DH : tk[j] * (3LL * D);

  // ---- qkv chunks: every tile goes straight to HBM ----
#pragma unroll 1
  for (int cq = 0; cq < NQC; ++cq) {
    const int c = NEC + cq;
    // chunk c has landed when at most the RT * TM tile stores issued after its DMA are still outstanding (first qkv chunk: the X1
    // stores and LayerNorm parameter loads sit behind the DMA too -- drain)
    // (with NS stages: behind the DMA of chunk c come the tile stores of PF chunks and the DMAs of the up to PF - 1 younger chunks)
    if (cq == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (NS == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QST) : "memory");
    else eq_wait_vmcnt((PF - 1 < NC - 1 - c ? PF - 1 : NC - 1 - c) * nd + (cq < PF ? cq : PF) * QST);
    __builtin_amdgcn_s_barrier();
    if (c + PF < NC && !defer) issue(c + PF, (c + PF) % NS);
    const unsigned char* buf = lds + (c % NS) * STAGE;
    frag F[2][3];
    floatx4 q[TM], qe[TM];                            // qe: the even tile of a pair, kept until its odd neighbour is done
    load_g(buf, std::integral_constant<int, 0>{}, F[0]);
    if (c + PF < NC && defer) issue(c + PF, (c + PF) % NS);
    eq_static_for<0, NG>([&](auto gc) {
      constexpr int g = decltype(gc)::value, u = g / GPT, cg = g % GPT;
      if constexpr (g + 1 < NG) load_g(buf, std::integral_constant<int, g + 1>{}, F[(g + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (cg == 0) {
#pragma unroll
        for (int j = 0; j < TM; ++j) q[j] = floatx4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int j = 0; j < TM; ++j) q[j] = Mma<T>::mma(F[g & 1][k], xb[cg * 3 + k][j], q[j]);
      if constexpr (cg == GPT - 1) {
        if constexpr (PAIR) {
          if constexpr ((u & 1) == 0) {
#pragma unroll
            for (int j = 0; j < TM; ++j) qe[j] = q[j];
          } else {
            const int f = (cq * RT + u - 1) * 16 + 16 * (h & 1) + 8 * (h >> 1);     // 8 consecutive features, inside one head (heads are 24 = 3 x 8 wide)
            const long long fo = a.hm_heads ? hm_feature_off<D>(f, S) : f;
#pragma unroll
            for (int j = 0; j < TM; ++j) *reinterpret_cast<uint4*>((T*)a.QKV + qrow[j] + fo) = pair_tiles16(qe[j], q[j]);
          }
        } else {
          const int f = (cq * RT + u) * 16 + 4 * h;
          const long long fo = a.hm_heads ? hm_feature_off<D>(f, S) : f;
#pragma unroll
          for (int j = 0; j < TM; ++j) store4<T>((T*)a.QKV + qrow[j] + fo, q[j]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

template <int ND, int TM, int NW, int RS, int NS = 2>
int launch_embed_qkv2(int ng, const CfenEmbedQkvArgs* ap, hipStream_t s) {
  Grouped<CfenEmbedQkvArgs> ga;
  for (int g = 0; g < CFEN_MAX_GROUPS; ++g) ga.g[g] = ap[g < ng ? g : 0];
  const long long per = (long long)NW * TM * 16;
  CFEN_CHECK_ARG(ap[0].M % per == 0, "embed_qkv2: token count must be a multiple of %lld", per);
  const long long blocks = ap[0].M / per;
  CFEN_CHECK_ARG(blocks > 0 && blocks < (1ll << 31), "embed_qkv: bad grid");
  CFEN_LAUNCH((k_embed_qkv2<ND, TM, NW, RS, NS>), dim3((unsigned)blocks, 1, ng), dim3(NW * 64), 0, s, ga, cfen_tune_embed_defer_refill());
  CFEN_CHECK_LAUNCH("embed_qkv");
  return CFEN_OK;
}

template <typename T, int ND, int TM, int NG>
int launch_embed_qkv_lds(int ng, const CfenEmbedQkvArgs* ap, hipStream_t s) {
  Grouped<CfenEmbedQkvArgs> ga;
  for (int g = 0; g < CFEN_MAX_GROUPS; ++g) ga.g[g] = ap[g < ng ? g : 0];
  const long long per = 8LL * TM * 16, blocks = (ap[0].M + per - 1) / per;
  CFEN_CHECK_ARG(blocks > 0 && blocks < (1ll << 31), "embed_qkv: bad grid");
  CFEN_LAUNCH((k_embed_qkv_lds<T, ND, TM, NG>), dim3((unsigned)blocks, 1, ng), dim3(512), 0, s, ga);
  CFEN_CHECK_LAUNCH("embed_qkv");
  return CFEN_OK;
}

template <typename T, int ND, int TM>
int launch_embed_qkv(int ng, const CfenEmbedQkvArgs* ap, hipStream_t s) {
  constexpr int NW = 4;
  Grouped<CfenEmbedQkvArgs> ga;
  for (int g = 0; g < CFEN_MAX_GROUPS; ++g) ga.g[g] = ap[g < ng ? g : 0];
  const long long per = (long long)NW * TM * 16, blocks = (ap[0].M + per - 1) / per;
  CFEN_CHECK_ARG(blocks > 0 && blocks < (1ll << 31), "embed_qkv: bad grid");
  CFEN_LAUNCH((k_embed_qkv<T, ND, TM, NW>), dim3((unsigned)blocks, 1, ng), dim3(NW * 64), 0, s, ga);
  CFEN_CHECK_LAUNCH("embed_qkv");
  return CFEN_OK;
}

template <typename T>
int run_embed_qkv(int ng, const CfenEmbedQkvArgs* ap, hipStream_t s) {
  constexpr int EPL = Mma<T>::EPL;
  CFEN_CHECK_ARG(ng >= 1 && ng <= CFEN_MAX_GROUPS && ap, "embed_qkv: 1..%d problems per launch", CFEN_MAX_GROUPS);
  for (int g = 0; g < ng; ++g) {
    const CfenEmbedQkvArgs& a = ap[g];
    CFEN_CHECK_ARG(a.fmap && a.We && a.be && a.pos && a.ln_g && a.ln_b && a.Wqkv && a.X1 && a.QKV, "embed_qkv: null pointer");
    CFEN_CHECK_ARG(cfen_aligned16(a.fmap) && cfen_aligned16(a.We) && cfen_aligned16(a.be) && cfen_aligned16(a.pos) && cfen_aligned16(a.ln_g) &&
                   cfen_aligned16(a.ln_b) && cfen_aligned16(a.Wqkv) && cfen_aligned16(a.X1) && cfen_aligned16(a.QKV), "embed_qkv: pointers must be 16-byte aligned");
    CFEN_CHECK_ARG(a.C > 0 && a.C % EPL == 0 && a.C % 4 == 0 && a.cs % EPL == 0 && a.cs >= a.C && a.p > 0 && a.ws % a.p == 0 && a.H % a.ws == 0 &&
                   a.W % a.ws == 0 && a.B > 0, "embed_qkv: bad token geometry");
    const int tw = a.ws / a.p;
    CFEN_CHECK_ARG(a.D == a.p * a.p * a.C && a.M == (long long)a.B * (a.H / a.ws) * (a.W / a.ws) * tw * tw, "embed_qkv: D / M do not match the map");
    CFEN_CHECK_ARG(a.D == ap[0].D && a.M == ap[0].M && a.hm_heads == ap[0].hm_heads, "embed_qkv: grouped problems must have the same shape");
    CFEN_CHECK_ARG(a.hm_heads == 0 || (a.hm_heads > 0 && a.D == a.hm_heads * HM_DH), "embed_qkv: the head-major layout needs head_dim %d (D = %d, heads = %d)",
                   HM_DH, a.D, a.hm_heads);
  }
  const int lds = cfen_tune_embed_lds();   // bit 0: D = 96, bit 1: D = 192 use the LDS-staged variant
  if constexpr (sizeof(T) == 2) {   // the fp32 stages would not fit 64 KB of LDS
    if (lds & 4) {   // LDS-DMA ring + pipelined fragment groups (k_embed_qkv2); needs whole workgroups of tokens
      if (ap[0].D == 96 && ap[0].M % 256 == 0) return launch_embed_qkv2<6, 4, 4, 32>(ng, ap, s);
      if (ap[0].D == 192 && ap[0].M % 128 == 0) {
        const int ns = cfen_tune_embed_stages();
        return ns == 2 ? launch_embed_qkv2<12, 2, 4, 32, 2>(ng, ap, s) : ns == 3 ? launch_embed_qkv2<12, 2, 4, 32, 3>(ng, ap, s)
             : ns == 5 ? launch_embed_qkv2<12, 2, 4, 32, 5>(ng, ap, s) : launch_embed_qkv2<12, 2, 4, 32, 4>(ng, ap, s);
      }
    }
    if (ap[0].D == 96 && (lds & 1)) return launch_embed_qkv_lds<T, 6, 4, 6>(ng, ap, s);
    if (ap[0].D == 192 && (lds & 2)) return launch_embed_qkv_lds<T, 12, 2, 4>(ng, ap, s);
  }
  switch (ap[0].D) {
    case 96: return launch_embed_qkv<T, 6, 4>(ng, ap, s);
    case 192: return launch_embed_qkv<T, 12, 2>(ng, ap, s);
    default:
      cfen_set_error("embed_qkv: fused kernel supports D in {96,192}, got %d", ap[0].D);
      return CFEN_ERR_ARG;
  }
}

}  // namespace

int& cfen_tune_embed_defer_refill() {   // k_embed_qkv2: 1 = refill behind the chunk's first fragment reads, 0 = right behind the barrier (rounds 2-5)
  static int v = 1;
  return v;
}
int& cfen_tune_embed_stages() {   // ring stages of k_embed_qkv2 at D = 192 ("embed.stages"): 2 (rounds 2-4), 3, 4 (default, round 5), 5
  static int v = 4;
  return v;
}

int& cfen_tune_embed_lds() {
  static int v = 6;
  return v;
}

bool cfen_embed_qkv_supported(int D) { return D == 96 || D == 192; }

int cfen_embed_qkv_impl_g(int dtype, int ng, const CfenEmbedQkvArgs* a, hipStream_t s) {
  if (dtype == 1) return run_embed_qkv<half_t>(ng, a, s);
  if (dtype == 0) return run_embed_qkv<float>(ng, a, s);
  cfen_set_error("embed_qkv: unknown dtype %d", dtype);
  return CFEN_ERR_ARG;
}
