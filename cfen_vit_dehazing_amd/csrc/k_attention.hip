// Multi-head self-attention (bias-free nn.MultiheadAttention, reference v3:1364,1383-1386) and LayerNorm
// (v3:1370-1371) for LViT windows and GViT pooled maps.
//
// Attention.  One workgroup = (sequence, head, group of 64 queries); each of its 4 waves owns 16
// queries.  Scores are computed TRANSPOSED, S^T = K Q^T, so that after the MFMA every lane holds
// scores of ONE query (column = lane & 15) for 4 keys per tile:
//   * the softmax row reduction is in-lane plus two xor-shuffles (lanes l, l^16, l^32, l^48),
//   * the probabilities are already laid out as the B operand of O^T = V^T P^T -- no LDS round trip.
// Keys stream through LDS in super-blocks (K row-major, V transposed) with an online softmax, so any
// sequence length works (S = 256 windows, S = 1024 windows of the 1024^2 config, S = 1..256 pooled
// maps) and head_dim 24 is zero-padded to the MFMA K of 32 (fp16) / 2x16 (fp32).
#include <stdlib.h>
#include "cfen_common.hpp"

namespace {

template <typename T> struct PV;
template <> struct PV<half_t> {
  // o[d][q] += sum over 32 keys; p0/p1 = probabilities of key tiles 0/1 for query (lane & 15)
  static CFEN_DEV floatx4 run(const unsigned char* vt, int h, floatx4 p0, floatx4 p1, floatx4 o) {
    half4 lo = *reinterpret_cast<const half4*>(vt + 8 * h);
    half4 hi = *reinterpret_cast<const half4*>(vt + 32 + 8 * h);
    half8 a = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    half8 b = {(half_t)p0[0], (half_t)p0[1], (half_t)p0[2], (half_t)p0[3],
               (half_t)p1[0], (half_t)p1[1], (half_t)p1[2], (half_t)p1[3]};
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, o, 0, 0, 0);
  }
};
template <> struct PV<float> {
  static CFEN_DEV floatx4 run(const unsigned char* vt, int h, floatx4 p0, floatx4 p1, floatx4 o) {
    floatx4 a0 = *reinterpret_cast<const floatx4*>(vt + 16 * h);
    floatx4 a1 = *reinterpret_cast<const floatx4*>(vt + 64 + 16 * h);
    o = Mma<float>::mma(a0, p0, o);
    return Mma<float>::mma(a1, p1, o);
  }
};

// V^T staging: a thread takes one 16-byte vector of V for TWO consecutive keys and writes (d, key pair) entries.
// For fp16 the two keys pack into one 32-bit LDS word: half as many ds_write instructions as a 16-bit scatter.
template <typename T> struct VtStore;
template <> struct VtStore<half_t> {
  static CFEN_DEV void put(unsigned char* vl, int vrow, int d, int key, half8 a, half8 b) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      typedef __attribute__((ext_vector_type(2))) _Float16 half2_t;
      half2_t pr = {a[e], b[e]};
      *reinterpret_cast<half2_t*>(vl + (d + e) * vrow + key * 2) = pr;
    }
  }
};
template <> struct VtStore<float> {
  static CFEN_DEV void put(unsigned char* vl, int vrow, int d, int key, floatx4 a, floatx4 b) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      typedef __attribute__((ext_vector_type(2))) float float2_t;
      float2_t pr = {a[e], b[e]};
      *reinterpret_cast<float2_t*>(vl + (d + e) * vrow + key * 4) = pr;
    }
  }
};

template <typename T, int NDT>
__global__ __launch_bounds__(256) void k_attention(PtrG<const T> QKVg, PtrG<T> Og, int S, int D, int heads,
                                                   int dh, int skb, int nqg, float scale_log2) {
  const T* __restrict__ QKV = QKVg.p[blockIdx.z];
  T* __restrict__ O = Og.p[blockIdx.z];
  constexpr int KC = Mma<T>::KC, EPL = Mma<T>::EPL;
  constexpr int NCQ = (NDT * 16 + KC - 1) / KC;    // K chunks covering the (padded) head dim
  constexpr int DHPK = NCQ * KC;
  constexpr int SZ = (int)sizeof(T);
  typedef typename Mma<T>::frag frag;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int krow = DHPK * SZ + 16;
  const int vrow = skb * SZ + 16;
  unsigned char* Kl = smem;
  unsigned char* Vl = smem + (size_t)skb * krow;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, h = lane >> 4;
  const int qg = blockIdx.x % nqg;
  const int sh = blockIdx.x / nqg;
  const int head = sh % heads, seq = sh / heads;
  const int ld = 3 * D;
  const T* Qp = QKV + (size_t)seq * S * ld + head * dh;
  const T* Kp = Qp + D;
  const T* Vp = Qp + 2 * D;
  const int q0 = (qg * 4 + wave) * 16;
  const bool active = q0 < S;

  frag qf[NCQ];
  {
    const int qrow = min(q0 + r16, S - 1);
#pragma unroll
    for (int c = 0; c < NCQ; ++c) {
      int d = c * KC + h * EPL;
      qf[c] = (d < dh) ? load_frag<T>(Qp + (size_t)qrow * ld + d) : Mma<T>::zero();
    }
  }
  floatx4 o[NDT];
#pragma unroll
  for (int i = 0; i < NDT; ++i) o[i] = floatx4{0.f, 0.f, 0.f, 0.f};
  float m_run = -1e30f, l_run = 0.f;

  constexpr int NVK = DHPK / EPL;        // 16-byte vectors per staged K row
  constexpr int NVV = NDT * 16 / EPL;    // 16-byte vectors per V row (padded head dim)
  for (int ks = 0; ks < S; ks += skb) {
    if (ks) __syncthreads();
    // staging in batches of four (K) / two (V) sweeps: unconditional loads at clamped addresses first, then the LDS stores with the validity
    // applied -- one `cond ? load : 0` + LDS store per sweep was a memory round trip per sweep (18 in a row for 256 keys of 96 features)
    for (int base = 0; base < skb * NVK; base += 4 * 256) {
      frag r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = min(base + u * 256 + tid, skb * NVK - 1), kr = idx / NVK, d = (idx - kr * NVK) * EPL;
        r[u] = load_frag<T>(Kp + (size_t)min(ks + kr, S - 1) * ld + (d < dh ? d : 0));
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = base + u * 256 + tid, kr = idx / NVK, v = idx - kr * NVK;
        if (idx < skb * NVK) *reinterpret_cast<frag*>(Kl + kr * krow + v * 16) = (ks + kr < S && v * EPL < dh) ? r[u] : Mma<T>::zero();
      }
    }
    for (int base = 0; base < (skb / 2) * NVV; base += 2 * 256) {
      frag r0[2], r1[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int idx = min(base + u * 256 + tid, (skb / 2) * NVV - 1), kp = idx / NVV, d = (idx - kp * NVV) * EPL, key = ks + 2 * kp;
        r0[u] = load_frag<T>(Vp + (size_t)min(key, S - 1) * ld + (d < dh ? d : 0));
        r1[u] = load_frag<T>(Vp + (size_t)min(key + 1, S - 1) * ld + (d < dh ? d : 0));
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int idx = base + u * 256 + tid, kp = idx / NVV, v = idx - kp * NVV, d = v * EPL, key = ks + 2 * kp;
        if (idx >= (skb / 2) * NVV) continue;
        const frag v0 = (key < S && d < dh) ? r0[u] : Mma<T>::zero(), v1 = (key + 1 < S && d < dh) ? r1[u] : Mma<T>::zero();
        VtStore<T>::put(Vl, vrow, d, 2 * kp, v0, v1);
      }
    }
    __syncthreads();
    if (!active) continue;
    const int kend = min(skb, S - ks);
    for (int kb = 0; kb < kend; kb += 32) {
      floatx4 st[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        st[t] = floatx4{0.f, 0.f, 0.f, 0.f};
        const unsigned char* kp = Kl + (kb + 16 * t + r16) * krow + h * 16;
#pragma unroll
        for (int c = 0; c < NCQ; ++c) st[t] = Mma<T>::mma(*reinterpret_cast<const frag*>(kp + c * 64), qf[c], st[t]);
      }
      float mx = -1e30f;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int key = ks + kb + 16 * t + 4 * h + r;
          float s = (key < S) ? st[t][r] * scale_log2 : -1e30f;
          st[t][r] = s;
          mx = fmaxf(mx, s);
        }
      mx = col_max(mx);
      const float m_new = fmaxf(m_run, mx);
      const float alpha = exp2f(m_run - m_new);
      float rs = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float p = exp2f(st[t][r] - m_new);
          st[t][r] = p;
          rs += p;
        }
      rs = col_sum(rs);
      l_run = l_run * alpha + rs;
      m_run = m_new;
#pragma unroll
      for (int i = 0; i < NDT; ++i) {
        o[i] *= alpha;
        o[i] = PV<T>::run(Vl + (i * 16 + r16) * vrow + kb * SZ, h, st[0], st[1], o[i]);
      }
    }
  }
  if (active && q0 + r16 < S) {
    const float inv = 1.f / l_run;
    T* op = O + ((size_t)seq * S + q0 + r16) * D + head * dh;
#pragma unroll
    for (int i = 0; i < NDT; ++i) {
      int d = i * 16 + 4 * h;
      if (d < dh) store4<T>(op + d, o[i] * inv);
    }
  }
}

// Window fast path (S = 64 or 256 keys: every LViT window and the larger GViT maps of the 512x512 configs).
// One workgroup = one (sequence, head): K and V^T are staged ONCE, each wave walks its query tiles with an
// exact two-pass softmax over all S keys held in registers (no online rescaling, branch-free unrolled loops
// so the K-fragment LDS reads of a query tile are all in flight together).  The row sum comes out of the PV
// MFMA itself when the head dim has a padding row (dh = 24 -> row 24 of V^T is all ones).
template <typename T, int NDT, int NKT, bool ONES>
__global__ __launch_bounds__(256) void k_attention_win(PtrG<const T> QKVg, PtrG<T> Og, int D, int heads, int dh,
                                                       float scale_log2, int nblk) {
  const T* __restrict__ QKV = QKVg.p[blockIdx.z];
  T* __restrict__ O = Og.p[blockIdx.z];
  constexpr int KC = Mma<T>::KC, EPL = Mma<T>::EPL, SZ = (int)sizeof(T);
  constexpr int NCQ = (NDT * 16 + KC - 1) / KC;
  constexpr int DHPK = NCQ * KC;
  constexpr int S = NKT * 16;
  constexpr int KROW = DHPK * SZ + 32;              // stride = 32 (mod 64) bytes: conflict-free ds_read_b128
  constexpr int VROW = S * SZ + 16;
  typedef typename Mma<T>::frag frag;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Kl = smem;
  unsigned char* Vl = smem + S * KROW;
  constexpr bool ones_row = ONES;                   // a free V^T row (dh < NDT*16) carries the softmax denominator

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, h = lane >> 4;
  const int blk = (int)xcd_chunked_block(blockIdx.x, gridDim.x);   // all heads of a window on one XCD: they share qkv lines
  if (blk >= nblk) return;
  const int head = blk % heads, seq = blk / heads;
  const int ld = 3 * D;
  const T* Qp = QKV + (size_t)seq * S * ld + head * dh;
  const T* Kp = Qp + D;
  const T* Vp = Qp + 2 * D;

  // every global load of the workgroup is issued up front and unconditionally (a piece past the head dimension re-reads piece 0 and is
  // replaced by zeros afterwards): the K pieces, the V pieces and, for the 16-bit type, this wave's query fragments.  As loops of
  // `cond ? load : 0` followed by the LDS store (and a query load per q iteration) the kernel walked ~10 memory round trips one by one.
  constexpr int NVK = DHPK / EPL, NVV = NDT * 16 / EPL;
  constexpr int NSK = (S * NVK + 255) / 256, NSV = ((S / 2) * NVV + 255) / 256, NQ = (NKT + 3) / 4;
  constexpr bool QPRE = sizeof(T) == 2;             // fp32 fragments are twice as many registers: its queries are fetched per iteration
  frag kst[NSK], v0st[NSV], v1st[NSV], qst[QPRE ? NQ : 1][NCQ];
#pragma unroll
  for (int i = 0; i < NSK; ++i) {
    const int idx = min(tid + i * 256, S * NVK - 1), kr = idx / NVK, d = (idx - kr * NVK) * EPL;
    kst[i] = load_frag<T>(Kp + (size_t)kr * ld + (d < dh ? d : 0));
  }
#pragma unroll
  for (int i = 0; i < NSV; ++i) {
    const int idx = min(tid + i * 256, (S / 2) * NVV - 1), kp = idx / NVV, d = (idx - kp * NVV) * EPL;
    v0st[i] = load_frag<T>(Vp + (size_t)(2 * kp) * ld + (d < dh ? d : 0));
    v1st[i] = load_frag<T>(Vp + (size_t)(2 * kp + 1) * ld + (d < dh ? d : 0));
  }
  if constexpr (QPRE) {
#pragma unroll
    for (int i = 0; i < NQ; ++i)
#pragma unroll
      for (int cc = 0; cc < NCQ; ++cc) {
        const int d = cc * KC + h * EPL, qrow = min((wave + 4 * i) * 16 + r16, S - 1);
        qst[i][cc] = load_frag<T>(Qp + (size_t)qrow * ld + (d < dh ? d : 0));
      }
  }
#pragma unroll
  for (int i = 0; i < NSK; ++i) {
    const int idx = tid + i * 256, kr = idx / NVK, v = idx - kr * NVK, d = v * EPL;
    if (idx < S * NVK) *reinterpret_cast<frag*>(Kl + kr * KROW + v * 16) = d < dh ? kst[i] : Mma<T>::zero();
  }
#pragma unroll
  for (int i = 0; i < NSV; ++i) {
    const int idx = tid + i * 256, kp = idx / NVV, v = idx - kp * NVV, d = v * EPL;
    if (idx >= (S / 2) * NVV) continue;
    frag v0 = d < dh ? v0st[i] : Mma<T>::zero(), v1 = d < dh ? v1st[i] : Mma<T>::zero();
    if (ones_row && d == dh / EPL * EPL) {
      v0[dh % EPL] = (T)1.0f;
      v1[dh % EPL] = (T)1.0f;
    }
    VtStore<T>::put(Vl, VROW, d, 2 * kp, v0, v1);
  }
  __syncthreads();

  const float c = scale_log2;
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    const int qt = wave + 4 * qi;
    if (qt >= NKT) break;
    const int q0 = qt * 16;
    frag qf[NCQ];
#pragma unroll
    for (int cc = 0; cc < NCQ; ++cc) {
      int d = cc * KC + h * EPL;
      if constexpr (QPRE) qf[cc] = (d < dh) ? qst[qi][cc] : Mma<T>::zero();
      else qf[cc] = (d < dh) ? load_frag<T>(Qp + (size_t)(q0 + r16) * ld + d) : Mma<T>::zero();
    }
    floatx4 st[NKT];
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
      floatx4 a = floatx4{0.f, 0.f, 0.f, 0.f};
      const unsigned char* kp = Kl + (t * 16 + r16) * KROW + h * 16;
#pragma unroll
      for (int cc = 0; cc < NCQ; ++cc) a = Mma<T>::mma(*reinterpret_cast<const frag*>(kp + cc * 64), qf[cc], a);
      st[t] = a;
    }
    float mx = -1e30f;
#pragma unroll
    for (int t = 0; t < NKT; ++t) mx = fmaxf(fmaxf(fmaxf(fmaxf(mx, st[t][0]), st[t][1]), st[t][2]), st[t][3]);   // chains of two: v_max3_f32
    mx = col_max(mx);
    const float mc = -mx * c;
    float rs = 0.f;
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p = __builtin_amdgcn_exp2f(fmaf(st[t][r], c, mc));
        st[t][r] = p;
        if (!ones_row) rs += p;
      }
    floatx4 o[NDT];
#pragma unroll
    for (int i = 0; i < NDT; ++i) o[i] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < NKT / 2; ++kb)
#pragma unroll
      for (int i = 0; i < NDT; ++i) o[i] = PV<T>::run(Vl + (i * 16 + r16) * VROW + kb * 32 * SZ, h, st[2 * kb], st[2 * kb + 1], o[i]);
    float l;
    if (ones_row) {   // denominator = row `dh` of O^T: tile dh/16, lane group (dh%16)/4, register dh%4
      float cand = 0.f;
#pragma unroll
      for (int i = 0; i < NDT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (i * 16 + r == dh - ((dh % 16) / 4) * 4) cand = o[i][r];
      l = col_sum(h == (dh % 16) / 4 ? cand : 0.f);   // broadcast from the lane group that owns row `dh`
    } else {
      l = col_sum(rs);
    }
    const float inv = 1.f / l;
    T* op = O + ((size_t)seq * S + q0 + r16) * D + head * dh;
#pragma unroll
    for (int i = 0; i < NDT; ++i) {
      int d = i * 16 + 4 * h;
      if (d < dh) store4<T>(op + d, o[i] * inv);
    }
  }
}

// Window attention on the HEAD-MAJOR qkv layout that k_embed_qkv writes for LViT levels 1-2 (fp16, head_dim 24, S = 64 / 256):
//     qkv[(window * heads + head) * 3 + {q, k, v}][S][24]
// so the q, k and v blocks of one (window, head) are three contiguous 12 KB runs: the staging loads are fully coalesced 16-byte
// pieces (the row-major [M][3D] layout hands every workgroup 48-byte slivers of 576-byte rows -- 2.7x the algorithmic traffic in
// rocprof).  K is staged row-major with a 96-byte pitch (conflict-free ds_read_b128, dims 24..31 zeroed); V is staged ROW-major as
// well (96-byte pitch, feature 24 = 1.0 so that row 24 of O^T is the softmax denominator) and read as the transposed MFMA operand
// with ds_read_b64_tr_b16 -- no transposing scatter of 4-byte pairs into LDS.  Same two-pass softmax as k_attention_win.
typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 fp16x4_t;
CFEN_DEV half4 lds_read_tr4(const unsigned char* p) {
  const fp16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)p);
  half4 o;
  __builtin_memcpy(&o, &v, 8);
  return o;
}

template <int NKT>
__global__ __launch_bounds__(256) void k_attention_hm(PtrG<const half_t> QKVg, PtrG<half_t> Og, int D, int heads, float scale_log2, int nblk) {
  constexpr int S = NKT * 16, DH = 24, KP = 96, VP = 96;
  __shared__ __attribute__((aligned(16))) unsigned char Kl[S * KP];
  __shared__ __attribute__((aligned(16))) unsigned char Vl[S * VP];
  const half_t* __restrict__ QKV = QKVg.p[blockIdx.z];
  half_t* __restrict__ O = Og.p[blockIdx.z];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, h = lane >> 4;
  const int blk = (int)xcd_chunked_block(blockIdx.x, gridDim.x);
  if (blk >= nblk) return;
  const int head = blk % heads, seq = blk / heads;
  const half_t* Qp = QKV + (size_t)blk * 3 * S * DH;
  const half_t* Kp = Qp + S * DH;
  const half_t* Vp = Kp + S * DH;

  // all global loads of the workgroup go out together, unconditionally: the K / V pieces of the staging sweep (piece 3 of a row is padding:
  // its lane re-reads piece 2 and keeps the constant) and this wave's query fragments.  Written as a loop of `if (c < 3) load; store to LDS`
  // and a query load at the top of every q iteration, the kernel was 4 + NKT / 4 memory round trips in a row.
  constexpr int NST = S * 4 / 256, NQ = NKT / 4;
  static_assert(S * 4 % 256 == 0 && NKT % 4 == 0, "staging sweep / query tiles per wave");
  half8 kst[NST], vst[NST], qst[NQ];
#pragma unroll
  for (int i = 0; i < NST; ++i) {
    const int idx = tid + i * 256, row = idx >> 2, c = min(idx & 3, 2);
    kst[i] = load_frag<half_t>(Kp + row * DH + c * 8);
    vst[i] = load_frag<half_t>(Vp + row * DH + c * 8);
  }
#pragma unroll
  for (int i = 0; i < NQ; ++i) qst[i] = load_frag<half_t>(Qp + ((wave + 4 * i) * 16 + r16) * DH + min(h, 2) * 8);
#pragma unroll
  for (int i = 0; i < NST; ++i) {
    const int idx = tid + i * 256, row = idx >> 2, c = idx & 3;
    half8 kv = Mma<half_t>::zero(), vv = Mma<half_t>::zero();
    vv[0] = (half_t)1.0f;                                      // piece 3 of a V row: feature 24 = 1 (denominator row), 25..31 = 0
    if (c < 3) { kv = kst[i]; vv = vst[i]; }
    *reinterpret_cast<half8*>(Kl + row * KP + c * 16) = kv;
    *reinterpret_cast<half8*>(Vl + row * VP + c * 16) = vv;
  }
  __syncthreads();

  const float c = scale_log2;
  const int li = lane & 15;
  const unsigned char* vbase = Vl + (4 * h + (li >> 2)) * VP + (li & 3) * 8;   // tr-read: lane 4q+p of a 16-lane group -> row q, columns 4p..4p+3
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    const int q0 = (wave + 4 * qi) * 16;
    const half8 qf = h < 3 ? qst[qi] : Mma<half_t>::zero();
    floatx4 st[NKT];
#pragma unroll
    for (int t = 0; t < NKT; ++t)
      st[t] = Mma<half_t>::mma(*reinterpret_cast<const half8*>(Kl + (t * 16 + r16) * KP + h * 16), qf, floatx4{0.f, 0.f, 0.f, 0.f});
    float mx = -1e30f;
#pragma unroll
    for (int t = 0; t < NKT; ++t) mx = fmaxf(fmaxf(fmaxf(fmaxf(mx, st[t][0]), st[t][1]), st[t][2]), st[t][3]);   // chains of two: v_max3_f32
    mx = col_max(mx);
    const float mc = -mx * c;
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) st[t][r] = __builtin_amdgcn_exp2f(fmaf(st[t][r], c, mc));
    floatx4 o[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int kb = 0; kb < NKT / 2; ++kb) {
      const half8 b = {(half_t)st[2 * kb][0], (half_t)st[2 * kb][1], (half_t)st[2 * kb][2], (half_t)st[2 * kb][3],
                       (half_t)st[2 * kb + 1][0], (half_t)st[2 * kb + 1][1], (half_t)st[2 * kb + 1][2], (half_t)st[2 * kb + 1][3]};
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const half4 lo = lds_read_tr4(vbase + (kb * 32) * VP + i * 32);        // keys kb*32 + 4h .. +3, feature i*16 + (lane & 15)
        const half4 hi = lds_read_tr4(vbase + (kb * 32 + 16) * VP + i * 32);   // keys kb*32 + 16 + 4h .. +3
        const half8 a = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        o[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, o[i], 0, 0, 0);
      }
    }
    const float l = col_sum(h == 2 ? o[1][0] : 0.f);   // row 24 of O^T (tile 1, lane group 2, register 0) = sum of the probabilities
    const float inv = 1.f / l;
    half_t* op = O + ((size_t)seq * S + q0 + r16) * D + head * DH + 4 * h;
    store4<half_t>(op, o[0] * inv);
    if (h < 2) store4<half_t>(op + 16, o[1] * inv);
  }
}

// The same window attention for the 1024-token windows of the 1024 x 1024 configuration (BASELINE config 4: --loadSize 512 --patch_size 64),
// round 3.  K and V of one (window, head) still fit a CU -- 1024 rows x 64 bytes each = 128 KB -- so they are staged ONCE and every query of
// the window reads them from LDS; the generic streaming kernel re-staged K / V per 64-query block through 60 KB of LDS (1.03 GB of HBM
// traffic per launch, 6.5 % MFMA-busy, 52 % of that configuration's forward: profiles/r03_cfg4_*).  What changes against k_attention_hm:
//   * 64-byte row pitch (no room for the 96-byte conflict-free pitch): the 16-byte pieces of a K row are XOR-swizzled with (4 - (row >> 2)) & 3
//     and the two 32-byte halves of a V row are swapped on rows with bit 2 set -- conflict-free ds_read_b128 / ds_read_b64_tr_b16 again;
//   * the scores of a query against 1024 keys do not fit registers: keys go in blocks of 256 with the running-maximum rescale (the
//     denominator row of O^T, V feature 24 = 1, is rescaled with the rest);
//   * a wave works on TWO query tiles at a time, so every K fragment and every transposed V fragment read from LDS feeds two MFMAs;
//   * 8 waves: two per SIMD.
template <int NKT, int NKB, int NW = 8>
__global__ __launch_bounds__(NW * 64) void k_attention_hm_long(PtrG<const half_t> QKVg, PtrG<half_t> Og, int D, int heads, float scale_log2, int nblk) {
  constexpr int S = NKT * 16, DH = 24, KP = 64, NT = NW * 64;
  static_assert(NKT % NKB == 0 && NKB % 2 == 0 && 2 * S * KP <= 160 * 1024 && NKT % (2 * NW) == 0, "geometry");
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * S * KP];
  unsigned char* const Kl = lds;
  unsigned char* const Vl = lds + S * KP;
  const half_t* __restrict__ QKV = QKVg.p[blockIdx.z];
  half_t* __restrict__ O = Og.p[blockIdx.z];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, h = lane >> 4;
  const int blk = (int)xcd_chunked_block(blockIdx.x, gridDim.x);
  if (blk >= nblk) return;
  const int head = blk % heads, seq = blk / heads;
  const half_t* Qp = QKV + (size_t)blk * 3 * S * DH;
  const half_t* Kp = Qp + S * DH;
  const half_t* Vp = Kp + S * DH;

  // ---- stage K and V: 3 data pieces + 1 constant piece per 64-byte row, swizzled ----
  constexpr int NPC = (S * 3 + NT - 1) / NT;   // data pieces per thread and matrix (6 at S = 1024)
  half8 kst[NPC], vst[NPC];
#pragma unroll
  for (int i = 0; i < NPC; ++i) {
    const int idx = min(tid + i * NT, S * 3 - 1);   // piece idx of the contiguous [S][24] block: row idx / 3, piece idx % 3
    kst[i] = load_frag<half_t>(Kp + idx * 8);
    vst[i] = load_frag<half_t>(Vp + idx * 8);
  }
#pragma unroll
  for (int i = 0; i < NPC; ++i) {
    const int idx = tid + i * NT, row = idx / 3, c = idx - row * 3;
    if (idx >= S * 3) break;
    const int ks = (4 - ((row >> 2) & 3)) & 3;
    *reinterpret_cast<half8*>(Kl + row * KP + ((c ^ ks) << 4)) = kst[i];
    // V: piece c of the row = features 8c .. 8c+7; half = c >> 1, swapped on rows with bit 2 set
    *reinterpret_cast<half8*>(Vl + row * KP + ((((c >> 1) ^ ((row >> 2) & 1)) << 5) | ((c & 1) << 4))) = vst[i];
  }
#pragma unroll
  for (int i = 0; i < (S + NT - 1) / NT; ++i) {
    const int row = tid + i * NT;
    if (row >= S) break;
    const int ks = (4 - ((row >> 2) & 3)) & 3;
    half8 one = Mma<half_t>::zero();
    *reinterpret_cast<half8*>(Kl + row * KP + ((3 ^ ks) << 4)) = one;            // K dims 24..31 = 0
    one[0] = (half_t)1.0f;                                                           // V feature 24 = 1 (denominator row), 25..31 = 0
    *reinterpret_cast<half8*>(Vl + row * KP + (((1 ^ ((row >> 2) & 1)) << 5) | 16)) = one;
  }
  __syncthreads();

  const float c = scale_log2;
  const int li = lane & 15;
  // K fragment of key tile t: row t*16 + r16, piece h -> swizzled by (row >> 2) & 3 = (r16 >> 2) (16 t is a multiple of 16)
  const unsigned char* kbase = Kl + r16 * KP + ((h ^ ((4 - (r16 >> 2)) & 3)) << 4);
  // V tr-read: lane 4q+p of a 16-lane group -> key row 4h + q (+ 32 kb, + 16), columns 4p..4p+3 of feature half i; (row >> 2) & 1 = h & 1
  const unsigned char* vbase = Vl + (4 * h + (li >> 2)) * KP + (li & 3) * 8;
  const int vsw = (h & 1) << 5;
  constexpr int QPW = NKT / NW;            // query tiles per wave (8), two at a time
#pragma unroll 1
  for (int qi = 0; qi < QPW; qi += 2) {
    half8 qf[2];
    int q0[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      q0[j] = (wave + NW * (qi + j)) * 16;
      qf[j] = load_frag<half_t>(Qp + (q0[j] + r16) * DH + min(h, 2) * 8);
      if (h == 3) qf[j] = Mma<half_t>::zero();
    }
    floatx4 o[2][2];
    float m[2] = {-1e30f, -1e30f};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) o[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int kb0 = 0; kb0 < NKT; kb0 += NKB) {
      floatx4 st[NKB][2];
#pragma unroll
      for (int t = 0; t < NKB; ++t) {
        const half8 kf = *reinterpret_cast<const half8*>(kbase + ((kb0 + t) * 16) * KP);
#pragma unroll
        for (int j = 0; j < 2; ++j) st[t][j] = Mma<half_t>::mma(kf, qf[j], floatx4{0.f, 0.f, 0.f, 0.f});
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float mx = -1e30f;
#pragma unroll
        for (int t = 0; t < NKB; ++t) mx = fmaxf(fmaxf(fmaxf(fmaxf(mx, st[t][j][0]), st[t][j][1]), st[t][j][2]), st[t][j][3]);   // v_max3_f32 chains
        mx = fmaxf(col_max(mx), m[j]);
        const float alpha = __builtin_amdgcn_exp2f((m[j] - mx) * c);     // first block: exp2(-huge) = 0 on zero accumulators
        m[j] = mx;
        const float mc = -mx * c;
#pragma unroll
        for (int t = 0; t < NKB; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) st[t][j][r] = __builtin_amdgcn_exp2f(fmaf(st[t][j][r], c, mc));
#pragma unroll
        for (int i = 0; i < 2; ++i) o[i][j] *= alpha;
      }
#pragma unroll
      for (int kb = 0; kb < NKB / 2; ++kb) {
        half8 b[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
          b[j] = half8{(half_t)st[2 * kb][j][0], (half_t)st[2 * kb][j][1], (half_t)st[2 * kb][j][2], (half_t)st[2 * kb][j][3],
                       (half_t)st[2 * kb + 1][j][0], (half_t)st[2 * kb + 1][j][1], (half_t)st[2 * kb + 1][j][2], (half_t)st[2 * kb + 1][j][3]};
        const unsigned char* vk = vbase + ((kb0 * 16) + kb * 32) * KP;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const half4 lo = lds_read_tr4(vk + ((i << 5) ^ vsw));               // keys +4h .. +3, feature i*16 + (lane & 15)
          const half4 hi = lds_read_tr4(vk + 16 * KP + ((i << 5) ^ vsw));     // keys +16 + 4h .. +3
          const half8 a = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
          for (int j = 0; j < 2; ++j) o[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b[j], o[i][j], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float l = col_sum(h == 2 ? o[1][j][0] : 0.f);   // row 24 of O^T = sum of the probabilities (rescaled with the rest)
      const float inv = 1.f / l;
      // the 24 features of the head leave as one 16-byte store from three of the four lane rows (pair_tiles16: row 0 features 0..7, row 1 16..23, row 2 8..15; row 3
      // holds the padding rows of the second tile)
      const uint4 v = pair_tiles16(o[0][j] * inv, o[1][j] * inv);
      half_t* op = O + ((size_t)seq * S + q0[j] + r16) * D + head * DH + 16 * (h & 1) + 8 * (h >> 1);
      if (h < 3) *reinterpret_cast<uint4*>(op) = v;
    }
  }
}

template <typename T, int NDT, int NKT>
int launch_attn_win(int ng, const void* const* qkv, void* const* out, int nseq, int heads, int dh, size_t smem, hipStream_t s) {
  const float scale_log2 = 1.4426950408889634f / sqrtf((float)dh);
  PtrG<const T> qg{};
  PtrG<T> og{};
  for (int g = 0; g < ng; ++g) { qg.p[g] = (const T*)qkv[g]; og.p[g] = (T*)out[g]; }
  if (dh < NDT * 16)
    CFEN_LAUNCH((k_attention_win<T, NDT, NKT, true>), dim3(cfen_grid8((long long)nseq * heads), 1, ng), dim3(256), smem, s, qg, og,
                       heads * dh, heads, dh, scale_log2, nseq * heads);
  else
    CFEN_LAUNCH((k_attention_win<T, NDT, NKT, false>), dim3(cfen_grid8((long long)nseq * heads), 1, ng), dim3(256), smem, s, qg, og,
                       heads * dh, heads, dh, scale_log2, nseq * heads);
  CFEN_CHECK_LAUNCH("attention");
  return CFEN_OK;
}

template <typename T, int NDT>
bool try_attn_win(int ng, const void* const* qkv, void* const* out, int nseq, int S, int heads, int dh, hipStream_t s, int* rc) {
  constexpr int KC = Mma<T>::KC;
  constexpr int NCQ = (NDT * 16 + KC - 1) / KC;
  const int SZ = (int)sizeof(T);
  if (S != 256 && S != 64) return false;
  const size_t smem = (size_t)S * (NCQ * KC * SZ + 32) + (size_t)NDT * 16 * (S * SZ + 16);
  if (smem > 64 * 1024) return false;   // larger dynamic-LDS kernel nodes crash hipGraph instantiation on ROCm 7.2
  *rc = S == 256 ? launch_attn_win<T, NDT, 16>(ng, qkv, out, nseq, heads, dh, smem, s) : launch_attn_win<T, NDT, 4>(ng, qkv, out, nseq, heads, dh, smem, s);
  return true;
}

template <typename T, int NDT>
int launch_attn(int ng, const void* const* qkv, void* const* out, int nseq, int S, int heads, int dh, hipStream_t s) {
  constexpr int KC = Mma<T>::KC;
  constexpr int NCQ = (NDT * 16 + KC - 1) / KC;
  const int SZ = (int)sizeof(T);
  const int sp = (S + 31) / 32 * 32;
  const int perkey = NCQ * KC * SZ + 16 + NDT * 16 * SZ;
  int skb = ((60 * 1024 - NDT * 16 * 16) / perkey) / 32 * 32;
  if (skb < 32) skb = 32;
  if (skb > sp) skb = sp;
  const size_t smem = (size_t)skb * (NCQ * KC * SZ + 16) + (size_t)NDT * 16 * (skb * SZ + 16);
  const int nqg = (S + 63) / 64;
  const long long blocks = (long long)nseq * heads * nqg;
  CFEN_CHECK_ARG(blocks > 0 && blocks < (1ll << 31), "attention: bad grid");
  const float scale_log2 = 1.4426950408889634f / sqrtf((float)dh);
  PtrG<const T> qg{};
  PtrG<T> og{};
  for (int g = 0; g < ng; ++g) { qg.p[g] = (const T*)qkv[g]; og.p[g] = (T*)out[g]; }
  CFEN_LAUNCH((k_attention<T, NDT>), dim3((unsigned)blocks, 1, ng), dim3(256), smem, s, qg, og, S,
                     heads * dh, heads, dh, skb, nqg, scale_log2);
  CFEN_CHECK_LAUNCH("attention");
  return CFEN_OK;
}

template <typename T>
int dispatch_attn(int ng, const void* const* qkv, void* const* out, int nseq, int S, int heads, int dh, hipStream_t s) {
  CFEN_CHECK_ARG(nseq > 0 && S > 0 && heads > 0 && dh > 0, "attention: empty problem");
  CFEN_CHECK_ARG(dh % Mma<T>::EPL == 0, "attention: head_dim %d must be a multiple of %d", dh, Mma<T>::EPL);
  CFEN_CHECK_ARG(ng >= 1 && ng <= CFEN_MAX_GROUPS, "attention: 1..%d problems per launch", CFEN_MAX_GROUPS);
  for (int g = 0; g < ng; ++g)
    CFEN_CHECK_ARG(qkv[g] && out[g] && cfen_aligned16(qkv[g]) && cfen_aligned16(out[g]), "attention: pointers must be non-null and 16-byte aligned");
  CFEN_CHECK_ARG((long long)nseq * heads < (1ll << 31), "attention: bad grid");
  int rc = CFEN_OK;
  if (dh <= 32) return try_attn_win<T, 2>(ng, qkv, out, nseq, S, heads, dh, s, &rc) ? rc : launch_attn<T, 2>(ng, qkv, out, nseq, S, heads, dh, s);
  if (dh <= 96) return try_attn_win<T, 6>(ng, qkv, out, nseq, S, heads, dh, s, &rc) ? rc : launch_attn<T, 6>(ng, qkv, out, nseq, S, heads, dh, s);
  if (dh <= 128) return try_attn_win<T, 8>(ng, qkv, out, nseq, S, heads, dh, s, &rc) ? rc : launch_attn<T, 8>(ng, qkv, out, nseq, S, heads, dh, s);
  cfen_set_error("attention: head_dim %d > 128 unsupported", dh);
  return CFEN_ERR_ARG;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over the last dim, statistics in fp32 (two passes over registers).  A row is handled by a
// group of G = 16, 32 or 64 lanes (G >= D / (16-byte vector)), so a wave normalises 4 / 2 / 1 rows and all
// lanes carry data even for the 192-byte rows of LViT level 1.
// Dn <= D: the row's trailing / interleaved padding entries are exact zeros that do not belong to the normalised vector
// (v5 LViT: 6- and 12-channel maps live at a channel stride of 8 / 16, so a token row carries zero slots); statistics are over
// the Dn real entries, the padding slots get gamma = beta = 0 from the packer.
template <typename T, int MAXV, int G>
__global__ __launch_bounds__(256) void k_layernorm(PtrG<const T> Xg, PtrG<T> Yg, PtrG<const float> gg, PtrG<const float> bg, int M, int D, float eps, int Dn) {
  const T* __restrict__ X = Xg.p[blockIdx.z];
  T* __restrict__ Y = Yg.p[blockIdx.z];
  const float* __restrict__ g = gg.p[blockIdx.z];
  const float* __restrict__ b = bg.p[blockIdx.z];
  constexpr int EPL = Vec16<T>::N;
  constexpr int RPW = 64 / G;                        // rows per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / G, gl = lane % G;
  const long long row = ((long long)blockIdx.x * 4 + wave) * RPW + sub;
  const bool live = row < M;
  const int nvec = D / EPL;
  const T* x = X + (live ? row : 0) * D;
  float v[MAXV][EPL];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    int idx = gl + i * G;
    if (idx < nvec) {
      Vec16<T>::load(x + idx * EPL, v[i]);
#pragma unroll
      for (int e = 0; e < EPL; ++e) sum += v[i][e];
    }
  }
  sum = group_sum<G>(sum);
  const float mean = sum / (float)Dn;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    int idx = gl + i * G;
    if (idx < nvec) {
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        float d = v[i][e] - mean;
        sq += d * d;
      }
    }
  }
  sq = group_sum<G>(sq);
  sq -= (float)(D - Dn) * mean * mean;              // the zero slots each added (0 - mean)^2
  const float rstd = rsqrtf(fmaxf(sq, 0.f) / (float)Dn + eps);
  if (!live) return;
  T* y = Y + row * D;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    int idx = gl + i * G;
    if (idx < nvec) {
      float o[EPL];
#pragma unroll
      for (int e = 0; e < EPL; ++e) o[e] = (v[i][e] - mean) * rstd * g[idx * EPL + e] + b[idx * EPL + e];
      Vec16<T>::store(y + idx * EPL, o);
    }
  }
}

template <typename T, int MAXV>
int launch_ln(int ng, const void* const* X, void* const* Y, const float* const* g, const float* const* b, int M, int D, float eps, hipStream_t s, int Dn) {
  constexpr int EPL = Vec16<T>::N;
  CFEN_CHECK_ARG(M > 0 && D > 0, "layernorm: empty problem");
  if (Dn <= 0) Dn = D;
  CFEN_CHECK_ARG(Dn <= D, "layernorm: %d real entries in rows of %d", Dn, D);
  CFEN_CHECK_ARG(D % EPL == 0 && D <= 64 * MAXV * EPL, "layernorm: D=%d unsupported (multiple of %d, <= %d)", D, EPL, 64 * MAXV * EPL);
  CFEN_CHECK_ARG(ng >= 1 && ng <= CFEN_MAX_GROUPS, "layernorm: 1..%d problems per launch", CFEN_MAX_GROUPS);
  PtrG<const T> xg{};
  PtrG<T> yg{};
  PtrG<const float> gg{}, bg{};
  for (int k = 0; k < ng; ++k) {
    CFEN_CHECK_ARG(X[k] && Y[k] && g[k] && b[k] && cfen_aligned16(X[k]) && cfen_aligned16(Y[k]), "layernorm: pointers must be non-null and 16-byte aligned");
    xg.p[k] = (const T*)X[k]; yg.p[k] = (T*)Y[k]; gg.p[k] = g[k]; bg.p[k] = b[k];
  }
  const int nvec = D / EPL;
  if (nvec <= 16) {
    CFEN_LAUNCH((k_layernorm<T, 1, 16>), dim3((M + 15) / 16, 1, ng), dim3(256), 0, s, xg, yg, gg, bg, M, D, eps, Dn);
  } else if (nvec <= 32) {
    CFEN_LAUNCH((k_layernorm<T, 1, 32>), dim3((M + 7) / 8, 1, ng), dim3(256), 0, s, xg, yg, gg, bg, M, D, eps, Dn);
  } else {
    CFEN_LAUNCH((k_layernorm<T, MAXV, 64>), dim3((M + 3) / 4, 1, ng), dim3(256), 0, s, xg, yg, gg, bg, M, D, eps, Dn);
  }
  CFEN_CHECK_LAUNCH("layernorm");
  return CFEN_OK;
}

}  // namespace

int cfen_attention_impl_g(int dtype, int ng, const void* const* qkv, void* const* out, int nseq, int S, int heads, int dh, hipStream_t s) {
  if (dtype == 1) return dispatch_attn<half_t>(ng, qkv, out, nseq, S, heads, dh, s);
  if (dtype == 0) return dispatch_attn<float>(ng, qkv, out, nseq, S, heads, dh, s);
  cfen_set_error("attention: unknown dtype %d", dtype);
  return CFEN_ERR_ARG;
}
int& cfen_tune_attn_hm_pair() {   // 1: 256-token windows run on k_attention_hm_long<16, 16> (two query tiles per K / V fragment, 8 waves); 2: 1024-token windows on 8 waves instead of 16
  static int v = 0;
  return v;
}

bool cfen_attention_hm_supported(int dtype, int S, int dh) { return dtype == 1 && dh == 24 && (S == 1024 || S == 256 || S == 64); }

int cfen_attention_hm_impl_g(int dtype, int ng, const void* const* qkv, void* const* out, int nseq, int S, int heads, int dh, hipStream_t s) {
  CFEN_CHECK_ARG(cfen_attention_hm_supported(dtype, S, dh), "attention (head-major): fp16, head_dim 24, S in {64, 256, 1024} only");
  CFEN_CHECK_ARG(ng >= 1 && ng <= CFEN_MAX_GROUPS && nseq > 0 && heads > 0 && (long long)nseq * heads < (1ll << 31), "attention (head-major): bad problem");
  PtrG<const half_t> qg{};
  PtrG<half_t> og{};
  for (int g = 0; g < ng; ++g) {
    CFEN_CHECK_ARG(qkv[g] && out[g] && cfen_aligned16(qkv[g]) && cfen_aligned16(out[g]), "attention (head-major): pointers must be non-null and 16-byte aligned");
    qg.p[g] = (const half_t*)qkv[g]; og.p[g] = (half_t*)out[g];
  }
  const float scale_log2 = 1.4426950408889634f / sqrtf((float)dh);
  const dim3 grid(cfen_grid8((long long)nseq * heads), 1, ng);
  // S = 1024: K / V take 128 KB of LDS, one workgroup per CU -- so it has 16 waves of 128 registers (key blocks of 128), not 8 of 208 (blocks of
  // 256): a wave issues a vector instruction every ~10 cycles, the softmax needs four waves per SIMD; 1024 x 1024 images, batch 4:
  // 6.29 -> 6.17 ms ("attn.hm_pair" = 2: the 8-wave form)
  if (S == 1024 && cfen_tune_attn_hm_pair() != 2)
    CFEN_LAUNCH((k_attention_hm_long<64, 8, 16>), grid, dim3(1024), 0, s, qg, og, heads * dh, heads, scale_log2, nseq * heads);
  else if (S == 1024)
    CFEN_LAUNCH((k_attention_hm_long<64, 16>), grid, dim3(512), 0, s, qg, og, heads * dh, heads, scale_log2, nseq * heads);
  else if (S == 256 && cfen_tune_attn_hm_pair() == 1)
    CFEN_LAUNCH((k_attention_hm_long<16, 8>), grid, dim3(512), 0, s, qg, og, heads * dh, heads, scale_log2, nseq * heads);
  else if (S == 256)
    CFEN_LAUNCH((k_attention_hm<16>), grid, dim3(256), 0, s, qg, og, heads * dh, heads, scale_log2, nseq * heads);
  else
    CFEN_LAUNCH((k_attention_hm<4>), grid, dim3(256), 0, s, qg, og, heads * dh, heads, scale_log2, nseq * heads);
  CFEN_CHECK_LAUNCH("attention (head-major)");
  return CFEN_OK;
}

int cfen_attention_impl(int dtype, const void* qkv, void* out, int nseq, int S, int heads, int dh, hipStream_t s) {
  return cfen_attention_impl_g(dtype, 1, &qkv, &out, nseq, S, heads, dh, s);
}

int cfen_layernorm_impl_g(int dtype, int ng, const void* const* X, void* const* Y, const float* const* g, const float* const* b, int M, int D,
                          float eps, hipStream_t s, int Dn) {
  if (dtype == 1) return launch_ln<half_t, 4>(ng, X, Y, g, b, M, D, eps, s, Dn);
  if (dtype == 0) return launch_ln<float, 8>(ng, X, Y, g, b, M, D, eps, s, Dn);
  cfen_set_error("layernorm: unknown dtype %d", dtype);
  return CFEN_ERR_ARG;
}
int cfen_layernorm_impl(int dtype, const void* X, void* Y, const float* g, const float* b, int M, int D, float eps, hipStream_t s) {
  return cfen_layernorm_impl_g(dtype, 1, &X, &Y, &g, &b, M, D, eps, s, 0);
}
