// One LViT block per workgroup-window, everything between the input map and the output map on chip.
//
// Replaces, for an LViT instance with embedding dim D = 96 (level 1: C = 24, 4 heads of 24, 32x32-pixel windows = 256 tokens), the
// launch chain k_embed_qkv2 -> k_attention_hm -> k_mlp2 and their X1 / QKV / ATT round trips through HBM:
//   Crop2x2 + unfold + linear_encoding + residual + position          (v3:1025-1056, 1140-1143, 1166)
//   norm1 + nn.MultiheadAttention(bias=False) + residual               (v3:1364-1371, 1383-1386)
//   norm2 + linear1/ReLU/linear2 + residual, mlp_head + residual       (v3:1387-1389, 1173)
//   fold + Join2x2                                                     (v3:1176-1186, 1046-1056)
//
// A workgroup of 16 waves (round 3; 8 in round 2) owns ONE window; wave w owns its tokens 16w .. 16w+15 for the whole chain.  As in k_mlp2 / k_embed_qkv2 the
// residual stream of a token tile lives in fp32 MFMA accumulators (rows = features, columns = tokens), an accumulator tile pair is the
// B operand of the next GEMM, and every weight matrix streams through an LDS ring by LDS-DMA in 32-row chunks (one raw s_barrier per
// chunk).  Round 3: the weights come as ONE stream of 1 KiB MFMA A fragments in consumption order (packing.pack_lvit_window, as
// k_stream.hip's format): a chunk is 6 or 12 contiguous KiB, a DMA instruction is a linear 1 KiB copy, fragment reads are lane-linear
// (conflict-free without row padding), and the 13 KiB stages leave room for THREE of them beside K / V: two chunks in flight.  With the
// two-stage ring of round 2 an MLP chunk took 0.94 us -- one LDS-DMA issue -> landed latency (MI355X_MICROARCH.md, ldsdma-fill: ~1.1 us)
// per chunk, 2.5x its MFMA time -- and with two chunks in flight it still takes 0.93 us: the DMA is not what a chunk waits for.  Bound
// experiment on this kernel (24 images, back to back; barriers / DMA refills / the MLP chunks' MFMAs removed one at a time and together):
// 435 -> 383 / 360 / 397 us, all three 343 us; 32 hidden units instead of 384: 263 of 385 us -- the embedding, K / V and attention chunks
// (13 of 37) are two thirds of the kernel: 128 exp2 + ~500 other vector instructions per lane, head and token-tile pair.  What round 2 brought:
//   * K and V of the whole window (256 keys x 4 heads x 24 dims, fp16) stay in LDS (2 x 56 KB; a kernel node may use the CU's full
//     160 KB of LDS -- tools/repro/lds_graph_probe.hip).  The K/V chunks of the qkv projection write their tiles there instead of HBM.
//   * Attention is run as "an MLP whose activation is softmax(Q K^T) V": the chunk of head h carries W_q[h] (32 rows) and the
//     32-column slice of the out-projection that belongs to h.  Q_h = W_q[h] LN1(x) comes out of the MFMA as the B operand of
//     S^T = K_h Q_h^T (the rows of W_q[h] are laid out on the host so that the accumulator pair packs into natural d order, padded
//     24 -> 32 with zero rows); the softmax runs in registers exactly as in k_attention_hm (V read as the transposed operand with
//     ds_read_b64_tr_b16); O_h^T packs into the B operand of x += W_p[:, h] O_h.  Neither q, k, v nor the attention output ever
//     exists in HBM.
// HBM traffic of an instance: read the map once, write it once, weights from L2.
#include <type_traits>
#include "cfen_common.hpp"
#include "cfen_lvit.hpp"

namespace {

template <int I, int N, class F>
CFEN_DEV void lv_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    lv_static_for<I + 1, N>(f);
  }
}

CFEN_DEV void lv_dma16(const void* g, unsigned char* l) {
  __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 lv_fp16x4;
CFEN_DEV half4 lv_read_tr4(const unsigned char* p) {
  const lv_fp16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) lv_fp16x4*)p);
  half4 o;
  __builtin_memcpy(&o, &v, 8);
  return o;
}

// ---- hand-issued LDS reads with counted waits (round 5; the k_stream.hip idiom) ----
// hipcc schedules the attention loops as "ds_read -> s_waitcnt lgkmcnt(0) -> MFMA -> s_nop 6 -> v_max3" per key tile through ONE fragment register
// set: every one of the 16 K fragments and of the 16 V fragment pairs per head and token tile exposed a whole LDS round trip (disassembly of
// k_lvit_window<6,16,1,0>, profiles/r05_lvit_window_disasm.txt).  Here the K fragment of key tile t is read INTO THE REGISTERS OF THE SCORE TILE IT
// PRODUCES (an MFMA may overwrite its own A operand), so up to LV_KPD reads are in flight at no register cost; LDS returns in order, so "fragment t
// has landed" = at most (reads issued after t) outstanding.
template <int OFF>
CFEN_DEV void lv_rd128(floatx4& f, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(addr), "n"(OFF));
}
typedef unsigned int lv_u32x2 __attribute__((ext_vector_type(2)));
template <int OFF>
CFEN_DEV void lv_rd_tr64(lv_u32x2& f, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f) : "v"(addr), "n"(OFF));
}
template <int N>
CFEN_DEV void lv_wait(floatx4& f) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N));
}
template <int N>
CFEN_DEV void lv_wait2(lv_u32x2& a, lv_u32x2& b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
// MFMA result -> VALU reader with inline asm in between: the hazard recognizer does not carry the matrix pipe's write-back latency across an asm
// statement (k_stream.hip: mfma_results_settle)
CFEN_DEV void lv_settle() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 11" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
CFEN_DEV half8 lv_as_half8(floatx4 v) {
  half8 o;
  __builtin_memcpy(&o, &v, 16);
  return o;
}

CFEN_DEV half8 lv_pack(floatx4 a, floatx4 b) {
  half8 f = {(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3], (half_t)b[0], (half_t)b[1], (half_t)b[2], (half_t)b[3]};
  return f;
}

// ND = D / 16 (6); NW waves x TM token tiles of 16 = the 256 tokens of one window (16 x 1 at four waves per SIMD: the default; 8 x 2 at two;
// 4 x 4 with one wave per SIMD and the whole 512-register file); 4 heads of 24
// FR (round 5): 1 = the embedding and K / V matrices stream in chunks of 64 rows (12 fragments: what a stage holds for the MLP chunks anyway) instead of 32 --
// 5 barrier-separated chunks in front of the attention instead of 9, each with 12 MFMAs per wave instead of 6; same fragment stream, same arithmetic, same bits.
// MC (round 6): 2 = the MLP chunks behind the first two run as DOUBLE chunks (64 hidden units, 24 KiB + 256 B of bias per barrier) on a four-slot ring laid over the K / V tiles, which are
// dead once the last head is done: 13 barriers instead of 24 in the MLP part.  Stamped sections (tools/dbg_lvit_sections.py): an MLP chunk took 1730-2060 cycles for 768 cycles of MFMA
// per SIMD -- after every barrier all 16 waves read their fragments at once, multiply at once and meet again; twice the work per barrier halves that lockstep cost.
// DR (round 6): 1 = a chunk's refill (the LDS-DMA of the chunk two ahead) is issued BEHIND the chunk's first fragment reads instead of right behind its barrier: an LDS-DMA piece costs
// the issuing wave 60-140 cycles (tools/dbg_mlp3_stamps.py), which then pass while its own fragment reads are in flight instead of in front of them.
template <int ND, int NW, int TM, int SM = 0, int FR = 0, int MC = 1, int DR = 0>   // SM = 0: softmax denominator summed on the vector pipe; 1: on the matrix pipe (round 4; MEASURED SLOWER: 138 / 395 us against
                                                // 131 / 381 us for 512 / 1536 windows, tools/bench_lvit_window.py -- "lvit.shape" = 3 runs it); 2 (round 5): as 0 with the K / V
                                                // fragment reads of the attention loops issued by hand, LV_KPD / one key block ahead of the MFMAs ("lvit.shape" = 4)
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4))) void k_lvit_window(Grouped<LvitArgs> ga, unsigned long long* stamps, int dbg) {
  typedef half_t T;
  // timing runs ("lvit.debug" = 64): s_memtime at the section boundaries of workgroup 0 -> stamps[wave][prologue, embedding, LN1 + K / V, attention, LN2 + MLP stage a, stage b, fold, 100 MHz ticks]
  auto now = [&]() -> unsigned long long {
    unsigned long long t = 0;
    if (stamps) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
  };
  const unsigned long long tk0 = now(), rt0 = stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;
  typedef half8 frag;
  const LvitArgs a = ga.g[blockIdx.z];       // by value: a reference re-read b1a / b1b from the argument segment (s_load + wait) behind every chunk's barrier
  constexpr int KC = 32, S = 256, DH = 24, NH = 4;
  static_assert(NW * TM * 16 == S, "one workgroup = one window");
  constexpr int D = ND * 16, NCH = ND / 2;
  static_assert(D == NH * DH && NCH == 3, "built for D = 96: 4 heads of 24, one fragment group per row tile");
  constexpr int N1 = 2 * NCH, N2 = ND;                     // fragments of a chunk's R1 (32 rows x D: row tile u, k-chunk k -> u * NCH + k) and
                                                           // R2 (D rows x 32-wide k slice: row tile i) parts, 1 KiB each
  constexpr int NINS = N1 + N2 + 1, STAGE = NINS * 1024, NI = (NINS + NW - 1) / NW, NS = 3;   // + the MLP chunks' 128 bytes of bias
  constexpr int R2 = N1 * 1024, R3 = (N1 + N2) * 1024;
  constexpr int KVP = NH * DH * 2 + 32;                    // K / V row pitch: 4 heads x 24 dims, 224 B (= 32 mod 64)
  constexpr int KOFF = 0, VOFF = S * KVP, RING = 2 * S * KVP;
  static_assert(RING + NS * STAGE <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[RING + NS * STAGE];
  unsigned char* ring = lds + RING;

  const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, h = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nhc = a.Hm / 32;
  constexpr int NE = FR ? 2 : D / 32, NKV = FR ? D / 32 : 2 * D / 32, NA = NH;    // embedding / K+V / attention chunks (FR: 64 + 32 rows, then 3 x 64)
  constexpr int NFRONT = (D / 32 + 2 * D / 32) * N1;       // fragments in front of the attention chunks (either chunking)
  const int nchunks = NE + NKV + NA + 2 * nhc;

  // ---- DMA plan: instruction i of this wave copies fragment i * NW + wave of the chunk (the stream is in consumption order) ----
  // chunk t of the stream: NE embedding + NKV key / value chunks of N1 fragments, then NA attention and 2 * nhc MLP chunks of N1 + N2
  const unsigned char* const ws = (const unsigned char*)a.Ws;
  auto nfrag_of = [&](int t) { return FR ? (t == 1 ? N1 : N1 + N2) : (t < NE + NKV ? N1 : N1 + N2); };
  auto fstart_of = [&](int t) {                              // first fragment of chunk t in the stream
    if (t >= NE + NKV) return NFRONT + (t - NE - NKV) * (N1 + N2);
    if (!FR) return t * N1;
    return t == 0 ? 0 : t == 1 ? 2 * N1 : 3 * N1 + (t - 2) * 2 * N1;
  };
  auto issue = [&](int t, int buf) -> int {                  // returns the number of DMA instructions THIS wave issued
    const int nf = nfrag_of(t);
    const unsigned char* src = ws + (size_t)fstart_of(t) * 1024 + lane * 16;
    const int m = t - NE - NKV - NA;                         // MLP chunk index (bias b1 of its 32 hidden units rides behind the fragments)
    const unsigned char* bsrc = m < 0 ? nullptr : (const unsigned char*)(m >= nhc ? a.b1b : a.b1a) + (size_t)(m >= nhc ? m - nhc : m) * 128 + min(lane, 7) * 16;
    int n = 0;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int blk = i * NW + wave;
      unsigned char* dst = ring + buf * STAGE + blk * 1024;
      if (blk < nf) { lv_dma16(src + blk * 1024, dst); ++n; }
      else if (blk == N1 + N2 && bsrc) { lv_dma16(bsrc, dst); ++n; }
    }
    return n;
  };
  // chunk t is read behind one barrier; chunks t + 1 and t + 2 are in flight behind it (three stages).  The wait leaves this wave's DMAs of
  // chunk t + 1 outstanding (a stricter wait where other vector loads were issued since: still correct)
  const int t0 = NE + NKV + NA;                              // first MLP chunk
  const bool dbl = MC == 2 && nhc % 2 == 0 && !(dbg & 2);
  int inflight = 0;                                          // DMA instructions of this wave for the chunk after the one being waited for
  int pend_kind = 0, pend_a = 0, pend_b = 0;                 // the refill the current chunk owes: 1 = issue(a, b), 2 = issue_span(a, 2, slot b of the K / V area)
  auto begin_chunk = [&](int t) -> const unsigned char* {
    if (inflight >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (inflight == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (inflight == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (inflight == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    inflight = 0;
    pend_kind = t + 2 < nchunks && !(dbl && t + 2 >= t0 + 2) ? 1 : 0;
    pend_a = t + 2;
    pend_b = (t + 2) % NS;
    if constexpr (!DR) {
      if (pend_kind) inflight = issue(pend_a, pend_b);
      pend_kind = 0;
    }
    return ring + (t % NS) * STAGE;
  };
  // MC = 2: MLP single chunks m .. m + cnt - 1 (one stage's: their b1 slices are contiguous) -> dst: cnt * 12 fragments, then cnt * 128 bytes of bias in the next KiB
  constexpr int DSTAGE = (2 * (N1 + N2) + 1) * 1024, NDS = RING / DSTAGE, NID = (2 * (N1 + N2) + 1 + NW - 1) / NW;
  static_assert(MC == 1 || NDS >= 3, "the double-chunk ring needs three slots in the K / V area");
  auto issue_span = [&](int m, int cnt, unsigned char* dst) -> int {
    const int nf = cnt * (N1 + N2);
    const unsigned char* src = ws + (size_t)(NFRONT + (NA + m) * (N1 + N2)) * 1024 + lane * 16;
    const unsigned char* bsrc = (const unsigned char*)(m >= nhc ? a.b1b : a.b1a) + (size_t)(m >= nhc ? m - nhc : m) * 128 + min(lane, cnt * 8 - 1) * 16;
    int n = 0;
#pragma unroll
    for (int i = 0; i < NID; ++i) {
      const int blk = i * NW + wave;
      if (blk < nf) { lv_dma16(src + blk * 1024, dst + blk * 1024); ++n; }
      else if (blk == nf) { lv_dma16(bsrc, dst + blk * 1024); ++n; }
    }
    return n;
  };
  const int ndbl = dbl ? nhc - 1 : 0;                        // double chunks behind the two single ones
  auto wait_inflight = [&]() {
    if (dbg & 1) inflight = 0;       // timing / debugging: every refill drained before the barrier
    if (inflight >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (inflight == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (inflight == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (inflight == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  // the two single MLP chunks in front (prefetched into the old ring while the last heads run): their barriers are the first behind the attention, so the doubles they issue may
  // overwrite K / V
  auto begin_mlp_single = [&](int t) -> const unsigned char* {
    wait_inflight();
    __builtin_amdgcn_s_barrier();
    const int k = t - t0;
    inflight = 0;
    pend_kind = k < ndbl ? 2 : 0;
    pend_a = 2 + 2 * k;
    pend_b = k;
    if constexpr (!DR) {
      if (pend_kind) inflight = issue_span(pend_a, 2, lds + pend_b * DSTAGE);
      pend_kind = 0;
    }
    return ring + (t % NS) * STAGE;
  };
  auto begin_double = [&](int k) -> const unsigned char* {
    wait_inflight();
    __builtin_amdgcn_s_barrier();
    inflight = 0;
    pend_kind = k + 2 < ndbl ? 2 : 0;
    pend_a = 2 + 2 * (k + 2);
    pend_b = (k + 2) % NDS;
    if constexpr (!DR) {
      if (pend_kind) inflight = issue_span(pend_a, 2, lds + pend_b * DSTAGE);
      pend_kind = 0;
    }
    return lds + (k % NDS) * DSTAGE;
  };
  auto refill = [&]() {                                      // DR: called once per chunk, behind its first fragment reads
    if constexpr (DR) {
      if (pend_kind == 1) inflight = issue(pend_a, pend_b);
      else if (pend_kind == 2) inflight = issue_span(pend_a, 2, lds + pend_b * DSTAGE);
      pend_kind = 0;
    }
  };
  issue(0, 0);
  inflight = nchunks > 1 ? issue(1, 1) : 0;

  // the 32 pad bytes of every K / V row are read by the last head's padded fragments: they must hold finite values
  for (int r = tid; r < 2 * S; r += NW * 64) {
    half8 z = Mma<T>::zero();
    unsigned char* row = lds + (r < S ? KOFF : VOFF) + (r & (S - 1)) * KVP + NH * DH * 2;
    *reinterpret_cast<half8*>(row) = z;
    *reinterpret_cast<half8*>(row + 16) = z;
  }

  // ---- window / token geometry ----
  const int nwx = a.W / a.ws, nwy = a.H / a.ws;
  const int win = blockIdx.x;
  const int wx = win % nwx, wy = (win / nwx) % nwy, b = win / (nwx * nwy);
  const int tw = a.ws / a.p;                               // 16 tokens per window row
  int tok[TM];
  floatx4 acc[ND][TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int t = wave * (TM * 16) + j * 16 + r16;
    tok[j] = t;
    const int y0 = wy * a.ws + (t / tw) * a.p, x0 = wx * a.ws + (t % tw) * a.p;
    const T* pix = (const T*)a.fmap + (((size_t)b * a.H + y0) * a.W + x0) * a.cs_in;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int f = i * 16 + 4 * h;
      const int ij = f / a.C, c = f - ij * a.C;
      acc[i][j] = load4<T>(pix + ((ij / a.p) * a.W + (ij % a.p)) * a.cs_in + c);
    }
  }
  frag xb[NCH][TM];
#pragma unroll
  for (int j = 0; j < TM; ++j)
#pragma unroll
    for (int c = 0; c < NCH; ++c) xb[c][j] = lv_pack(acc[c * 2][j], acc[c * 2 + 1][j]);
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    const floatx4 bb = *reinterpret_cast<const floatx4*>(a.be + i * 16 + 4 * h);
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] += bb + load4<T>((const T*)a.pos + (size_t)tok[j] * D + i * 16 + 4 * h);
  }

  const int a1 = lane * 16;                    // lane part of an R1 fragment address (lane-linear 1 KiB fragments)
  const int a2 = R2 + lane * 16;               // ... R2
  const int a3 = R3 + 16 * h;                  // ... bias vector
  // R1 fragment group (row tile u of the chunk): NCH = 3 fragments
  auto load_r1 = [&](const unsigned char* buf, int u, frag (&f)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) f[k] = *reinterpret_cast<const frag*>(buf + a1 + (u * NCH + k) * 1024);
  };
  // R2 fragment group: feature-row tiles 3 ig .. 3 ig + 2 of the chunk's 32-wide k slice
  auto load_r2 = [&](const unsigned char* buf, int ig, frag (&f)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) f[k] = *reinterpret_cast<const frag*>(buf + a2 + (ig * 3 + k) * 1024);
  };
  auto layer_norm_to_xb = [&](const float* gamma, const float* beta) {
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      float sm = 0.f;
#pragma unroll
      for (int i = 0; i < ND; ++i) sm += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
      sm = col_sum(sm);
      float mean = sm * (1.f / D);
      asm volatile("" : "+v"(mean));      // the mean is rounded ONCE in every instantiation: hipcc's fp-contract fused sm * (1 / D) into the subtractions below in some template
                                          // instantiations and not in others (1-ulp differences between workgroup shapes that the tests compare bit for bit)
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
        floatx4 t2[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int i = c * 2 + u;
          const floatx4 g = *reinterpret_cast<const floatx4*>(gamma + i * 16 + 4 * h);
          const floatx4 bt = *reinterpret_cast<const floatx4*>(beta + i * 16 + 4 * h);
          t2[u] = (acc[i][j] - mean) * rstd * g + bt;
        }
        xb[c][j] = lv_pack(t2[0], t2[1]);
      }
    }
  };

  const unsigned long long tk1 = now();
  // ---- y = W_e x + (b_e + x + pos): embedding chunks (accumulator indices are compile-time) ----
  lv_static_for<0, NE>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    const unsigned char* buf = begin_chunk(c);
    constexpr int NP = FR ? (c == 0 ? 2 : 1) : 1;             // pairs of 16-row tiles in this chunk
    constexpr int T0 = FR ? (c == 0 ? 0 : 4) : c * 2;         // first accumulator (feature) tile of the chunk
    lv_static_for<0, NP>([&](auto pc) {
      constexpr int pr = decltype(pc)::value;
      frag F[2][3];
      load_r1(buf, 2 * pr, F[0]);
      load_r1(buf, 2 * pr + 1, F[1]);
      if constexpr (pr == 0) refill();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
          for (int j = 0; j < TM; ++j) acc[T0 + 2 * pr + u][j] = Mma<T>::mma(F[u][k], xb[k][j], acc[T0 + 2 * pr + u][j]);
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  });
  const unsigned long long tk2 = now();
  layer_norm_to_xb(a.ln1_g, a.ln1_b);        // acc keeps x1 (the residual stream); xb = LN1(x1)

  // ---- K and V of the window -> LDS: rows [0, 96) of Wkv are the K features of the 4 heads, rows [96, 192) the V features ----
#pragma unroll 1
  for (int c = 0; c < NKV; ++c) {
    const unsigned char* buf = begin_chunk(NE + c);
    constexpr int NP = FR ? 2 : 1;                           // pairs of 16-row tiles per chunk
#pragma unroll
    for (int pr = 0; pr < NP; ++pr) {
      frag F[2][3];
      load_r1(buf, 2 * pr, F[0]);
      load_r1(buf, 2 * pr + 1, F[1]);
      if (pr == 0) refill();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        floatx4 q[TM];
#pragma unroll
        for (int j = 0; j < TM; ++j) q[j] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
          for (int j = 0; j < TM; ++j) q[j] = Mma<T>::mma(F[u][k], xb[k][j], q[j]);
        const int t = (c * NP + pr) * 2 + u;               // feature tile 0..11: K tiles 0..5, V tiles 0..5
        unsigned char* base = lds + (t < ND ? KOFF : VOFF) + (t < ND ? t : t - ND) * 32 + 8 * h;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          const half4 v = {(half_t)q[j][0], (half_t)q[j][1], (half_t)q[j][2], (half_t)q[j][3]};
          *reinterpret_cast<half4*>(base + tok[j] * KVP) = v;
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }

  const unsigned long long tk3 = now();
  // ---- attention, one head per chunk: Q_h = W_q[h] LN1(x) -> softmax(K_h Q_h^T) -> O_h -> x += W_p[:, h] O_h ----
  const float cs = a.scale_log2;
  const frag ones = {(half_t)1, (half_t)1, (half_t)1, (half_t)1, (half_t)1, (half_t)1, (half_t)1, (half_t)1};
  const int li = lane & 15;
  const int vlane = (4 * h + (li >> 2)) * KVP + (li & 3) * 8;     // tr-read: lane 4q+p of a 16-lane group -> key row q, columns 4p..4p+3
  const unsigned lbase = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)lds;   // LDS byte address of the array (asm reads)
#pragma unroll 1
  for (int hl = 0; hl < NA; ++hl) {
    const unsigned char* buf = begin_chunk(NE + NKV + hl);     // the first of these barriers also publishes every wave's K / V tiles
    frag F[2][3];
    load_r1(buf, 0, F[0]);
    load_r1(buf, 1, F[1]);
    refill();
    __builtin_amdgcn_sched_barrier(0);
    floatx4 hq[2][TM];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int j = 0; j < TM; ++j) hq[u][j] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int j = 0; j < TM; ++j) hq[u][j] = Mma<T>::mma(F[u][k], xb[k][j], hq[u][j]);
    frag att[TM];
    const unsigned char* Kh = lds + KOFF + hl * (DH * 2) + r16 * KVP + h * 16;
    const unsigned char* Vh = lds + VOFF + hl * (DH * 2) + vlane;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const frag qb = lv_pack(hq[0][j], hq[1][j]);           // Q_h^T for 16 queries: k slot 8h'+e holds d = 8h'+e (host row layout), d >= 24 zero
      floatx4 st[S / 16];
      if constexpr (SM == 2) {
        // S^T = K_h Q_h^T with LV_KPD fragment reads in flight: fragment t lands in st[t]'s own registers and the MFMA overwrites it with the scores
        constexpr int NT = S / 16, KPD = 8;
        const unsigned ka = lbase + (unsigned)(KOFF + hl * (DH * 2) + r16 * KVP + h * 16);
        __builtin_amdgcn_sched_barrier(0);
        lv_static_for<0, KPD>([&](auto tc) { lv_rd128<decltype(tc)::value * 16 * KVP>(st[decltype(tc)::value], ka); });
        lv_static_for<0, NT>([&](auto tc) {
          constexpr int t = decltype(tc)::value;
          if constexpr (t + KPD < NT) lv_rd128<(t + KPD) * 16 * KVP>(st[t + KPD], ka);
          lv_wait<(t + KPD < NT ? KPD : NT - 1 - t)>(st[t]);
          st[t] = Mma<T>::mma(lv_as_half8(st[t]), qb, floatx4{0.f, 0.f, 0.f, 0.f});
        });
        lv_settle();
      } else {
#pragma unroll
      for (int t = 0; t < S / 16; ++t)
        st[t] = Mma<T>::mma(*reinterpret_cast<const frag*>(Kh + (t * 16) * KVP), qb, floatx4{0.f, 0.f, 0.f, 0.f});
      }
      // row maximum: two chains of three-operand maxima (v_max3_f32: half the instructions of a two-operand tree; round 4)
      float mx = -1e30f, mx2 = -1e30f;
      float rs = 0.f;
      if constexpr (SM < 8) {
#pragma unroll
      for (int t = 0; t < S / 16; ++t) {
        mx = fmaxf(fmaxf(mx, st[t][0]), st[t][1]);
        mx2 = fmaxf(fmaxf(mx2, st[t][2]), st[t][3]);
      }
      mx = col_max(fmaxf(mx, mx2));
      const float mc = -mx * cs;
#pragma unroll
      for (int t = 0; t < S / 16; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          st[t][r] = __builtin_amdgcn_exp2f(fmaf(st[t][r], cs, mc));
          if constexpr (SM != 1) rs += st[t][r];
        }
      } else {
        // TIMING EXPERIMENT ONLY ("lvit.shape" 8 / 9, results invalid): SM = 8 drops the softmax's exp / sum (keeps the max), SM = 9 drops the whole
        // softmax -- the raw scores go to the PV product.  What the kernel's time does when 60 % / 75 % of the attention loop's vector instructions
        // disappear says whether it is bound by them (profiles/r05_lvit_window_disasm.txt)
        if constexpr (SM == 8) {
#pragma unroll
          for (int t = 0; t < S / 16; ++t) {
            mx = fmaxf(fmaxf(mx, st[t][0]), st[t][1]);
            mx2 = fmaxf(fmaxf(mx2, st[t][2]), st[t][3]);
          }
          rs = col_max(fmaxf(mx, mx2)) + 2.f;
        } else {
          rs = 1.f;
        }
      }
      // SM = 1: the softmax denominator comes off the matrix pipe (idle three quarters of this loop): an all-ones A fragment against the
      // packed probabilities sums the 32 keys of a block for every query -- 8 MFMAs instead of 64 v_add_f32 and a cross-lane reduction, and
      // the sum is over the SAME fp16-rounded probabilities the numerator uses
      floatx4 o[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};
      floatx4 den = {0.f, 0.f, 0.f, 0.f};
      if constexpr (SM == 2) {
        // O^T = V_h^T P^T: the probabilities are packed first (64 score registers -> 32), then the four transposed 8-byte reads of key block kb + 1 are
        // in flight while block kb multiplies
        constexpr int NB = S / 32;
        frag pb[NB];
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) pb[kb] = lv_pack(st[2 * kb], st[2 * kb + 1]);
        const unsigned va = lbase + (unsigned)(VOFF + hl * (DH * 2) + vlane);
        lv_u32x2 V[2][4];                                       // [ring slot][i * 2 + {lo, hi}]
        auto rd_block = [&](auto kc, lv_u32x2 (&v)[4]) {
          constexpr int kb = decltype(kc)::value;
          lv_rd_tr64<(kb * 32) * KVP>(v[0], va);
          lv_rd_tr64<(kb * 32 + 16) * KVP>(v[1], va);
          lv_rd_tr64<(kb * 32) * KVP + 32>(v[2], va);
          lv_rd_tr64<(kb * 32 + 16) * KVP + 32>(v[3], va);
        };
        __builtin_amdgcn_sched_barrier(0);
        rd_block(std::integral_constant<int, 0>{}, V[0]);
        lv_static_for<0, NB>([&](auto kc) {
          constexpr int kb = decltype(kc)::value;
          if constexpr (kb + 1 < NB) rd_block(std::integral_constant<int, kb + 1>{}, V[(kb + 1) & 1]);
          lv_u32x2 (&v)[4] = V[kb & 1];
          lv_wait2<(kb + 1 < NB ? 6 : 2)>(v[0], v[1]);
          {
            frag f;
            const lv_u32x2 lo = v[0], hi = v[1];
            __builtin_memcpy(&f, &lo, 8);
            __builtin_memcpy(reinterpret_cast<unsigned char*>(&f) + 8, &hi, 8);
            o[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f, pb[kb], o[0], 0, 0, 0);
          }
          lv_wait2<(kb + 1 < NB ? 4 : 0)>(v[2], v[3]);
          {
            frag f;
            const lv_u32x2 lo = v[2], hi = v[3];
            __builtin_memcpy(&f, &lo, 8);
            __builtin_memcpy(reinterpret_cast<unsigned char*>(&f) + 8, &hi, 8);
            o[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f, pb[kb], o[1], 0, 0, 0);
          }
        });
        lv_settle();
      } else {
#pragma unroll
        for (int kb = 0; kb < S / 32; ++kb) {
          const frag pb = lv_pack(st[2 * kb], st[2 * kb + 1]);
          if constexpr (SM == 1) den = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, pb, den, 0, 0, 0);
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const half4 lo = lv_read_tr4(Vh + (kb * 32) * KVP + i * 32);
            const half4 hi = lv_read_tr4(Vh + (kb * 32 + 16) * KVP + i * 32);
            const frag va = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            o[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(va, pb, o[i], 0, 0, 0);
          }
        }
      }
      // (round 5, measured and reverted: the V region's base kept opaque in a register so that all 32 transposed reads are base + immediate -- 29 -> 3
      // v_add_u32 per head -- and v_rcp_f32 instead of the IEEE division -- 10 -> 1 instructions: 330 -> 314 vector instructions per head and token
      // tile, 357.6 against 355.5 us for 1536 windows: no gain; the kernel is not bound by its vector-instruction count alone)
      const float inv = 1.f / (SM == 1 ? den[0] : col_sum(rs));   // every row of `den` is the column (query) sum
      att[j] = lv_pack(o[0] * inv, o[1] * inv);              // rows d >= 24 of O^T are another head's values: W_p's columns for them are zero
    }
    // x += W_p[:, h] O_h
    frag G[2][3];
    load_r2(buf, 0, G[0]);
    load_r2(buf, 1, G[1]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ig = 0; ig < 2; ++ig)
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[ig * 3 + k][j] = Mma<T>::mma(G[ig][k], att[j], acc[ig * 3 + k][j]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }

  const unsigned long long tk4 = now();
  // ---- y1 = x + W2a relu(W1a LN2(x) + b1a) + b2a;  y2 = y1 + W2b relu(W1b y1 + b1b) + b2b (as k_mlp2, 32 hidden units a chunk) ----
  layer_norm_to_xb(a.ln2_g, a.ln2_b);
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    const floatx4 bb = *reinterpret_cast<const floatx4*>(a.b2a + i * 16 + 4 * h);
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] += bb;
  }
  // one 32-unit chunk: fragments at wb (R1 then R2), its 128 bytes of bias at bb
  auto mlp_body = [&](const unsigned char* wb, const unsigned char* bb, bool first) {
    frag F[2][3], G[2][3];
    floatx4 bia[2];
    load_r1(wb, 0, F[0]);
    bia[0] = *reinterpret_cast<const floatx4*>(bb + 16 * h);
    load_r1(wb, 1, F[1]);
    bia[1] = *reinterpret_cast<const floatx4*>(bb + 16 * h + 64);
    if (first) refill();
    __builtin_amdgcn_sched_barrier(0);
    floatx4 hacc[2][TM];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int j = 0; j < TM; ++j) hacc[u][j] = bia[u];
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int j = 0; j < TM; ++j) hacc[u][j] = Mma<T>::mma(F[u][k], xb[k][j], hacc[u][j]);
      if (u == 0) {
        load_r2(wb, 0, G[0]);
        load_r2(wb, 1, G[1]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    frag hb[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const half8 v = lv_pack(hacc[0][j], hacc[1][j]);
      hb[j] = __builtin_elementwise_max(v, Mma<T>::zero());
    }
#pragma unroll
    for (int ig = 0; ig < 2; ++ig)
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[ig * 3 + k][j] = Mma<T>::mma(G[ig][k], hb[j], acc[ig * 3 + k][j]);
  };
  auto mlp_chunk = [&](int t) {
    const unsigned char* buf = dbl ? begin_mlp_single(t) : begin_chunk(t);
    mlp_body(buf, buf + R3, true);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  // stage b (mlp_head): its input is the stage-a result, which becomes the new residual
  auto stage_switch = [&]() {
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int c = 0; c < NCH; ++c) xb[c][j] = lv_pack(acc[c * 2][j], acc[c * 2 + 1][j]);
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const floatx4 bb = *reinterpret_cast<const floatx4*>(a.b2b + i * 16 + 4 * h);
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] += bb;
    }
  };
  unsigned long long tk5 = 0;
  if (dbl) {
    mlp_chunk(t0);
    mlp_chunk(t0 + 1);
#pragma unroll 1
    for (int k = 0; k < ndbl; ++k) {
      if (2 + 2 * k == nhc) {
        tk5 = now();
        stage_switch();
      }
      const unsigned char* buf = begin_double(k);
      mlp_body(buf, buf + 2 * (N1 + N2) * 1024, true);
      mlp_body(buf + (N1 + N2) * 1024, buf + 2 * (N1 + N2) * 1024 + 128, false);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  } else {
#pragma unroll 1
    for (int t = 0; t < nhc; ++t) mlp_chunk(t0 + t);
    tk5 = now();
    stage_switch();
#pragma unroll 1
    for (int t = nhc; t < 2 * nhc; ++t) mlp_chunk(t0 + t);
  }
  const unsigned long long tk6 = now();

  // ---- fold + window join into the output map ----
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int t = tok[j];
    const int y0 = wy * a.ws + (t / tw) * a.p, x0 = wx * a.ws + (t % tw) * a.p;
    // adjacent feature tiles leave as one 16-byte store per lane (cfen_common.hpp pair_tiles16): 8 consecutive features = 8 channels of one patch pixel (C = 24)
#pragma unroll
    for (int i = 0; i < ND; i += 2) {
      const int f = i * 16 + 16 * (h & 1) + 8 * (h >> 1);
      const int ij = f / a.C, c = f - ij * a.C;
      T* dst = (T*)a.out + (((size_t)b * a.H + y0 + ij / a.p) * a.W + x0 + ij % a.p) * a.cs_out + c;
      *reinterpret_cast<uint4*>(dst) = pair_tiles16(acc[i][j], acc[i + 1][j]);
    }
  }
  if (stamps) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long tk7 = now(), rt1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && blockIdx.z == 0 && lane == 0 && wave < 4) {
      unsigned long long* o = stamps + wave * 8;
      o[0] = tk1 - tk0; o[1] = tk2 - tk1; o[2] = tk3 - tk2; o[3] = tk4 - tk3; o[4] = tk5 - tk4; o[5] = tk6 - tk5; o[6] = tk7 - tk6; o[7] = rt1 - rt0;
    }
  }
}

}  // namespace

int& cfen_tune_lvit_debug() {
  static int v = 0;
  return v;
}

int& cfen_tune_lvit_shape() {   // 2 (default): 16 waves x 1 token tile (four waves per SIMD, 128 registers, no spills: 2.858 -> 2.83 ms -- a wave issues a vector
                                // instruction every ~10 cycles and two thirds of this kernel are vector-instruction bound); 0: 8 waves x 2 token tiles;
                                // 1: 4 waves x 4 token tiles (one wave per SIMD, 512 registers)
  static int v = 2;
  return v;
}

bool cfen_lvit_window_supported(int dtype, int D, int heads, int S, int hidden) {
  return dtype == 1 && D == 96 && heads == 4 && S == 256 && hidden > 0 && hidden % 32 == 0;
}

int cfen_lvit_window_impl_g(int dtype, int ng, const LvitArgs* ap, hipStream_t s) {
  CFEN_CHECK_ARG(ng >= 1 && ng <= CFEN_MAX_GROUPS && ap, "lvit_window: 1..%d problems per launch", CFEN_MAX_GROUPS);
  Grouped<LvitArgs> ga;
  for (int g = 0; g < CFEN_MAX_GROUPS; ++g) ga.g[g] = ap[g < ng ? g : 0];
  for (int g = 0; g < ng; ++g) {
    const LvitArgs& a = ap[g];
    const int D = a.p * a.p * a.C, tw = a.p ? a.ws / a.p : 0;
    CFEN_CHECK_ARG(a.p > 0 && a.ws > 0 && a.ws % a.p == 0 && cfen_lvit_window_supported(dtype, D, 4, tw * tw, a.Hm),
                   "lvit_window: fp16, C = 24, p = 2, 32-pixel windows (256 tokens of dim 96, 4 heads), hidden %% 32 == 0 only");
    CFEN_CHECK_ARG(a.B > 0 && a.H % a.ws == 0 && a.W % a.ws == 0 && a.cs_in >= a.C && a.cs_out >= a.C && a.cs_in % 4 == 0 && a.cs_out % 8 == 0 && a.C % 8 == 0 &&
                   cfen_aligned16(a.out), "lvit_window: bad map geometry (the output map takes 16-byte stores: C and its channel stride multiples of 8)");
    CFEN_CHECK_ARG(a.fmap && a.out && a.Ws && a.be && a.pos && a.ln1_g && a.ln1_b && a.ln2_g && a.ln2_b && a.b1a && a.b2a && a.b1b && a.b2b,
                   "lvit_window: null pointer");
    CFEN_CHECK_ARG(cfen_aligned16(a.fmap) && cfen_aligned16(a.out) && cfen_aligned16(a.Ws) && cfen_aligned16(a.be) && cfen_aligned16(a.pos) &&
                   cfen_aligned16(a.ln1_g) && cfen_aligned16(a.ln1_b) && cfen_aligned16(a.ln2_g) && cfen_aligned16(a.ln2_b) && cfen_aligned16(a.b1a) &&
                   cfen_aligned16(a.b2a) && cfen_aligned16(a.b1b) && cfen_aligned16(a.b2b), "lvit_window: pointers must be 16-byte aligned");
    CFEN_CHECK_ARG(a.B == ap[0].B && a.H == ap[0].H && a.W == ap[0].W && a.Hm == ap[0].Hm, "lvit_window: grouped problems must have the same shape");
  }
  const long long blocks = (long long)ap[0].B * (ap[0].H / ap[0].ws) * (ap[0].W / ap[0].ws);
  CFEN_CHECK_ARG(blocks > 0 && blocks < (1ll << 31), "lvit_window: bad grid");
  unsigned long long* stamps = nullptr;
  const bool stamping = (cfen_tune_lvit_debug() & 64) != 0;
  if (stamping) {   // timing run: section times of workgroup 0 to stderr after the launch (tools/dbg_lvit_sections.py)
    static unsigned long long* buf = nullptr;
    if (!buf && hipMalloc(&buf, 4 * 8 * sizeof(unsigned long long)) != hipSuccess) buf = nullptr;
    stamps = buf;
    if (stamps) (void)hipMemsetAsync(stamps, 0, 4 * 8 * sizeof(unsigned long long), s);
  }
  if (cfen_tune_lvit_shape() == 2)             // default (round 6): 16 waves x 1 token tile, every refill behind the chunk's first fragment reads
    CFEN_LAUNCH((k_lvit_window<6, 16, 1, 0, 0, 1, 1>), dim3((unsigned)blocks, 1, ng), dim3(1024), 0, s, ga, stamps, cfen_tune_lvit_debug() & 63);
  else if (cfen_tune_lvit_shape() == 12)       // the refill right behind the barrier (rounds 3-5)
    CFEN_LAUNCH((k_lvit_window<6, 16, 1>), dim3((unsigned)blocks, 1, ng), dim3(1024), 0, s, ga, stamps, cfen_tune_lvit_debug() & 63);
  else if (cfen_tune_lvit_shape() == 15)       // double MLP chunks on a ring over the dead K / V tiles (MEASURED: 350 against 374 us for 1536 windows alone, equal in flight; the
    CFEN_LAUNCH((k_lvit_window<6, 16, 1, 0, 0, 2>), dim3((unsigned)blocks, 1, ng), dim3(1024), 0, s, ga, stamps, cfen_tune_lvit_debug() & 63);      // instantiation spills 112-120 bytes a lane at 128 registers)
  else if (cfen_tune_lvit_shape() == 13)       // ... with deferred refills: 452 us (spills in the hot loops)
    CFEN_LAUNCH((k_lvit_window<6, 16, 1, 0, 0, 2, 1>), dim3((unsigned)blocks, 1, ng), dim3(1024), 0, s, ga, stamps, cfen_tune_lvit_debug() & 63);
  else if (cfen_tune_lvit_shape() == 3)
    CFEN_LAUNCH((k_lvit_window<6, 16, 1, 1>), dim3((unsigned)blocks, 1, ng), dim3(1024), 0, s, ga, stamps, cfen_tune_lvit_debug() & 63);
  else if (cfen_tune_lvit_shape() == 4)
    CFEN_LAUNCH((k_lvit_window<6, 16, 1, 2>), dim3((unsigned)blocks, 1, ng), dim3(1024), 0, s, ga, stamps, cfen_tune_lvit_debug() & 63);
  else if (cfen_tune_lvit_shape() == 5)
    CFEN_LAUNCH((k_lvit_window<6, 8, 2, 2>), dim3((unsigned)blocks, 1, ng), dim3(512), 0, s, ga, stamps, cfen_tune_lvit_debug() & 63);
  else if (cfen_tune_lvit_shape() == 6)
    CFEN_LAUNCH((k_lvit_window<6, 16, 1, 0, 1>), dim3((unsigned)blocks, 1, ng), dim3(1024), 0, s, ga, stamps, cfen_tune_lvit_debug() & 63);
  else if (cfen_tune_lvit_shape() == 8)
    CFEN_LAUNCH((k_lvit_window<6, 16, 1, 8>), dim3((unsigned)blocks, 1, ng), dim3(1024), 0, s, ga, stamps, cfen_tune_lvit_debug() & 63);
  else if (cfen_tune_lvit_shape() == 9)
    CFEN_LAUNCH((k_lvit_window<6, 16, 1, 9>), dim3((unsigned)blocks, 1, ng), dim3(1024), 0, s, ga, stamps, cfen_tune_lvit_debug() & 63);
  else if (cfen_tune_lvit_shape() == 1)
    CFEN_LAUNCH((k_lvit_window<6, 4, 4>), dim3((unsigned)blocks, 1, ng), dim3(256), 0, s, ga, stamps, cfen_tune_lvit_debug() & 63);
  else
    CFEN_LAUNCH((k_lvit_window<6, 8, 2>), dim3((unsigned)blocks, 1, ng), dim3(512), 0, s, ga, stamps, cfen_tune_lvit_debug() & 63);
  CFEN_CHECK_LAUNCH("lvit_window");
  if (stamping && stamps && !cfen_recorder()) {
    unsigned long long hst[4 * 8];
    if (hipStreamSynchronize(s) == hipSuccess && hipMemcpy(hst, stamps, sizeof(hst), hipMemcpyDeviceToHost) == hipSuccess)
      for (int w = 0; w < 4; ++w) {
        const unsigned long long* o = hst + w * 8;
        fprintf(stderr, "lvit_window stamps wave %d: prologue %llu, embedding %llu, LN1 + K/V %llu, attention %llu, LN2 + MLP a %llu, MLP b %llu, fold %llu cyc; %.1f us in all\n", w, o[0], o[1], o[2], o[3],
                o[4], o[5], o[6], (double)o[7] / 100.0);
      }
  }
  return CFEN_OK;
}
