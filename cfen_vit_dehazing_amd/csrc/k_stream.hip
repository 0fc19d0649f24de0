// Token blocks on a FRAGMENT STREAM (round 3): the fused kernels for the embedding dims the register-resident kernels of k_mlp.hip /
// k_embed.hip could not hold at two waves per SIMD -- D = 384 (LViT level 3, GViT level 1) -- and a spill-free D = 192 variant.
//
//   k_mlp3   x (+ Wp att) -> LN2 -> FFN -> +res -> mlp_head -> +res -> fold      (v3:1386-1389, 1173, 1186; replaces proj / ln2_ffn1 / ffn2 /
//                                                                                head1 / head2_fold = five k_gemm_dma launches and the
//                                                                                4 D-wide hidden activations they passed through HBM)
//   k_front3 patch gather -> linear_encoding + res + pos -> X1, LN1 -> qkv (head-major)   (v3:1140-1143, 1166, 1364-1371)
//
// What is different from k_mlp2 / k_embed_qkv2:
//   * ONE wave per SIMD with the whole 512-register file (4 waves a workgroup): the residual stream of 32 tokens x 384 features (or
//     64 x 192) lives in 192 accumulator registers, its fp16 copy (the B operand) in 96 more -- no spills, and every weight fragment read
//     from LDS feeds TM = 2 (4) MFMAs instead of 1, which is what an 8-wave / 256-register shape at D = 384 would be limited to.
//   * the weights are a FRAGMENT STREAM, packed once on the host (packing.pack_stream_*): the 1 KiB A-operand fragments (16 rows x 32 k,
//     lane l = 16 bytes of row l & 15, k quarter l >> 4) in exactly the order the kernel consumes them.  A "phase" is ND consecutive
//     fragments = one k-chunk of Wp, one 32-unit slice of W1, or the matching slice of W2; HBM / L2 reads are one linear run per
//     workgroup, the LDS image is lane-linear (conflict-free ds_read_b128, no padding, no swizzle), and the ring of R phases is filled by
//     LDS-DMA with a counted vmcnt: R - 2 phases stay in flight across the one raw s_barrier per phase.
#include <stdlib.h>
#include <type_traits>
#include "cfen_common.hpp"
#include "cfen_internal.hpp"
#include "cfen_mlp.hpp"

namespace {

template <int I, int N, class F>
CFEN_DEV void sfor(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sfor<I + 1, N>(f);
  }
}

// (a __device__ helper, not the builtin inside a lambda: hipcc's host pass drops the stub of a kernel whose lambda calls it)
CFEN_DEV void st_dma(const void* g, unsigned char* l) {
  __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// the same with a wave-uniform source base in SGPRs + one per-lane byte offset (no 64-bit per-lane address registers: the spread refills of k_mlp3 issue
// one piece at a time from inside the MFMA stream, where every live VGPR is taken) and a wave-uniform LDS byte address.  M0 is compiler-reserved and not
// preserved around an asm statement: saved and restored here (cdna_hip_programming.md, "LDS-DMA recipe").  Invisible to hipcc's vmcnt bookkeeping, like every
// wait of the ring.
CFEN_DEV void st_dma_s(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}

// wait until at most `younger` phases of DPW LDS-DMAs each are still in flight (younger <= Y)
template <int DPW, int Y>
CFEN_DEV void wait_phases(int younger) {
  if constexpr (Y == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    if (younger >= Y) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Y * DPW) : "memory");
    else wait_phases<DPW, Y - 1>(younger);
  }
}

// LDS fragment reads and their waits by hand.  hipcc's own waits around a "read the next group, multiply the current one" loop came out as
// s_waitcnt lgkmcnt(0) right behind the freshly issued reads (every second group waited a full LDS round trip: the kernel ran 137 us with
// its DMA refills removed against 65 us of MFMA time).  Inline-asm reads are invisible to the compiler's counter, so EVERY LDS read between a
// phase's barrier and its end goes through these two helpers and the waits are counted here: LDS returns in order, read f is complete once
// at most (reads issued after f) are outstanding; the "+v" operand keeps the MFMAs that consume the register behind the wait.
template <int OFF>
CFEN_DEV void lds_rd(half8& f, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(addr), "n"(OFF));
}
template <int OFF>
CFEN_DEV void lds_rd(floatx4& f, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(addr), "n"(OFF));
}
template <int N>
CFEN_DEV void lds_wait(half8& f) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N));
}
template <int N>
CFEN_DEV void lds_wait(half8& f, floatx4& b0, floatx4& b1) {
  asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(f), "+v"(b0), "+v"(b1) : "n"(N));
}

// MFMA result -> VALU reader with inline asm in between: hipcc's hazard recognizer does not carry the matrix pipe's write-back latency
// across an asm statement (found the hard way: k_front3's row-major qkv tiles of the second token tile came out short of their last
// k-chunk).  Wherever vector code reads an accumulator whose last MFMA sits behind one of this file's asm reads / waits, this pins 12
// wait states (8-pass MFMA -> VALU read) between them; the sched_barriers keep the MFMAs above and the readers below.
CFEN_DEV void mfma_results_settle() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 11" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// timing experiments: a value whose store is left out stays live (its producers are not dead code)
CFEN_DEV void keep_live(const uint4& v) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 vv = {v.x, v.y, v.z, v.w};
  asm volatile("" ::"v"(vv));
}

CFEN_DEV half8 pack_pair(const floatx4& a, const floatx4& b) {
  half8 f = {(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3], (half_t)b[0], (half_t)b[1], (half_t)b[2], (half_t)b[3]};
  return f;
}

// ND = D / 16 feature tiles, TM token tiles of 16 per wave, R ring slots, HB = largest hidden width (bias staging area).
// A ring slot holds a DOUBLE phase = 2 ND fragments: the W1 slice AND the W2 slice of one 32-unit hidden sub-step (or two k-chunks of Wp).
// Measured with single phases (one barrier per ND fragments): 0.63 us a phase whether it held 12 or 24 KiB -- barrier skew, the refill of the
// fragment pipeline and the DMA issue cost ~0.3 us each time, as much as the 48 MFMAs of the phase; one barrier per sub-step halves that.
// DBG (timing experiments only, results invalid), bit flags: 1 = no LDS-DMA refills after the prologue, 2 = no MFMAs in the hidden loop, 4 = no LDS fragment reads (the MFMAs
// multiply whatever the fragment registers hold), 32 = refills as 4-byte-per-lane pieces (same instruction count, a quarter of the bytes into LDS)
// WPE (round 5): waves per SIMD the kernel is compiled for.  1 = the whole 512-register file for one wave per SIMD (one workgroup a CU); 2 = 256 registers,
// so that TWO workgroups of a short ring (R = 3 slots of 24 KiB at D = 192: 78 KB) share a CU -- k_mlp2's occupancy on k_mlp3's fragment-stream ring.
// NW (round 5): waves per workgroup.  4 = one per SIMD; 8 with TM = 1 and WPE = 2 = the same 128 tokens a workgroup on two waves per SIMD (each covers the
// other's DMA issue and waits) at twice the LDS fragment reads per token -- the D = 384 A/B of this round.
// SPR (round 6): how a phase's DPW LDS-DMA pieces per wave are issued.  0 = one burst in front of the phase's MFMAs (rounds 3-5); 1 = SPREAD, one piece behind every NW-th
// fragment's MFMAs (the texture path then sees one piece per wave per 8 MFMAs instead of 48 pieces at once: the burst cost the issuing = computing wave 60-185 cycles a
// piece, MI355X_MICROARCH.md "LDS-DMA piece issue cost"); 2 = spread and staggered by wave (wave w issues behind fragment 4k + w).
// STAMP (timing build, results valid but ~5-10 % slower): per-wave cycle sums of [wait for the phase's DMA | barrier | phase body] of workgroup 0 -> `stamps`.
// UMAJ (round 6): the W1 fragments of a sub-step are CONSUMED tile by tile (u slowest) instead of k-chunk by k-chunk: the first hidden tile's results are converted to
// fp16 / ReLU'd between the MFMAs of the second tile, so that only half of the repack sits in the MFMA pipe's idle gap between the two halves of a sub-step (one wave per
// SIMD: nothing else fills that gap; tools/repro/mfma_rate_probe.hip prices the whole repack at 3.2 cycles per MFMA of the phase).  Only the read order of the LDS image
// changes -- the stream, the accumulation order of every chain and the results stay bit for bit.
template <int ND, int TM, int R, int HB, int DBG = 0, int WPE = 1, int NW = 4, int PDX = 0, int SPR = 0, int STAMP = 0, int UMAJ = 0>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_mlp3(Grouped<Mlp3Args> ga, unsigned long long* stamps) {
  typedef half_t T;
  typedef half8 frag;
  const Mlp3Args a = ga.g[blockIdx.z];
  constexpr int D = ND * 16, NCH = ND / 2, NF = 2 * ND, SLOT = NF * 1024, DPW = NF / NW, RING = R * SLOT;
  static_assert(NF % NW == 0 && NCH % 2 == 0 && R >= 3 && (R - 2) * DPW < 64 && HB % 256 == 0, "ring geometry");
  static_assert(RING + 2 * HB * 4 <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[RING + 2 * HB * 4];

  const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, h = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long tok0 = ((long long)blockIdx.x * NW + wave) * (TM * 16);
  const int nt = a.H / 32;                      // 32-unit hidden sub-steps per stage = double phases per stage
  const int npp = a.Wp ? NCH / 2 : 0;           // projection double phases
  const int NP = npp + nt * (a.Wb ? 2 : 1);

  auto phase_src = [&](int q) {
    return q < npp        ? (const unsigned char*)a.Wp + (size_t)q * SLOT
           : q < npp + nt ? (const unsigned char*)a.Wa + (size_t)(q - npp) * SLOT
                          : (const unsigned char*)a.Wb + (size_t)(q - npp - nt) * SLOT;
  };
  auto issue = [&](int q, int slot) {
    const unsigned char* src = phase_src(q);
#pragma unroll
    for (int k = 0; k < DPW; ++k) {
      const int f = k * NW + wave;
      if constexpr (DBG & 32) __builtin_amdgcn_global_load_lds(src + f * 1024 + lane * 4, (__attribute__((address_space(3))) void*)(lds + slot * SLOT + f * 1024), 4, 0, 0);
      else st_dma(src + f * 1024 + lane * 16, lds + slot * SLOT + f * 1024);
    }
  };
  // ---- prologue: hidden biases -> LDS, the first R - 1 phases into the ring, tokens into registers ----
  {
    const int pieces = a.H * 4 / 16, nb = (pieces + 63) / 64;
    for (int blk = wave; blk < nb; blk += NW) {
      const int pc = min(blk * 64 + lane, pieces - 1);
      st_dma((const unsigned char*)a.b1a + pc * 16, lds + RING + blk * 1024);
      if (a.Wb) st_dma((const unsigned char*)a.b1b + pc * 16, lds + RING + HB * 4 + blk * 1024);
    }
  }
#pragma unroll
  for (int q = 0; q < R - 1; ++q)
    if (q < NP) issue(q, q);

  floatx4 acc[ND][TM];
  const T* X = (const T*)a.X;
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    long long t = tok0 + j * 16 + r16;
    if (t >= a.M) t = a.M - 1;
    const T* xp = X + t * D + 4 * h;
#pragma unroll
    for (int i = 0; i < ND; ++i) acc[i][j] = load4<T>(xp + i * 16);
  }
  frag xb[NCH][TM];
  if (a.Wp) {   // attention output as the projection's B operand (natural k order), borrowed registers
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      long long t = tok0 + j * 16 + r16;
      if (t >= a.M) t = a.M - 1;
      const T* ap = (const T*)a.A + t * D + h * 8;
#pragma unroll
      for (int c = 0; c < NCH; ++c) xb[c][j] = load_frag<T>(ap + c * 32);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // one drain: biases, ring prologue and tokens have landed

  int p = 0, cur = 0, fill = R - 1;
  // timing build: s_memtime stamps at points where no LDS read is in flight (the scalar load shares lgkmcnt with the hand-counted ds_reads)
  unsigned long long tk0 = 0, tk_wait = 0, tk_bar = 0, tk_body = 0, tk_last = 0, tk_first = 0, rt0 = 0;
  auto now = [&]() -> unsigned long long {
    if constexpr (STAMP) {
      unsigned long long t;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
      return t;
    } else {
      return 0ull;
    }
  };
  if constexpr (STAMP) {
    tk0 = now();
    rt0 = __builtin_amdgcn_s_memrealtime();
  }
  // phase p has landed and is visible to every wave; every wave is done with the slot of phase p - 1
  auto begin = [&]() {
    unsigned long long t0 = 0, t1 = 0;
    if constexpr (STAMP) {
      t0 = now();
      if (tk_last) tk_body += t0 - tk_last;
      else tk_first = t0;
    }
    wait_phases<DPW, R - 2>(min(R - 2, NP - 1 - p));
    if constexpr (STAMP) { t1 = now(); tk_wait += t1 - t0; }
    __builtin_amdgcn_s_barrier();
    if constexpr (STAMP) { tk_last = now(); tk_bar += tk_last - t1; }
  };
  // ... which phase p + R - 1 may now overwrite (called after the phase's first fragment reads are issued)
  const unsigned lbase = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)lds;
  const unsigned char* rbase = nullptr;   // SPR: (wave-uniform) source / LDS offset of the refill that the running phase spreads between its MFMAs
  int rdst = 0;
  bool rgo = false;
  auto refill = [&]() {
    if constexpr (SPR) {
      rgo = !(DBG & 1) && p + R - 1 < NP;
      rbase = phase_src(rgo ? p + R - 1 : 0) + wave * 1024;
      rdst = fill * SLOT + wave * 1024;
    } else {
      if (!(DBG & 1) && p + R - 1 < NP) issue(p + R - 1, fill);
    }
    fill = cur;
    cur = cur + 1 == R ? 0 : cur + 1;
    ++p;
  };
  const unsigned lane16 = lane * 16;
  auto piece = [&](int k) {   // piece k of this wave's DPW: fragment k * NW + wave of the refilled phase
    if (rgo) st_dma_s(rbase + k * (NW * 1024), lane16, lbase + rdst + k * (NW * 1024));
  };
  // LDS byte addresses (the low 32 bits of a generic pointer into LDS are its LDS offset)
  const unsigned lfrag = lbase + lane * 16;     // + slot * SLOT + fragment * 1024
  constexpr int PD = PDX ? PDX : (TM >= 4 ? 4 : 6), NB = PD + 2;   // fragment reads in flight ahead of the MFMAs (>= 256 MFMA cycles of cover) / registers of the fragment ring
  static_assert(PD < NB && PD <= NF, "fragment ring");
  // one double phase: NF fragments, each consumed by body(index, fragment); `pre` runs once the first PD reads are issued (the DMA refill)
  // (`um`: the first ND fragments are read u-major -- consumption index g = u * NCH + c is fragment 2 c + u of the LDS image)
  auto phase = [&](unsigned sa, auto&& pre, auto&& body, auto um) {
    constexpr bool UM = decltype(um)::value;
    frag F[NB];
    if constexpr (DBG & 4) {
#pragma unroll
      for (int i = 0; i < NB; ++i) F[i] = xb[i % NCH][0];
    }
    sfor<0, PD>([&](auto fc) {
      constexpr int f = decltype(fc)::value, lf = (UM && f < ND) ? 2 * (f % NCH) + f / NCH : f;
      if constexpr (!(DBG & 4)) lds_rd<lf * 1024>(F[f % NB], sa);
    });
    pre();
    if constexpr (SPR == 5) {
#pragma unroll
      for (int k = 0; k < DPW / 3; ++k) piece(k);
    }
    sfor<0, NF>([&](auto fc) {
      constexpr int f = decltype(fc)::value;
      if constexpr (!(DBG & 4)) {
        constexpr int g = f + PD, lg = (UM && g < ND) ? 2 * (g % NCH) + g / NCH : g;
        if constexpr (g < NF) lds_rd<lg * 1024>(F[g % NB], sa);
        lds_wait<(f + PD < NF ? PD : NF - 1 - f)>(F[f % NB]);
      }
      body(fc, F[f % NB]);
      if constexpr ((SPR == 1 && f % NW == 1) || (SPR == 3 && f % NW == 3) || (SPR == 4 && f % NW == 0)) piece(f / NW);
      if constexpr (SPR == 2) {
        if (f % NW == wave) piece(f / NW);
      }
      if constexpr (SPR == 5 && f == ND - 1) {
#pragma unroll
        for (int k = DPW / 3; k < 2 * (DPW / 3); ++k) piece(k);
      }
      if constexpr (SPR == 5 && f >= ND && (f - ND) % (ND / (DPW / 3)) == ND / (DPW / 3) - 2) piece(2 * (DPW / 3) + (f - ND) / (ND / (DPW / 3)));
    });
  };

  // ---- x += Wp att (out_proj + residual, v3:1386): double phase q = k-chunks 2q, 2q + 1 of all ND feature tiles ----
  if (a.Wp) {
    sfor<0, NCH / 2>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      begin();
      phase(lfrag + cur * SLOT, [&]() { refill(); }, [&](auto fc, const frag& fr) {
        constexpr int f = decltype(fc)::value, c = 2 * q + f / ND, i = f % ND;
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = Mma<T>::mma(fr, xb[c][j], acc[i][j]);
      }, std::false_type{});
    });
  }

  // ---- stage-a input: LayerNorm(x) (or x) as B fragments; residual + output bias go into the accumulators ----
  mfma_results_settle();
  if (a.ln_g) {
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
        xb[c][j] = pack_pair(t[0], t[1]);
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int c = 0; c < NCH; ++c) xb[c][j] = pack_pair(acc[c * 2][j], acc[c * 2 + 1][j]);
  }
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    const floatx4 bb = *reinterpret_cast<const floatx4*>(a.b2a + i * 16 + 4 * h);
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] += bb;
  }
  // the LayerNorm parameter / bias loads above are ordinary VMEM loads in front of the ring's counted waits: drain them once (the ring
  // is R - 1 phases ahead, so this costs nothing), from here on only LDS-DMAs are in flight
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // one 32-unit hidden sub-step: hidden = W1 slice . xb + b1 (fragments f < ND: f = 2 c + u), then y += W2[:, slice] . relu(hidden)
  // (fragments f >= ND: feature tile f - ND)
  auto substep = [&](int t, int stage) {
    begin();
    floatx4 bv0, bv1;
    const unsigned ba = lbase + RING + stage * (HB * 4) + (t * 32 + 4 * h) * 4;
    lds_rd<0>(bv0, ba);
    lds_rd<64>(bv1, ba);
    floatx4 hacc[2][TM];
    frag hb[TM];
    typedef _Float16 half4v __attribute__((ext_vector_type(4)));
    half4v hlo[TM];
    constexpr bool EARLY = UMAJ && NCH + 1 + 2 * TM <= ND;      // room for one early conversion step per token tile between the MFMAs of the second hidden tile
    auto cvt_relu = [&](const floatx4& v) {
      half4v r = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
      half4v z = {(half_t)0, (half_t)0, (half_t)0, (half_t)0};
      return __builtin_elementwise_max(r, z);
    };
    phase(lfrag + cur * SLOT, [&]() { refill(); }, [&](auto fc, frag& fr) {
      constexpr int f = decltype(fc)::value;
      if constexpr (f == 0) lds_wait<(DBG & 4) ? 0 : PD>(fr, bv0, bv1);   // (the bias reads are older than every fragment read: landed with fragment 0)
      if constexpr (f < ND) {
        constexpr int u = UMAJ ? f / NCH : f % 2, c = UMAJ ? f % NCH : f / 2;      // UMAJ: tile by tile; else (c, u) order, u fastest: four independent accumulation chains in rotation
        if constexpr (c == 0) {
#pragma unroll
          for (int j = 0; j < TM; ++j) hacc[u][j] = u ? bv1 : bv0;
        }
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          if constexpr (DBG & 2) hacc[u][j][0] += (float)fr[0] * (float)xb[c][j][0];
          else hacc[u][j] = Mma<T>::mma(fr, xb[c][j], hacc[u][j]);
        }
        if constexpr (EARLY && f >= NCH + 1 && (f - NCH - 1) % 2 == 0 && (f - NCH - 1) / 2 < TM) {
          // tile 0 of token tile j is complete (its last MFMA is >= 4 MFMAs back): convert it now, pinned between the MFMAs of tile 1
          constexpr int j = (f - NCH - 1) / 2;
          __builtin_amdgcn_sched_barrier(0);
          hlo[j] = cvt_relu(hacc[0][j]);
          asm volatile("" : "+v"(hlo[j]));      // materialise it HERE (volatile asm keeps its place among the fragment reads / waits): hipcc sinks the conversion to its use otherwise
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        if constexpr (f == ND) {
          mfma_results_settle();
#pragma unroll
          for (int j = 0; j < TM; ++j) {
            if constexpr (EARLY) {
              const half4v hi = cvt_relu(hacc[1][j]);
              hb[j] = half8{hlo[j][0], hlo[j][1], hlo[j][2], hlo[j][3], hi[0], hi[1], hi[2], hi[3]};
            } else {
              const half8 v = pack_pair(hacc[0][j], hacc[1][j]);
              half8 z;
#pragma unroll
              for (int e = 0; e < 8; ++e) z[e] = (half_t)0;
              hb[j] = __builtin_elementwise_max(v, z);
            }
          }
        }
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          if constexpr (DBG & 2) acc[f - ND][j][0] += (float)fr[0] * (float)hb[j][0];
          else acc[f - ND][j] = Mma<T>::mma(fr, hb[j], acc[f - ND][j]);
        }
      }
    }, std::integral_constant<bool, UMAJ != 0>{});
  };
#pragma unroll 1
  for (int t = 0; t < nt; ++t) substep(t, 0);
  mfma_results_settle();
  if (a.Wb) {   // stage b (mlp_head): its input is the stage-a result, which becomes the new residual
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int c = 0; c < NCH; ++c) xb[c][j] = pack_pair(acc[c * 2][j], acc[c * 2 + 1][j]);
    // (b2b is added in the epilogue: a global load here would sit in front of the ring's counted waits)
#pragma unroll 1
    for (int t = 0; t < nt; ++t) substep(t, 1);
    mfma_results_settle();
  }

  unsigned long long tk_loop_end = 0;
  if constexpr (STAMP) {
    tk_loop_end = now();
    tk_body += tk_loop_end - tk_last;
  }
  // ---- epilogue: (+ b2b) token-major store, or fold + window join into the NHWC map ----
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    floatx4 bb = floatx4{0.f, 0.f, 0.f, 0.f};
    if (a.Wb) bb = *reinterpret_cast<const floatx4*>(a.b2b + i * 16 + 4 * h);
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] += bb;
  }
#pragma unroll
  // adjacent feature tiles leave as one 16-byte store per lane (pair_tiles16: 8 consecutive features = 8 channels of one patch pixel, C is a multiple of 8)
  for (int j = 0; j < TM; ++j) {
    const long long t = tok0 + j * 16 + r16;
    const bool live = t < a.M;                         // (every lane takes part in the exchange)
    const long long tc = live ? t : 0;
    const int f0 = 16 * (h & 1) + 8 * (h >> 1);
    if (!a.fmap) {
      T* yp = (T*)a.Y + tc * D + f0;
#pragma unroll
      for (int i = 0; i < ND; i += 2) {
        const uint4 v = pair_tiles16(acc[i][j], acc[i + 1][j]);
        if (live) *reinterpret_cast<uint4*>(yp + i * 16) = v;
      }
    } else {
      const int tw = a.ws / a.p, S = tw * tw;
      const int nwx = a.mapW / a.ws, nwy = a.mapH / a.ws;
      const int tt = (int)(tc % S);
      const long long wi = tc / S;
      const int wx = (int)(wi % nwx), wy = (int)((wi / nwx) % nwy);
      const long long b = wi / ((long long)nwx * nwy);
      const int y0 = wy * a.ws + (tt / tw) * a.p, x0 = wx * a.ws + (tt % tw) * a.p;
#pragma unroll
      for (int i = 0; i < ND; i += 2) {
        const int f = i * 16 + f0;
        const int ij = f / a.C, c = f - ij * a.C;
        const int pi = ij / a.p, pj = ij - pi * a.p;
        T* dst = (T*)a.fmap + ((b * a.mapH + y0 + pi) * a.mapW + x0 + pj) * a.cs + c;
        const uint4 v = pair_tiles16(acc[i][j], acc[i + 1][j]);
        if (live) *reinterpret_cast<uint4*>(dst) = v;
      }
    }
  }
  if constexpr (STAMP) {   // workgroup 0 of problem 0: [wave][total, wait, barrier, body, phases, prologue, epilogue, 100 MHz ticks] (cycles)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t_end = now(), rt1 = __builtin_amdgcn_s_memrealtime();
    if (stamps && blockIdx.x == 0 && blockIdx.z == 0 && lane == 0) {
      unsigned long long* o = stamps + wave * 8;
      o[0] = t_end - tk0; o[1] = tk_wait; o[2] = tk_bar; o[3] = tk_body; o[4] = (unsigned long long)NP; o[5] = tk_first - tk0; o[6] = t_end - tk_loop_end; o[7] = rt1 - rt0;
    }
  }
}

// fragment offsets of k_mlp3p's runs (compile-time: a local class cannot carry a member template)
template <int ND> struct Mlp3pOffProj { template <int k> static constexpr int at() { return (k / (ND / 2)) * ND + k % (ND / 2); } };
struct Mlp3pOffW1 { template <int k> static constexpr int at() { return 2 * k; } };
struct Mlp3pOffW2 { template <int k> static constexpr int at() { return k; } };

// ---- k_mlp3p (round 6): the same chain at D = 384 on TWO waves per SIMD ----------------------------------------------------------------------------
// k_mlp3<24, 2, ...> runs one wave per SIMD because 384 features x 32 tokens of fp32 residual + its fp16 copy need 288 registers; its phase is then the SUM of MFMA time, refill
// issue, repack gap and fragment-read stalls (DESIGN 6.1 stamps).  Here a wave PAIR (waves q and q + 4: the same SIMD) owns the 32 tokens of pair q:
//   role r = wave >> 2 holds the residual of feature tiles 12 r .. 12 r + 11 (96 accumulator registers) and the WHOLE fp16 B operand (96 registers): ~240 in all, two waves a SIMD;
//   of a 32-unit hidden sub-step role r computes hidden tile r (the fragments (c, u = r) of the W1 slice), the two tiles are exchanged through 8 KB of LDS (own tile out, one
//   workgroup barrier, partner's tile in) and role r adds W2[its 192 features, the 32 units] . relu(h) -- LDS fragment reads and MFMAs per SIMD are those of k_mlp3, but the refill
//   issue, the repack and the read stalls of one wave sit under the other wave's MFMAs;
//   LayerNorm sums cross the pair through the same 8 KB; the fp16 operand (LN(x), then y1) crosses in two rounds through the ring slot that is free at that moment.
// Same fragment streams, same arithmetic per element except the LayerNorm sums' association (own 192 features + the partner's): not bit-equal to k_mlp3, equal to tolerance.
// Only workgroup barriers -- no spin waits.  LDS: 3 double-phase slots (144 KB) + the current stage's hidden biases (6 KB) + the exchange area (8 KB).
// Measured and dropped (profiles/r06_mlp3_pair_variants.txt; the code is in the history): other placements of the refill pieces (split over both halves, spread between the MFMAs,
// all on role 0 behind its last MFMA of either half), s_setprio on either role, deeper fragment read-ahead, and a skewed pipeline with one barrier per sub-step.
template <int ND, int R, int HB, int STAMP = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_mlp3p(Grouped<Mlp3Args> ga, unsigned long long* stamps) {
  typedef half_t T;
  typedef half8 frag;
  typedef _Float16 half4v __attribute__((ext_vector_type(4)));
  const Mlp3Args a = ga.g[blockIdx.z];
  constexpr int TM = 2, NW = 8, NHT = ND / 2, D = ND * 16, NCH = ND / 2, NF = 2 * ND, SLOT = NF * 1024, DPW = NF / NW, RING = R * SLOT, BIAS = HB * 4, EXCH = NW * 1024;
  static_assert(ND % 4 == 0 && NF % NW == 0 && R >= 3 && (R - 2) * DPW < 64 && NCH % 2 == 0, "ring geometry");
  static_assert(RING + BIAS + EXCH <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[RING + BIAS + EXCH];

  const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, h = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q4 = wave & 3, role = wave >> 2;                       // token pair, feature / hidden-tile half
  const long long tok0 = ((long long)blockIdx.x * 4 + q4) * (TM * 16);
  const int nt = a.H / 32;
  const int npp = NCH / 2;
  const int NP = npp + 2 * nt;

  auto phase_src = [&](int q) {
    return q < npp        ? (const unsigned char*)a.Wp + (size_t)q * SLOT
           : q < npp + nt ? (const unsigned char*)a.Wa + (size_t)(q - npp) * SLOT
                          : (const unsigned char*)a.Wb + (size_t)(q - npp - nt) * SLOT;
  };
  auto issue = [&](int q, int slot) {
    const unsigned char* src = phase_src(q);
#pragma unroll
    for (int k = 0; k < DPW; ++k) {
      const int f = k * NW + wave;
      st_dma(src + f * 1024 + lane * 16, lds + slot * SLOT + f * 1024);
    }
  };
  auto stage_bias = [&](const float* b1) {       // H * 4 bytes of hidden bias -> LDS (1 KiB pieces; the caller drains)
    const int pieces = a.H * 4 / 16, nb = (pieces + 63) / 64;
    for (int blk = wave; blk < nb; blk += NW) {
      const int pc = min(blk * 64 + lane, pieces - 1);
      st_dma((const unsigned char*)b1 + pc * 16, lds + RING + blk * 1024);
    }
  };
  unsigned long long tk0 = 0, rt0 = 0;
  if constexpr (STAMP) {
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tk0)::"memory");
    rt0 = __builtin_amdgcn_s_memrealtime();
  }
  auto now = [&]() -> unsigned long long {
    unsigned long long tt = 0;
    if constexpr (STAMP) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt)::"memory");
    return tt;
  };
  unsigned long long sec[7] = {0, 0, 0, 0, 0, 0, 0};      // STAMP: section ends -- prologue loads | projection | LayerNorm + operand exchange | stage a | stage switch | stage b | (epilogue = the end)
  // ---- prologue: stage-a biases, the first R - 1 phases, own feature half of x, the whole attention row (projection B operand) ----
  stage_bias(a.b1a);
#pragma unroll
  for (int q = 0; q < R - 1; ++q)
    if (q < NP) issue(q, q);
  floatx4 acc[NHT][TM];
  // the fp16 B operand, all 12 k-chunks: xo = the chunks of this wave's OWN feature half (k-chunks 6 r .. 6 r + 5), xp = the partner's.  Two arrays with compile-time indices: a
  // role-dependent index into one array sends it to scratch (hipcc merges the two role branches into a dynamic index).  During the projection they simply hold the attention
  // row's chunks 0 .. 5 / 6 .. 11.
  frag xo[NCH / 2][TM], xp[NCH / 2][TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    long long t = tok0 + j * 16 + r16;
    if (t >= a.M) t = a.M - 1;
    const T* xrow = (const T*)a.X + t * D + (role * NHT) * 16 + 4 * h;
#pragma unroll
    for (int i = 0; i < NHT; ++i) acc[i][j] = load4<T>(xrow + i * 16);
    // the attention row (the projection's B operand): this wave's OWN k-chunks only; the partner's half crosses through LDS below (both roles loading the whole row was 98 KB
    // more per workgroup in a prologue that runs at the cold-burst rate of a CU, ~14 B/cycle)
    const T* ap = (const T*)a.A + t * D + (role * (NCH / 2)) * 32 + h * 8;
    if (role == 0) {
#pragma unroll
      for (int c = 0; c < NCH / 2; ++c) xo[c][j] = load_frag<T>(ap + c * 32);
    } else {
#pragma unroll
      for (int c = 0; c < NCH / 2; ++c) xp[c][j] = load_frag<T>(ap + c * 32);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  sec[0] = now();

  int p = 0, cur = 0, fill = R - 1;
  auto begin = [&]() {
    wait_phases<DPW, R - 2>(min(R - 2, NP - 1 - p));
    __builtin_amdgcn_s_barrier();
  };
  auto refill = [&]() {
    if (p + R - 1 < NP) issue(p + R - 1, fill);
  };
  auto advance = [&]() {
    fill = cur;
    cur = cur + 1 == R ? 0 : cur + 1;
    ++p;
  };
  const unsigned lbase = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)lds;
  const unsigned lfrag = lbase + lane * 16;
  constexpr int PD = 3, NB = PD + 2;      // fragment reads ahead of the MFMAs (3 = 4 > 5, 6 measured: the waves do not wait for these reads)
  // a run of N fragments at compile-time offsets OFF(k) * 1024 from `sa`: reads PD ahead, counted waits (k_mlp3's idiom)
  auto run = [&](unsigned sa, auto offc, auto nc, auto&& pre, auto&& body) {
    constexpr int N = decltype(nc)::value;
    frag F[NB];
    sfor<0, (PD < N ? PD : N)>([&](auto fc) {
      constexpr int f = decltype(fc)::value;
      lds_rd<decltype(offc)::template at<f>() * 1024>(F[f % NB], sa);
    });
    pre();
    sfor<0, N>([&](auto fc) {
      constexpr int f = decltype(fc)::value;
      if constexpr (f + PD < N) lds_rd<decltype(offc)::template at<f + PD>() * 1024>(F[(f + PD) % NB], sa);
      lds_wait<(f + PD < N ? PD : N - 1 - f)>(F[f % NB]);
      body(fc, F[f % NB]);
    });
  };
  typedef Mlp3pOffProj<ND> OffProj;     // k-chunk half k / 12, own feature tile k % 12 (base + 12 r fragments)
  typedef Mlp3pOffW1 OffW1;             // fragment (c = k, u = role) (base + role fragments)
  typedef Mlp3pOffW2 OffW2;             // own feature tile k (base + ND + 12 r fragments)

  unsigned char* exch = lds + RING + BIAS;
  // a float per (token tile, lane) across the pair: own value out, barrier, partner's in
  auto pair_sum = [&](float (&v)[TM], int off) {
#pragma unroll
    for (int j = 0; j < TM; ++j) *reinterpret_cast<float*>(exch + wave * 1024 + off + (j * 64 + lane) * 4) = v[j];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const float o = *reinterpret_cast<const float*>(exch + (wave ^ 4) * 1024 + off + (j * 64 + lane) * 4);
      v[j] = role ? o + v[j] : v[j] + o;              // the SAME association in both waves of the pair: role 0's half first
    }
  };
  // `mine` holds six k-chunks of the B operand, the partner's six go to `theirs`, through the ring slot that is free right now (`fill`: the phase before the current one)
  auto share_half = [&](frag (&mine)[NCH / 2][TM], frag (&theirs)[NCH / 2][TM]) {
    unsigned char* area = lds + fill * SLOT;                        // 48 KiB: 8 waves x 3 chunks x 2 tiles x 1 KiB per round
    __builtin_amdgcn_s_barrier();                                   // every wave is done with that slot's fragments (and with the exchange area)
    sfor<0, 2>([&](auto rdc) {
      constexpr int rd = decltype(rdc)::value;
#pragma unroll
      for (int cc = 0; cc < 3; ++cc)
#pragma unroll
        for (int j = 0; j < TM; ++j) *reinterpret_cast<frag*>(area + ((wave * 3 + cc) * TM + j) * 1024 + lane * 16) = mine[rd * 3 + cc][j];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int cc = 0; cc < 3; ++cc)
#pragma unroll
        for (int j = 0; j < TM; ++j) theirs[rd * 3 + cc][j] = *reinterpret_cast<const frag*>(area + (((wave ^ 4) * 3 + cc) * TM + j) * 1024 + lane * 16);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                                 // (the second one also protects the slot against the next phase's refill)
    });
  };
  auto share_xb = [&]() { share_half(xo, xp); };                    // the role-relative form: xo = own chunks, xp = the partner's (LayerNorm output, stage-a result)
  // ---- x += Wp att: double phase qq = k-chunks 2 qq, 2 qq + 1; this wave: its 12 feature tiles of both ----
  // xo = k-chunks 0 .. 5 of the attention row, xp = 6 .. 11 in BOTH roles here (role 0 loaded xo, role 1 xp: the other half crosses through LDS; slot R - 1 is free, the
  // prologue filled slots 0 .. R - 2)
  if (role == 0) share_half(xo, xp);
  else share_half(xp, xo);
  sfor<0, NCH / 2>([&](auto qc) {
    constexpr int qq = decltype(qc)::value;
    begin();
    // the refill by role, as in the sub-steps: role 0's six pieces at the head of the phase, role 1's behind its first k-chunk (all 48 in one burst: 4550 cycles per
    // projection phase against the sub-steps' 2600, section stamps)
    const int rq = p + R - 1, rslot = fill;
    run(lfrag + cur * SLOT + role * (NHT * 1024), OffProj{}, std::integral_constant<int, 2 * NHT>{},
        [&]() {
          if (role == 0 && rq < NP) issue(rq, rslot);
          advance();
        },
        [&](auto fc, const frag& fr) {
          constexpr int f = decltype(fc)::value, c = 2 * qq + f / NHT, i = f % NHT;
#pragma unroll
          for (int j = 0; j < TM; ++j) {
            if constexpr (c < NCH / 2) acc[i][j] = Mma<T>::mma(fr, xo[c][j], acc[i][j]);
            else acc[i][j] = Mma<T>::mma(fr, xp[c - NCH / 2][j], acc[i][j]);
          }
          if constexpr (f == NHT - 1) {
            if (role == 1 && rq < NP) issue(rq, rslot);
          }
        });
  });
  mfma_results_settle();
  sec[1] = now();

  // ---- LayerNorm(x) over all 384 features -> xb; the stage-a output bias joins the residual ----
  {
    float sm[TM], qv[TM], mean[TM], rstd[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      float s0 = 0.f;
#pragma unroll
      for (int i = 0; i < NHT; ++i) s0 += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
      sm[j] = col_sum(s0);
    }
    pair_sum(sm, 0);
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      mean[j] = sm[j] * (1.f / D);
      asm volatile("" : "+v"(mean[j]));
      float q0 = 0.f;
#pragma unroll
      for (int i = 0; i < NHT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = acc[i][j][r] - mean[j];
          q0 += d * d;
        }
      qv[j] = col_sum(q0);
    }
    pair_sum(qv, 512);
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      rstd[j] = rsqrtf(qv[j] * (1.f / D) + a.eps);
#pragma unroll
      for (int c = 0; c < NCH / 2; ++c) {
        floatx4 t[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int i = c * 2 + u, fi = (role * NHT + i) * 16 + 4 * h;
          const floatx4 g = *reinterpret_cast<const floatx4*>(a.ln_g + fi);
          const floatx4 b = *reinterpret_cast<const floatx4*>(a.ln_b + fi);
          t[u] = (acc[i][j] - mean[j]) * rstd[j] * g + b;
        }
        xo[c][j] = pack_pair(t[0], t[1]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NHT; ++i) {
    const floatx4 bb = *reinterpret_cast<const floatx4*>(a.b2a + (role * NHT + i) * 16 + 4 * h);
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] += bb;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // LayerNorm parameters / bias loads: ordinary VMEM in front of the ring's counted waits (the ring is R - 1 phases ahead)
  share_xb();
  sec[2] = now();

  unsigned long long tk_b1 = 0, tk_fh = 0, tk_x = 0, tk_sh = 0;     // STAMP: cycle sums of [wait + first barrier | first half | repack + exchange (second barrier) | second half]
  // ---- one 32-unit hidden sub-step ----
  auto substep = [&](int t) {
    const unsigned long long s0 = now();
    begin();
    const unsigned long long s1 = now();
    floatx4 bv;
    lds_rd<0>(bv, lbase + RING + (t * 32 + role * 16 + 4 * h) * 4);
    floatx4 hacc[TM];
    half4v own[TM];
    // hidden tile `role`: h = W1[units 16 r .. 16 r + 15 of the slice] . xb + b1, the k-chunks in the order own half (xo), partner's half (xp): fragment (c, u = role) of the
    // LDS image is fragment 2 c + role, own chunk k is c = 6 role + k, the partner's c = 6 (1 - role) + k.  ONE read pipeline over both halves (12 fragments, PD ahead).
    {
      const unsigned sa1 = lfrag + cur * SLOT + (NCH * role + role) * 1024, sa2 = lfrag + cur * SLOT + (NCH * (1 - role) + role) * 1024;
      constexpr int N = NCH, HN = NCH / 2;
      frag F[NB];
      sfor<0, PD>([&](auto fc) {
        constexpr int f = decltype(fc)::value;
        lds_rd<2 * (f % HN) * 1024>(F[f % NB], f < HN ? sa1 : sa2);
      });
      if (role == 0) refill();        // role 0's six pieces here, role 1's at the head of the second half: each burst passes under the partner's MFMAs
      sfor<0, N>([&](auto fc) {
        constexpr int f = decltype(fc)::value, g = f + PD;
        if constexpr (g < N) lds_rd<2 * (g % HN) * 1024>(F[g % NB], g < HN ? sa1 : sa2);
        lds_wait<(g < N ? PD : N - 1 - f)>(F[f % NB]);
        if constexpr (f == 0) {
          asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(bv) : "n"(PD));   // (the bias read is older than every fragment read)
#pragma unroll
          for (int j = 0; j < TM; ++j) hacc[j] = bv;
        }
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          if constexpr (f < HN) hacc[j] = Mma<T>::mma(F[f % NB], xo[f][j], hacc[j]);
          else hacc[j] = Mma<T>::mma(F[f % NB], xp[f - HN][j], hacc[j]);
        }
      });
    }
    // the second half's first fragment reads go out NOW: they do not depend on the exchange, and their LDS round trip passes under the repack / barrier / partner read
    const unsigned sa3 = lfrag + cur * SLOT + (ND + role * NHT) * 1024;
    frag G[NB];
    sfor<0, PD>([&](auto fc) {
      constexpr int f = decltype(fc)::value;
      lds_rd<f * 1024>(G[f % NB], sa3);
    });
    mfma_results_settle();
    const unsigned long long s2 = now();
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const half4v v = {(half_t)hacc[j][0], (half_t)hacc[j][1], (half_t)hacc[j][2], (half_t)hacc[j][3]};
      const half4v z = {(half_t)0, (half_t)0, (half_t)0, (half_t)0};
      own[j] = __builtin_elementwise_max(v, z);
      *reinterpret_cast<half4v*>(exch + wave * 1024 + (j * 64 + lane) * 8) = own[j];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // both tiles of every pair are in the exchange area (overwritten again behind the next sub-step's first barrier)
    frag hb[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const half4v o = *reinterpret_cast<const half4v*>(exch + (wave ^ 4) * 1024 + (j * 64 + lane) * 8);
      const half4v lo = role ? o : own[j], hi = role ? own[j] : o;
      hb[j] = frag{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long s3 = now();
    // y[own 192 features] += W2[own rows, the slice] . relu(h)
    if (role == 1) refill();
    sfor<0, NHT>([&](auto fc) {
      constexpr int f = decltype(fc)::value, g = f + PD;
      if constexpr (g < NHT) lds_rd<g * 1024>(G[g % NB], sa3);
      lds_wait<(g < NHT ? PD : NHT - 1 - f)>(G[f % NB]);
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[f][j] = Mma<T>::mma(G[f % NB], hb[j], acc[f][j]);
    });
    advance();
    if constexpr (STAMP) {
      const unsigned long long s4 = now();
      tk_b1 += s1 - s0; tk_fh += s2 - s1; tk_x += s3 - s2; tk_sh += s4 - s3;
    }
  };
#pragma unroll 1
  for (int t = 0; t < nt; ++t) substep(t);
  mfma_results_settle();
  sec[3] = now();
  // ---- stage b (mlp_head): its input is the stage-a result ----
#pragma unroll
  for (int j = 0; j < TM; ++j)
#pragma unroll
    for (int c = 0; c < NCH / 2; ++c) xo[c][j] = pack_pair(acc[c * 2][j], acc[c * 2 + 1][j]);
  // the stage-b output bias joins the residual here (its loads pass under the exchange's barriers; in the epilogue they were a memory round trip in front of the stores)
  floatx4 bb2[NHT];
#pragma unroll
  for (int i = 0; i < NHT; ++i) bb2[i] = *reinterpret_cast<const floatx4*>(a.b2b + (role * NHT + i) * 16 + 4 * h);
  share_xb();                                              // (its first barrier: every wave is done with stage a's biases too)
#pragma unroll
  for (int i = 0; i < NHT; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] += bb2[i];
  stage_bias(a.b1b);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // one drain at the switch: the new biases (and whatever the ring had in flight)
  sec[4] = now();
#pragma unroll 1
  for (int t = 0; t < nt; ++t) substep(t);
  mfma_results_settle();
  sec[5] = now();

  // ---- epilogue: token-major store or fold into the NHWC map (own 12 feature tiles) ----
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const long long t = tok0 + j * 16 + r16;
    const bool live = t < a.M;
    const long long tc = live ? t : 0;
    const int f0 = 16 * (h & 1) + 8 * (h >> 1);
    if (!a.fmap) {
      T* yp = (T*)a.Y + tc * D + role * NHT * 16 + f0;
#pragma unroll
      for (int i = 0; i < NHT; i += 2) {
        const uint4 v = pair_tiles16(acc[i][j], acc[i + 1][j]);
        if (live) *reinterpret_cast<uint4*>(yp + i * 16) = v;
      }
    } else {
      const int tw = a.ws / a.p, S = tw * tw;
      const int nwx = a.mapW / a.ws, nwy = a.mapH / a.ws;
      const int tt = (int)(tc % S);
      const long long wi = tc / S;
      const int wx = (int)(wi % nwx), wy = (int)((wi / nwx) % nwy);
      const long long b = wi / ((long long)nwx * nwy);
      const int y0 = wy * a.ws + (tt / tw) * a.p, x0 = wx * a.ws + (tt % tw) * a.p;
#pragma unroll
      for (int i = 0; i < NHT; i += 2) {
        const int f = (role * NHT + i) * 16 + f0;
        const int ij = f / a.C, c = f - ij * a.C;
        const int pi = ij / a.p, pj = ij - pi * a.p;
        T* dst = (T*)a.fmap + ((b * a.mapH + y0 + pi) * a.mapW + x0 + pj) * a.cs + c;
        const uint4 v = pair_tiles16(acc[i][j], acc[i + 1][j]);
        if (live) *reinterpret_cast<uint4*>(dst) = v;
      }
    }
  }
  if constexpr (STAMP) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
    if (stamps && blockIdx.x == 0 && blockIdx.z == 0 && lane == 0) {
      unsigned long long* o = stamps + wave * 8;
      o[0] = t1 - tk0; o[1] = rt1 - rt0; o[2] = tk_b1; o[3] = tk_fh; o[4] = tk_x; o[5] = tk_sh; o[6] = (unsigned long long)(2 * nt);
      unsigned long long* q = stamps + 64 + wave * 8;
      q[0] = sec[0] - tk0; q[1] = sec[1] - sec[0]; q[2] = sec[2] - sec[1]; q[3] = sec[3] - sec[2]; q[4] = sec[4] - sec[3]; q[5] = sec[5] - sec[4]; q[6] = t1 - sec[5];
    }
  }
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate): a switch over the 64 encodable values
#define CFEN_VMCASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
CFEN_DEV void wait_vmcnt_rt(int n) {
  switch (n < 63 ? n : 63) {
    CFEN_VMCASE(0) CFEN_VMCASE(1) CFEN_VMCASE(2) CFEN_VMCASE(3) CFEN_VMCASE(4) CFEN_VMCASE(5) CFEN_VMCASE(6) CFEN_VMCASE(7)
    CFEN_VMCASE(8) CFEN_VMCASE(9) CFEN_VMCASE(10) CFEN_VMCASE(11) CFEN_VMCASE(12) CFEN_VMCASE(13) CFEN_VMCASE(14) CFEN_VMCASE(15)
    CFEN_VMCASE(16) CFEN_VMCASE(17) CFEN_VMCASE(18) CFEN_VMCASE(19) CFEN_VMCASE(20) CFEN_VMCASE(21) CFEN_VMCASE(22) CFEN_VMCASE(23)
    CFEN_VMCASE(24) CFEN_VMCASE(25) CFEN_VMCASE(26) CFEN_VMCASE(27) CFEN_VMCASE(28) CFEN_VMCASE(29) CFEN_VMCASE(30) CFEN_VMCASE(31)
    CFEN_VMCASE(32) CFEN_VMCASE(33) CFEN_VMCASE(34) CFEN_VMCASE(35) CFEN_VMCASE(36) CFEN_VMCASE(37) CFEN_VMCASE(38) CFEN_VMCASE(39)
    CFEN_VMCASE(40) CFEN_VMCASE(41) CFEN_VMCASE(42) CFEN_VMCASE(43) CFEN_VMCASE(44) CFEN_VMCASE(45) CFEN_VMCASE(46) CFEN_VMCASE(47)
    CFEN_VMCASE(48) CFEN_VMCASE(49) CFEN_VMCASE(50) CFEN_VMCASE(51) CFEN_VMCASE(52) CFEN_VMCASE(53) CFEN_VMCASE(54) CFEN_VMCASE(55)
    CFEN_VMCASE(56) CFEN_VMCASE(57) CFEN_VMCASE(58) CFEN_VMCASE(59) CFEN_VMCASE(60) CFEN_VMCASE(61) CFEN_VMCASE(62)
    default: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
  }
}
#undef CFEN_VMCASE

// head-major qkv element offset of feature f of [q | k | v] relative to its window's first element, without the token term
// (k_embed.hip hm_feature_off; head_dim 24 wherever the layout is used)
template <int D>
CFEN_DEV int st_hm_feature_off(int f, int S) {
  const int part = f / D, fd = f - part * D, hd = fd / 24, d = fd - hd * 24;
  return ((hd * 3 + part) * S) * 24 + d;
}

// LViT front half on the fragment-stream ring (D = 384): patch gather -> y = W_e x + b_e + x + pos -> X1; qkv = W_qkv LN1(y), written
// head-major for k_attention_hm (or row-major).  Both matrices are ROW-TILE streams (packing.pack_stream_rows): phase t = output rows
// t*32 .. +31, fragments (c, u) = k-chunk c, row tile u (u fastest).  The qkv phases store their tiles from inside the ring loop, so the ring's waits
// count every vector-memory operation the wave issues (`vm_issued` against the mark taken when a slot's DMA went out).
// DBG (timing experiments, results invalid): 1 = no LDS-DMA refills after the prologue, 8 = no qkv stores, 16 = no X1 stores.  `stamps` (timing build of the launcher, "front3.debug" & 64):
// s_memtime at the section boundaries of workgroup 0 -> [wave][prologue + gather, embedding phases, X1 stores + LayerNorm, qkv phases, 100 MHz ticks]
// PP (round 6): the patch edge as a compile-time constant (2: LViT level 3, C = 96; 4: GViT level 1 on the pooled map, C = 24; 0 = read it from the arguments).  The prologue's token ->
// pixel arithmetic ran on 64-bit run-time divisions (five per token tile, two per feature tile): 11-15 us of a 60 us launch on one wave per SIMD (tools/dbg_front3.py stamps).
template <int ND, int TM, int R, int DBG = 0, int PP = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_front3(Grouped<CfenEmbedQkvArgs> ga, unsigned long long* stamps) {
  typedef half_t T;
  typedef half8 frag;
  const CfenEmbedQkvArgs a = ga.g[blockIdx.z];
  constexpr int NW = 4, D = ND * 16, NCH = ND / 2, NF = 2 * ND, SLOT = NF * 1024, DPW = NF / NW, RING = R * SLOT;
  constexpr int NE = ND / 4, NQ = 3 * ND / 4, NP = NE + NQ;     // embedding / qkv double phases (four 16-row output tiles each, as k_mlp3)
  static_assert(NF % NW == 0 && ND % 4 == 0 && R >= 3 && RING <= 160 * 1024, "ring geometry");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[RING];
  const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, h = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long tok0 = ((long long)blockIdx.x * NW + wave) * (TM * 16);   // the launcher guarantees M % (NW * TM * 16) == 0

  auto now = [&]() -> unsigned long long {      // (only where no LDS read is in flight: the scalar load shares lgkmcnt with the hand-counted ds_reads)
    unsigned long long t = 0;
    if (stamps) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
  };
  const unsigned long long tk0 = now(), rt0 = stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;
  int vm_issued = 0;              // vector-memory operations this wave has issued since the last full drain
  int mark[R];                    // vm_issued right after the DMAs of the phase that sits in each slot
#pragma unroll
  for (int q = 0; q < R; ++q) mark[q] = 0;
  auto issue = [&](int q, int slot) {
    const unsigned char* src = q < NE ? (const unsigned char*)a.We + (size_t)q * SLOT : (const unsigned char*)a.Wqkv + (size_t)(q - NE) * SLOT;
#pragma unroll
    for (int k = 0; k < DPW; ++k) {
      const int f = k * NW + wave;
      st_dma(src + f * 1024 + lane * 16, lds + slot * SLOT + f * 1024);
    }
    vm_issued += DPW;
#pragma unroll
    for (int k = 0; k < R; ++k)
      if (k == slot) mark[k] = vm_issued;
  };
#pragma unroll
  for (int q = 0; q < R - 1; ++q) issue(q, q);

  // ---- gather x^T into accumulator layout (32-bit token arithmetic: the launcher checks M < 2^31) ----
  const int pe = PP ? PP : a.p, Cc = PP ? D / (PP * PP) : a.C;
  const unsigned tw = a.ws / pe, S = tw * tw, nwx = a.W / a.ws, nwy = a.H / a.ws;
  floatx4 acc[ND][TM];
  long long tk[TM];
  unsigned tts[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const unsigned t = (unsigned)tok0 + j * 16 + r16;
    tk[j] = t;
    const unsigned wi = t / S, tt = t - wi * S;
    tts[j] = tt;
    const unsigned wr = wi / nwx, wx = wi - wr * nwx, b = wr / nwy, wy = wr - b * nwy;
    const unsigned ty = tt / tw, tx = tt - ty * tw;
    const int y0 = wy * a.ws + ty * pe, x0 = wx * a.ws + tx * pe;
    const T* pix = (const T*)a.fmap + (((long long)b * a.H + y0) * a.W + x0) * a.cs;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int f = i * 16 + 4 * h;
      const int ij = f / Cc, c = f - ij * Cc;
      acc[i][j] = load4<T>(pix + ((ij / pe) * a.W + (ij % pe)) * a.cs + c);
    }
  }
  frag xb[NCH][TM];
#pragma unroll
  for (int j = 0; j < TM; ++j)
#pragma unroll
    for (int c = 0; c < NCH; ++c) xb[c][j] = pack_pair(acc[c * 2][j], acc[c * 2 + 1][j]);
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    const floatx4 bb = *reinterpret_cast<const floatx4*>(a.be + i * 16 + 4 * h);
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] += bb + load4<T>((const T*)a.pos + (size_t)tts[j] * D + i * 16 + 4 * h);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // one drain: ring prologue, pixels, position rows
  vm_issued = 0;
#pragma unroll
  for (int q = 0; q < R; ++q) mark[q] = 0;

  int p = 0, cur = 0, fill = R - 1;
  auto begin = [&]() {
    int m = 0;
#pragma unroll
    for (int k = 0; k < R; ++k)
      if (k == cur) m = mark[k];
    wait_vmcnt_rt(vm_issued - m);    // everything issued up to and including this phase's DMAs has completed
    __builtin_amdgcn_s_barrier();
  };
  auto refill = [&]() {
    if (!(DBG & 1) && p + R - 1 < NP) issue(p + R - 1, fill);
    fill = cur;
    cur = cur + 1 == R ? 0 : cur + 1;
    ++p;
  };
  const unsigned lbase = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)lds;
  const unsigned lfrag = lbase + lane * 16;
  constexpr int PD = 6, NB = 8;
  auto phase = [&](unsigned sa, auto&& pre, auto&& body) {
    frag F[NB];
    sfor<0, PD>([&](auto fc) {
      constexpr int f = decltype(fc)::value;
      lds_rd<f * 1024>(F[f % NB], sa);
    });
    pre();
    sfor<0, NF>([&](auto fc) {
      constexpr int f = decltype(fc)::value;
      if constexpr (f + PD < NF) lds_rd<(f + PD) * 1024>(F[(f + PD) % NB], sa);
      lds_wait<(f + PD < NF ? PD : NF - 1 - f)>(F[f % NB]);
      body(fc, F[f % NB]);
    });
  };

  const unsigned long long tk1 = now();
  // ---- y = W_e x + (b_e + x + pos): double phase q = output tiles 4q .. 4q + 3 (two row groups of two tiles) ----
  sfor<0, NE>([&](auto qc) {
    constexpr int q = decltype(qc)::value;
    begin();
    phase(lfrag + cur * SLOT, [&]() { refill(); }, [&](auto fc, const frag& fr) {
      constexpr int f = decltype(fc)::value, u = 2 * (f / ND) + f % 2, c = (f % ND) / 2;   // row group f / ND, inside it (c, tile) with the tile fastest
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[4 * q + u][j] = Mma<T>::mma(fr, xb[c][j], acc[4 * q + u][j]);
    });
  });
  mfma_results_settle();
  const unsigned long long tk2 = now();
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    T* yp = (T*)a.X1 + tk[j] * D + 16 * (h & 1) + 8 * (h >> 1);
#pragma unroll
    for (int i = 0; i < ND; i += 2) {
      const uint4 v = pair_tiles16(acc[i][j], acc[i + 1][j]);   // one 16-byte store per lane and tile pair
      if constexpr (DBG & 16) keep_live(v);      // (the value stays live: only the store goes)
      else *reinterpret_cast<uint4*>(yp + i * 16) = v;
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
      xb[c][j] = pack_pair(t[0], t[1]);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // X1 stores, LayerNorm parameters (the ring is R - 1 phases ahead: its DMAs have long landed)
  vm_issued = 0;
#pragma unroll
  for (int q = 0; q < R; ++q) mark[q] = 0;
  long long qrow[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j)
    qrow[j] = a.hm_heads ? (long long)((unsigned)tk[j] / S) * (3LL * S * D) + tts[j] * 24 : tk[j] * (3LL * D);

  const unsigned long long tk3 = now();
  // ---- qkv = W_qkv LN(y): double phase t = output tiles 4t .. 4t + 3, each stored as soon as its last k-chunk is in ----
#pragma unroll 1
  for (int t = 0; t < NQ; ++t) {
    begin();
    floatx4 qa[2][TM];      // the two 16-row tiles of a row group accumulate in rotation (four chains with the two token tiles)
    phase(lfrag + cur * SLOT, [&]() { refill(); }, [&](auto fc, const frag& fr) {
      constexpr int f = decltype(fc)::value, g2 = f / ND, v = f % 2, c = (f % ND) / 2;
      if constexpr (c == 0) {
#pragma unroll
        for (int j = 0; j < TM; ++j) qa[v][j] = floatx4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < TM; ++j) qa[v][j] = Mma<T>::mma(fr, xb[c][j], qa[v][j]);
      if constexpr (c == NCH - 1 && v == 1) {
        mfma_results_settle();
        {   // the two tiles of the row group leave as one 16-byte store per lane: 8 consecutive features, inside one head (heads are 24 = 3 x 8 wide)
          const int fq = (4 * t + 2 * g2) * 16 + 16 * (h & 1) + 8 * (h >> 1);
          const long long fo = a.hm_heads ? st_hm_feature_off<D>(fq, (int)S) : fq;
#pragma unroll
          for (int j = 0; j < TM; ++j) {
            const uint4 v = pair_tiles16(qa[0][j], qa[1][j]);
            if constexpr (DBG & 8) keep_live(v);
            else *reinterpret_cast<uint4*>((T*)a.QKV + qrow[j] + fo) = v;
          }
        }
        if (!(DBG & 8)) vm_issued += TM;
      }
    });
  }
  if (stamps) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long tk4 = now(), rt1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && blockIdx.z == 0 && lane == 0) {
      unsigned long long* o = stamps + wave * 8;
      o[0] = tk1 - tk0; o[1] = tk2 - tk1; o[2] = tk3 - tk2; o[3] = tk4 - tk3; o[4] = rt1 - rt0;
    }
  }
}

template <int ND, int TM, int R, int DBG = 0, int PP = 0>
int launch_front3(int ng, const CfenEmbedQkvArgs* ap, hipStream_t s) {
  Grouped<CfenEmbedQkvArgs> ga;
  for (int g = 0; g < CFEN_MAX_GROUPS; ++g) ga.g[g] = ap[g < ng ? g : 0];
  const long long per = 4LL * TM * 16;
  CFEN_CHECK_ARG(ap[0].M % per == 0, "front3: token count must be a multiple of %lld", per);
  const long long blocks = ap[0].M / per;
  CFEN_CHECK_ARG(blocks > 0 && ap[0].M < (1ll << 31), "front3: bad grid");
  unsigned long long* stamps = nullptr;
  const bool stamping = (cfen_tune_front3_debug() & 64) != 0;
  if (stamping) {   // timing run: the stamped workgroup's section times go to stderr after the launch (tools/dbg_front3.py)
    static unsigned long long* buf = nullptr;
    if (!buf && hipMalloc(&buf, 4 * 8 * sizeof(unsigned long long)) != hipSuccess) buf = nullptr;
    stamps = buf;
    if (stamps) (void)hipMemsetAsync(stamps, 0, 4 * 8 * sizeof(unsigned long long), s);
  }
  CFEN_LAUNCH((k_front3<ND, TM, R, DBG, PP>), dim3((unsigned)blocks, 1, ng), dim3(256), 0, s, ga, stamps);
  CFEN_CHECK_LAUNCH("front3");
  if (stamping && stamps && !cfen_recorder()) {
    unsigned long long hst[4 * 8];
    if (hipStreamSynchronize(s) == hipSuccess && hipMemcpy(hst, stamps, sizeof(hst), hipMemcpyDeviceToHost) == hipSuccess)
      for (int w = 0; w < 4; ++w) {
        const unsigned long long* o = hst + w * 8;
        fprintf(stderr, "front3 stamps wave %d: prologue + gather %llu, %d embedding phases %llu, X1 stores + LayerNorm %llu, %d qkv phases %llu cyc (%.0f / phase); %.1f us in all\n", w, o[0],
                ND / 4, o[1], o[2], 3 * ND / 4, o[3], (double)o[3] / (3 * ND / 4), (double)o[4] / 100.0);
      }
  }
  return CFEN_OK;
}

template <int ND, int TM, int R, int HB, int DBG = 0, int WPE = 1, int NW = 4, int PDX = 0, int SPR = 0, int STAMP = 0, int UMAJ = 0>
int launch_mlp3(int ng, const Mlp3Args* ap, hipStream_t s) {
  Grouped<Mlp3Args> ga;
  for (int g = 0; g < CFEN_MAX_GROUPS; ++g) ga.g[g] = ap[g < ng ? g : 0];
  CFEN_CHECK_ARG(ap[0].H <= HB, "mlp3: hidden width %d exceeds the %d this variant stages biases for", ap[0].H, HB);
  const long long per = (long long)NW * TM * 16, blocks = (ap[0].M + per - 1) / per;
  CFEN_CHECK_ARG(blocks > 0 && blocks < (1ll << 31), "mlp3: bad grid");
  unsigned long long* stamps = nullptr;
  if constexpr (STAMP) {   // timing build: the stamped workgroup's cycle sums go to stderr after the launch (tools/dbg_mlp3_stamps.py)
    static unsigned long long* buf = nullptr;
    if (!buf && hipMalloc(&buf, NW * 8 * sizeof(unsigned long long)) != hipSuccess) buf = nullptr;
    stamps = buf;
    if (stamps) (void)hipMemsetAsync(stamps, 0, NW * 8 * sizeof(unsigned long long), s);
  }
  CFEN_LAUNCH((k_mlp3<ND, TM, R, HB, DBG, WPE, NW, PDX, SPR, STAMP, UMAJ>), dim3((unsigned)blocks, 1, ng), dim3(NW * 64), 0, s, ga, stamps);
  CFEN_CHECK_LAUNCH("mlp3");
  if constexpr (STAMP) {
    if (stamps && !cfen_recorder()) {
      unsigned long long hst[NW * 8];
      if (hipStreamSynchronize(s) == hipSuccess && hipMemcpy(hst, stamps, sizeof(hst), hipMemcpyDeviceToHost) == hipSuccess)
        for (int w = 0; w < NW; ++w) {
          const unsigned long long* o = hst + w * 8;
          const double np = (double)o[4], ghz = o[7] ? (double)o[0] / ((double)o[7] * 10.0) : 0.0;
          fprintf(stderr, "mlp3 stamps wave %d: total %llu cyc (%.1f us, %.2f GHz), %d phases: wait %.0f barrier %.0f body %.0f cyc/phase; prologue %llu epilogue %llu cyc\n", w, o[0],
                  (double)o[7] / 100.0, ghz, (int)o[4], o[1] / np, o[2] / np, o[3] / np, o[5], o[6]);
        }
    }
  }
  return CFEN_OK;
}

template <int ND, int R, int HB, int STAMP = 0>
int launch_mlp3p(int ng, const Mlp3Args* ap, hipStream_t s) {
  Grouped<Mlp3Args> ga;
  for (int g = 0; g < CFEN_MAX_GROUPS; ++g) ga.g[g] = ap[g < ng ? g : 0];
  CFEN_CHECK_ARG(ap[0].H <= HB && ap[0].H % 256 == 0, "mlp3 (pair): hidden width %d (a multiple of 256, at most %d)", ap[0].H, HB);
  const long long per = 128, blocks = (ap[0].M + per - 1) / per;
  CFEN_CHECK_ARG(blocks > 0 && blocks < (1ll << 31), "mlp3 (pair): bad grid");
  unsigned long long* stamps = nullptr;
  if constexpr (STAMP) {
    static unsigned long long* buf = nullptr;
    if (!buf && hipMalloc(&buf, 128 * sizeof(unsigned long long)) != hipSuccess) buf = nullptr;
    stamps = buf;
    if (stamps) (void)hipMemsetAsync(stamps, 0, 128 * sizeof(unsigned long long), s);
  }
  CFEN_LAUNCH((k_mlp3p<ND, R, HB, STAMP>), dim3((unsigned)blocks, 1, ng), dim3(512), 0, s, ga, stamps);
  CFEN_CHECK_LAUNCH("mlp3 (pair)");
  if constexpr (STAMP) {
    if (stamps && !cfen_recorder()) {
      unsigned long long hst[128];
      if (hipStreamSynchronize(s) == hipSuccess && hipMemcpy(hst, stamps, sizeof(hst), hipMemcpyDeviceToHost) == hipSuccess)
        for (int w = 0; w < 8; ++w) {
          const unsigned long long *o = hst + w * 8, *q = hst + 64 + w * 8;
          const double n = o[6] ? (double)o[6] : 1.0;
          fprintf(stderr, "mlp3p stamps wave %d (role %d): %llu cyc in %.1f us; per sub-step: wait + barrier %.0f, first half %.0f, repack + exchange %.0f, second half %.0f cyc; sections: prologue loads %llu, "
                  "projection %llu, LayerNorm + operand exchange %llu, stage a %llu, switch %llu, stage b %llu, epilogue %llu cyc\n", w, w >> 2, o[0],
                  (double)o[1] / 100.0, o[2] / n, o[3] / n, o[4] / n, o[5] / n, q[0], q[1], q[2], q[3], q[4], q[5], q[6]);
        }
    }
  }
  return CFEN_OK;
}

}  // namespace

int& cfen_tune_mlp3_pair() {   // D = 384 blocks on k_mlp3p (two waves per SIMD, the hidden dimension split over a wave pair): 0 = k_mlp3<24, ...>, 1 on (default, round 6: 134-150 us
  static int v = 1;            // a launch against 152-172; +0.6-1.0 % on the headline step, level on the other two configurations), 2 on with stamps ("mlp3.pair")
  return v;
}
int& cfen_tune_mlp3_tm192() {   // the D = 192 variant: 22 (default, round 5) = 2 token tiles a wave at 256 registers on a three-slot ring, two 78 KB workgroups a CU; 24 = the same
  static int v = 22;            // on four slots (one workgroup a CU); 4 / 3 / 2 token tiles a wave on the six-slot ring of one 150 KB workgroup a CU (512 registers)
  return v;
}
int& cfen_tune_mlp3_debug() {
  static int v = 0;
  return v;
}
int& cfen_tune_front3_debug() {
  static int v = 0;
  return v;
}

bool cfen_mlp3_supported(int dtype, int D, int H) { return dtype == 1 && (D == 384 || D == 192) && H % 32 == 0 && H > 0 && H <= (D == 384 ? 1536 : 768); }

int cfen_mlp3_impl_g(int dtype, int ng, const Mlp3Args* ap, hipStream_t s) {
  CFEN_CHECK_ARG(ng >= 1 && ng <= CFEN_MAX_GROUPS && ap, "mlp3: 1..%d problems per launch", CFEN_MAX_GROUPS);
  CFEN_CHECK_ARG(cfen_mlp3_supported(dtype, ap[0].D, ap[0].H), "mlp3: the fragment-stream MLP runs fp16 at D = 384 (H <= 1536) / D = 192 (H <= 768), H %% 32 == 0; got dtype %d D %d H %d",
                 dtype, ap[0].D, ap[0].H);
  for (int g = 0; g < ng; ++g) {
    const Mlp3Args& a = ap[g];
    CFEN_CHECK_ARG(a.M > 0 && a.X && a.Wa && a.b1a && a.b2a && (a.Y || a.fmap), "mlp3: null pointer");
    CFEN_CHECK_ARG((a.Wb == nullptr) == (a.b1b == nullptr) && (a.Wb == nullptr) == (a.b2b == nullptr), "mlp3: incomplete second stage");
    CFEN_CHECK_ARG((a.ln_g == nullptr) == (a.ln_b == nullptr), "mlp3: LayerNorm needs gamma and beta");
    CFEN_CHECK_ARG((a.A == nullptr) == (a.Wp == nullptr), "mlp3: projection prologue needs A and Wp");
    CFEN_CHECK_ARG(cfen_aligned16(a.X) && cfen_aligned16(a.A) && cfen_aligned16(a.Wp) && cfen_aligned16(a.Y) && cfen_aligned16(a.fmap) && cfen_aligned16(a.Wa) &&
                   cfen_aligned16(a.Wb) && cfen_aligned16(a.b1a) && cfen_aligned16(a.b2a) && cfen_aligned16(a.b1b) && cfen_aligned16(a.b2b) &&
                   cfen_aligned16(a.ln_g) && cfen_aligned16(a.ln_b), "mlp3: pointers must be 16-byte aligned");
    if (a.fmap) {
      CFEN_CHECK_ARG(a.C > 0 && a.C % 8 == 0 && a.cs >= a.C && a.cs % 8 == 0 && a.p > 0 && a.ws % a.p == 0 && a.mapH % a.ws == 0 && a.mapW % a.ws == 0 &&
                     a.p * a.p * a.C == a.D, "mlp3: bad fold geometry");
      CFEN_CHECK_ARG(a.M % ((long long)(a.ws / a.p) * (a.ws / a.p)) == 0, "mlp3: token count does not tile the map");
    }
    CFEN_CHECK_ARG(a.M == ap[0].M && a.D == ap[0].D && a.H == ap[0].H && (a.Wb == nullptr) == (ap[0].Wb == nullptr) &&
                   (a.ln_g == nullptr) == (ap[0].ln_g == nullptr) && (a.Wp == nullptr) == (ap[0].Wp == nullptr) && (a.fmap == nullptr) == (ap[0].fmap == nullptr),
                   "mlp3: grouped problems must have the same shape");
  }
  if (ap[0].D == 384 && cfen_tune_mlp3_pair() && ap[0].Wp && ap[0].ln_g && ap[0].Wb && ap[0].H % 256 == 0)
    return cfen_tune_mlp3_pair() == 2 ? launch_mlp3p<24, 3, 1536, 1>(ng, ap, s) : launch_mlp3p<24, 3, 1536>(ng, ap, s);
  if (ap[0].D == 384 && cfen_tune_mlp3_debug() == 1) return launch_mlp3<24, 2, 3, 1536, 1>(ng, ap, s);
  if (ap[0].D == 384 && cfen_tune_mlp3_debug() == 2) return launch_mlp3<24, 2, 3, 1536, 2>(ng, ap, s);
  if (ap[0].D == 384 && cfen_tune_mlp3_debug() == 8) return launch_mlp3<24, 1, 3, 1536, 0, 2, 8>(ng, ap, s);   // round-5 A/B: 8 waves x 1 tile, two waves per SIMD
  // (round-6 A/B variants 14-17, 20, 65, 72 -- other piece positions, 6 reads ahead, more stamped builds -- are recorded in profiles/r06_mlp3_d384_variants.txt / _stamps.txt and
  //  no longer compiled: each instantiation of this kernel is 15-25 s of build time)
  if (ap[0].D == 384 && cfen_tune_mlp3_debug() == 11) return launch_mlp3<24, 2, 3, 1536, 0, 1, 4, 0, 1>(ng, ap, s);       // round-6 A/Bs: DMA pieces spread between the MFMAs
  if (ap[0].D == 384 && cfen_tune_mlp3_debug() == 18) return launch_mlp3<24, 2, 3, 1536, 0, 1, 4, 4, 5>(ng, ap, s);       // ... with 4 fragment reads ahead
  if (ap[0].D == 384 && cfen_tune_mlp3_debug() == 19) return launch_mlp3<24, 2, 3, 1536, 0, 1, 4, 4, 5, 0, 1>(ng, ap, s);    // ... and the first hidden tile repacked early (UMAJ)
  if (ap[0].D == 384 && cfen_tune_mlp3_debug() == 100) return launch_mlp3<24, 2, 3, 1536>(ng, ap, s);                     // the burst issue of rounds 3-5
  if (ap[0].D == 384 && cfen_tune_mlp3_debug() == 64) return launch_mlp3<24, 2, 3, 1536, 0, 1, 4, 0, 0, 1>(ng, ap, s);    // stamped timing builds
  if (ap[0].D == 384 && cfen_tune_mlp3_debug() == 67) return launch_mlp3<24, 2, 3, 1536, 1, 1, 4, 0, 0, 1>(ng, ap, s);    // stamped, no refills
  if (ap[0].D == 384 && cfen_tune_mlp3_debug() == 68) return launch_mlp3<24, 2, 3, 1536, 4, 1, 4, 0, 0, 1>(ng, ap, s);    // stamped, no LDS fragment reads (MFMAs + refills + barrier)
  if (ap[0].D == 384 && cfen_tune_mlp3_debug() == 69) return launch_mlp3<24, 2, 3, 1536, 5, 1, 4, 0, 0, 1>(ng, ap, s);    // stamped, neither (MFMAs + barrier + the mid-phase repack)
  if (ap[0].D == 384 && cfen_tune_mlp3_debug() == 70) return launch_mlp3<24, 2, 3, 1536, 32, 1, 4, 0, 0, 1>(ng, ap, s);   // stamped, 4-byte refill pieces
  if (ap[0].D == 384 && cfen_tune_mlp3_debug() == 71) return launch_mlp3<24, 2, 3, 1536, 6, 1, 4, 0, 0, 1>(ng, ap, s);    // stamped, refills + barrier only (no MFMAs, no reads)
  // round 6 default: refills in thirds (phase start / mid-phase gap / spread over the second half), 4 fragment reads ahead, first hidden tile repacked early:
  // 167 -> 156 us on 192 workgroups, 158 -> 152 on 16 (same box, tools/dbg_mlp3_stamps.py; profiles/r06_mlp3_d384_variants.txt), bit for bit the round-5 results
  if (ap[0].D == 384) return launch_mlp3<24, 2, 3, 1536, 0, 1, 4, 4, 5, 0, 1>(ng, ap, s);
  if (cfen_tune_mlp3_tm192() == 22) return launch_mlp3<12, 2, 3, 768, 0, 2>(ng, ap, s);   // two 78 KB workgroups a CU, three-slot ring (round 5 A/B)
  if (cfen_tune_mlp3_tm192() == 25) return launch_mlp3<12, 2, 3, 768, 0, 2, 4, 4, 5, 0, 1>(ng, ap, s);   // round-6 A/Bs: the D = 384 kernel's refill placement (thirds) + early repack + 4 reads ahead
  if (cfen_tune_mlp3_tm192() == 28) return launch_mlp3<12, 2, 3, 768, 0, 2, 8>(ng, ap, s);        // ONE 8-wave workgroup a CU (two waves per SIMD) sharing one three-slot ring: 256 tokens per weight byte streamed
  if (cfen_tune_mlp3_tm192() == 24) return launch_mlp3<12, 2, 4, 768, 0, 2>(ng, ap, s);   // one 102 KB workgroup a CU at 256 registers, four slots
  if (cfen_tune_mlp3_tm192() == 3) return launch_mlp3<12, 3, 6, 768>(ng, ap, s);
  if (cfen_tune_mlp3_tm192() == 2) return launch_mlp3<12, 2, 6, 768>(ng, ap, s);
  return launch_mlp3<12, 4, 6, 768>(ng, ap, s);
}

bool cfen_front3_supported(int dtype, int D, long long M) { return dtype == 1 && D == 384 && M % 128 == 0; }

// We / Wqkv of the arguments are ROW-TILE fragment streams here (packing.pack_stream_rows), everything else as cfen_embed_qkv_impl_g
int cfen_front3_impl_g(int dtype, int ng, const CfenEmbedQkvArgs* ap, hipStream_t s) {
  CFEN_CHECK_ARG(ng >= 1 && ng <= CFEN_MAX_GROUPS && ap, "front3: 1..%d problems per launch", CFEN_MAX_GROUPS);
  CFEN_CHECK_ARG(cfen_front3_supported(dtype, ap[0].D, ap[0].M), "front3: the fragment-stream front half runs fp16 at D = 384 on whole 128-token workgroups (dtype %d, D %d, M %lld)",
                 dtype, ap[0].D, ap[0].M);
  for (int g = 0; g < ng; ++g) {
    const CfenEmbedQkvArgs& a = ap[g];
    CFEN_CHECK_ARG(a.fmap && a.We && a.be && a.pos && a.ln_g && a.ln_b && a.Wqkv && a.X1 && a.QKV, "front3: null pointer");
    CFEN_CHECK_ARG(cfen_aligned16(a.fmap) && cfen_aligned16(a.We) && cfen_aligned16(a.be) && cfen_aligned16(a.pos) && cfen_aligned16(a.ln_g) &&
                   cfen_aligned16(a.ln_b) && cfen_aligned16(a.Wqkv) && cfen_aligned16(a.X1) && cfen_aligned16(a.QKV), "front3: pointers must be 16-byte aligned");
    CFEN_CHECK_ARG(a.C > 0 && a.C % 8 == 0 && a.cs % 8 == 0 && a.cs >= a.C && a.p > 0 && a.ws % a.p == 0 && a.H % a.ws == 0 && a.W % a.ws == 0 && a.B > 0,
                   "front3: bad token geometry");
    const int tw = a.ws / a.p;
    CFEN_CHECK_ARG(a.D == a.p * a.p * a.C && a.M == (long long)a.B * (a.H / a.ws) * (a.W / a.ws) * tw * tw, "front3: D / M do not match the map");
    CFEN_CHECK_ARG(a.D == ap[0].D && a.M == ap[0].M && a.hm_heads == ap[0].hm_heads && a.p == ap[0].p && a.C == ap[0].C, "front3: grouped problems must have the same shape");
    CFEN_CHECK_ARG(a.hm_heads == 0 || a.D == a.hm_heads * 24, "front3: the head-major layout needs head_dim 24");
  }
  if ((cfen_tune_front3_debug() & 63) == 1) return launch_front3<24, 2, 3, 1>(ng, ap, s);     // timing experiments (results invalid): no refills / no qkv stores / neither store
  if ((cfen_tune_front3_debug() & 63) == 8) return launch_front3<24, 2, 3, 8>(ng, ap, s);
  if ((cfen_tune_front3_debug() & 63) == 24) return launch_front3<24, 2, 3, 24>(ng, ap, s);
  if ((cfen_tune_front3_debug() & 63) == 2) return launch_front3<24, 2, 3>(ng, ap, s);        // A/B: the run-time geometry (rounds 3-5)
  if (ap[0].p == 2 && ap[0].C == 96) return launch_front3<24, 2, 3, 0, 2>(ng, ap, s);
  if (ap[0].p == 4 && ap[0].C == 24) return launch_front3<24, 2, 3, 0, 4>(ng, ap, s);
  return launch_front3<24, 2, 3>(ng, ap, s);
}
