// Persistent GEMM chain for the few-token transformer blocks (GViT: 128 .. 2048 tokens against 6 .. 100 MB of weights per block).
//
// Replaces, for one GViT instance (reference models/networks_iid_hlgvit_crs_gd4_cfs_v3.py:1272-1325 with the encoder layer v3:1359-1390),
// the dependent chain of nn.Linear launches  linear_encoding (v3:1143,1166) -> in_proj (v3:1364)  and  out_proj (v3:1386) -> linear1 ->
// linear2 (v3:1388-1389) -> mlp_head.0 -> mlp_head.3 + fold (v3:1173,1186)  by ONE launch per run of GEMMs.
//
// Why one launch (measured): inside the forward the GViT lane runs beside LViT kernels that take whole CUs; what the lane
// costs is ~0.64 x its CU-time + 0.18 x its duration + a constant per fork.  Ten small launches per block each had to win their CUs back from
// the neighbour lane and ramp up on 256-512 workgroups; a TEAM of gridDim.x workgroups (one per CU, 48 by default) that keeps its CUs for the
// whole chain and meets at a grid barrier between GEMMs has a quarter of the CU-time.
//
// One GEMM phase  Y = epi(X W^T):  work unit = 128 tokens x 128 output features x one K slice.  The 8 waves of the workgroup own one 16-row
// feature tile each (weights = MFMA A operand, cfen_common.hpp); the unit's 128 token rows are the shared B operand.
//   * token rows: LDS-DMA into a ring of NS stages of [128 rows][128 B] (64 k), XOR-swizzled on the source address (as k_gemm_dma);
//   * weights: a FRAGMENT STREAM packed once on the host (packing.pack_stream_tiles: [N/16][K/32][64 lanes][16 B]); every wave DMAs the two
//     1 KiB fragments of its own tile for each K-step into its private 2 KiB slice of the stage -- each weight byte passes the DMA path ONCE
//     per 128 tokens and is read back by ONE conflict-free lane-linear ds_read_b128;
//   * counted vmcnt, one raw s_barrier per K-step, NS - 1 K-steps (96 KB per CU) in flight.
// Epilogue: LayerNorm fold (row statistics accumulated from the staged token tiles, as k_gemm_dma), bias, ReLU, residual, position table,
// optional fold into the NHWC map; outputs leave with write-through (sc1) stores so that the grid barrier needs no L2 write-back.
// Split-K (K-heavy phases): slabs + arrival ticket + reduction by the last arriver, the protocol of k_gemm_dma with every reducing wave
// acquiring.  Grid barrier between phases: every wave drains its stores, one lane adds to the team counter, polls it relaxed, acquires.
#include "cfen_common.hpp"
#include "cfen_internal.hpp"

int& cfen_tune_gvit_debug() {   // timing experiments (results invalid): 1 no weight DMAs, 2 no token DMAs, 4 no fragment reads / MFMAs ("gvit.debug")
  static int v = 0;
  return v;
}
int& cfen_tune_gvit_max_concurrent() {   // a grid barrier needs its whole team resident; with several forwards of the chain plan in flight (replica plans on their
                                         // own streams) two launches asking for more CUs than the chip has can each end up partially resident and spin on each other
                                         // until GV_SPIN_LIMIT.  The host caps team x ng x this at 256 CUs (cfen_net.cpp) and hipnet refuses a replica beyond it.
  static int v = 1;
  return v;
}
int& cfen_tune_gvit_team() {   // workgroups (= CUs) per GViT block of the persistent chain ("gvit.team")
  static int v = 48;
  return v;
}

namespace {

constexpr int GV_NS = 4;                      // ring stages
constexpr int GV_XST = 128 * 128;             // token part of a stage: 128 rows x 128 bytes
constexpr int GV_AST = 8 * 2048;              // weight part: 8 waves x 2 fragments of 1 KiB
constexpr int GV_STAGE = GV_XST + GV_AST;     // 32 KiB
constexpr int GV_LOADS = 4;                   // LDS-DMA instructions per wave and K-step (2 token pieces + 2 fragments)
constexpr int GV_MISC = 2048;                 // row statistics (128 x 2 floats) + flags
constexpr unsigned GV_SPIN_LIMIT = 1u << 20;  // polls before a waiting workgroup gives up (about a second) and raises the error word

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

CFEN_DEV void gv_dma16(const void* g, unsigned char* l) {
  __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int Y> CFEN_DEV void gv_wait_steps(int younger) {
  if constexpr (Y == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    if (younger >= Y) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Y * GV_LOADS) : "memory");
    else gv_wait_steps<Y - 1>(younger);
  }
}

// the team's grid barrier: `target` arrivals on *bar (monotone over the launch; zero when the launch starts)
CFEN_DEV void gv_grid_barrier(unsigned* bar, unsigned target, unsigned* err) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's (write-through) stores have left
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > GV_SPIN_LIMIT) {   // cannot happen while the team fits the chip; never hang the GPU if it does
        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // every wave: nothing older than the barrier is served from this CU's L1
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

struct GvFold { int H, W, cs, C, p; };   // Y is an NHWC map the tokens tile with p x p patches (GViT: one window = the whole map)

// where features n .. n+3 of token m go
CFEN_DEV size_t gv_out_off(const GvFold& f, int fold, int ldy, int m, int n) {
  if (!fold) return (size_t)m * ldy + n;
  const int tw = f.W / f.p, S = tw * (f.H / f.p);
  const int t = m % S, b = m / S;
  const int ij = n / f.C, c = n - ij * f.C;
  const int y = (t / tw) * f.p + ij / f.p, x = (t % tw) * f.p + ij % f.p;
  return (((size_t)b * f.H + y) * f.W + x) * f.cs + c;
}

struct GvPhase {
  const half_t* X; const half_t* W; const float* bias; const float* lnf_s; const half_t* R; const half_t* P; half_t* Y;
  int ldx, ldr, ldy, period, N, K, relu, nsplit, fold;
};
constexpr int GV_MAX_PHASES = 5;
struct GvArgs {
  GvPhase ph[GV_MAX_PHASES];
  int nph, M;
  GvFold fold;
  unsigned* bar;    // grid-barrier counter of this launch (zero before it)
  unsigned* cnt;    // split-K arrival counters, one per (token block, feature block) tile, zero between launches
  float* part;      // split-K slabs [tile][slice][8][512] float4
  unsigned* err;    // raised when a wait gave up
  float eps;
  int dbg;
  int release;
};

__global__ __launch_bounds__(512) void k_gvit_chain(Grouped<GvArgs> ga) {
  const GvArgs& a = ga.g[blockIdx.y];
  __shared__ __attribute__((aligned(1024))) unsigned char lds[GV_NS * GV_STAGE + GV_MISC];
  float* stats = reinterpret_cast<float*>(lds + GV_NS * GV_STAGE);
  unsigned* flag = reinterpret_cast<unsigned*>(lds + GV_NS * GV_STAGE + 1024);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, h = lane >> 4;
  const int T = gridDim.x, M = a.M, dbg = a.dbg;
  const int tbs = (M + 127) >> 7;

  for (int p = 0; p < a.nph; ++p) {
    const GvPhase ph = a.ph[p];
    const int rbs = ph.N >> 7;
    const int nunits = tbs * rbs * ph.nsplit;
    const int nk = (ph.K >> 6) / ph.nsplit;       // K-steps of 64 per slice
    const int kcs = ph.K >> 5;                    // fragments per feature tile
    for (int u = blockIdx.x; u < nunits; u += T) {
      const int tb = u % tbs, rest = u / tbs;
      const int ks = rest % ph.nsplit, rb = rest / ph.nsplit;
      const int m0 = tb << 7, n0 = (rb << 7) + (wave << 4);
      // ---- DMA assignment: token pieces (this wave fills rows wave*16 .. +15 as two 8-row instructions), own weight fragments
      const half_t* xsrc[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = wave * 16 + i * 8 + (lane >> 3), slot = lane & 7;
        xsrc[i] = ph.X + (size_t)min(m0 + row, M - 1) * ph.ldx + (size_t)ks * nk * 64 + ((slot ^ (row & 7)) << 3);
      }
      const half_t* wsrc = ph.W + ((size_t)(n0 >> 4) * kcs + (size_t)ks * nk * 2) * 512 + lane * 8;
#define GV_ISSUE(kt_, slot_)                                                                     \
  do {                                                                                           \
    unsigned char* st_ = lds + (slot_) * GV_STAGE;                                               \
    if (!(dbg & 2)) {                                                                            \
      gv_dma16(xsrc[0] + (kt_) * 64, st_ + (wave * 16) * 128);                                   \
      gv_dma16(xsrc[1] + (kt_) * 64, st_ + (wave * 16 + 8) * 128);                               \
    }                                                                                            \
    if (!(dbg & 1)) {                                                                            \
      gv_dma16(wsrc + (size_t)(kt_) * 1024, st_ + GV_XST + wave * 2048);                         \
      gv_dma16(wsrc + (size_t)(kt_) * 1024 + 512, st_ + GV_XST + wave * 2048 + 1024);            \
    }                                                                                            \
  } while (0)

      __builtin_amdgcn_s_barrier();   // every wave is done with the ring (previous unit's last K-step, previous epilogue's scratch)
#pragma unroll
      for (int st = 0; st < GV_NS - 1; ++st)
        if (st < nk) GV_ISSUE(st, st);

      floatx4 acc[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
      float ls[2] = {0.f, 0.f}, lq[2] = {0.f, 0.f};
      const int lnoff = (tid >> 3) * 128 + (tid & 7) * 16;   // + 64 rows for the second piece
      const int sw = r16 & 7;
      int buf = 0, fill = GV_NS - 1;
      for (int kt = 0; kt < nk; ++kt) {
        gv_wait_steps<GV_NS - 2>(min(GV_NS - 2, nk - 1 - kt));
        __builtin_amdgcn_s_barrier();
        if (kt + GV_NS - 1 < nk) GV_ISSUE(kt + GV_NS - 1, fill);
        const unsigned char* st = lds + buf * GV_STAGE;
        if (ph.lnf_s) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const half8 v = *reinterpret_cast<const half8*>(st + lnoff + j * 64 * 128);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; ls[j] += f; lq[j] += f * f; }
          }
        }
        if (!(dbg & 4))
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const half8 af = *reinterpret_cast<const half8*>(st + GV_XST + wave * 2048 + c * 1024 + lane * 16);
          const int po = ((c * 4 + h) ^ sw) << 4;
          half8 bf[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) bf[i] = *reinterpret_cast<const half8*>(st + (i * 16 + r16) * 128 + po);
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[i], acc[i], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        fill = buf;
        buf = buf + 1 == GV_NS ? 0 : buf + 1;
      }
#undef GV_ISSUE

      const int tile = rb * tbs + tb;
      if (ph.nsplit > 1 && !(dbg & 32)) {
        // slabs out write-through, drained, ticket; the last arriver adds the slabs in slice order (its own from memory): deterministic
        const size_t slab = (size_t)8 * 512 * 4;   // floats per (tile, slice)
        const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(a.part, 0, -1, 0x00020000);
        const unsigned pbase = (unsigned)(((size_t)tile * ph.nsplit + ks) * slab * 4);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const floatx4 v = acc[i];
          const u32x4 bits = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
          if (a.release == 2) __builtin_amdgcn_raw_buffer_store_b128(bits, prs, (int)(pbase + (i * 512 + tid) * 16), 0, 0);    // plain stores: the canonical recipe
          else __builtin_amdgcn_raw_buffer_store_b128(bits, prs, (int)(pbase + (i * 512 + tid) * 16), 0, 16);   // aux 16 = sc1
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (a.release) {   // "gemm.splitk_release" (default 1, round 4): an agent-scope release on top of the write-through stores -- see the note at the host entry
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (tid == 0) *flag = __hip_atomic_fetch_add(a.cnt + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (*flag != (unsigned)(ph.nsplit - 1)) continue;   // not the last slice of this tile (workgroup-uniform)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // every reducing wave
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) __hip_atomic_store(a.cnt + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float* p0 = a.part + (size_t)tile * ph.nsplit * slab;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = *reinterpret_cast<const floatx4*>(p0 + (i * 512 + tid) * 4);
        for (int sl = 1; sl < ph.nsplit; ++sl) {
          const float* ps = p0 + (size_t)sl * slab;
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i] += *reinterpret_cast<const floatx4*>(ps + (i * 512 + tid) * 4);
        }
      }
      const int n = n0 + 4 * h;
      if (ph.lnf_s) {   // row statistics of the 128 tokens -> LDS; acc <- rstd (acc - mean s)
        __builtin_amdgcn_s_barrier();   // every wave has left the K loop: the flags / stats area is free
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float sv = ls[j], qv = lq[j];
          sv += dpp_mov<0xB1>(sv); qv += dpp_mov<0xB1>(qv);
          sv += dpp_mov<0x4E>(sv); qv += dpp_mov<0x4E>(qv);
          sv += dpp_mov<0x141>(sv); qv += dpp_mov<0x141>(qv);
          if ((tid & 7) == 0) {
            const float mean = sv / (float)ph.K;
            const float var = fmaxf(qv / (float)ph.K - mean * mean, 0.f);
            stats[2 * ((tid >> 3) + 64 * j)] = mean;
            stats[2 * ((tid >> 3) + 64 * j) + 1] = rsqrtf(var + a.eps);
          }
        }
        __syncthreads();
        const floatx4 sn = *reinterpret_cast<const floatx4*>(ph.lnf_s + n);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float mean = stats[2 * (i * 16 + r16)], rstd = stats[2 * (i * 16 + r16) + 1];
          acc[i] = (acc[i] - sn * mean) * rstd;
        }
      }
      // ---- epilogue: all loads, then the arithmetic, then the (write-through) stores
      floatx4 bias = {0.f, 0.f, 0.f, 0.f};
      if (ph.bias) bias = *reinterpret_cast<const floatx4*>(ph.bias + n);
      half4 rv[8], pv[8];
      if (ph.R) {
#pragma unroll
        for (int i = 0; i < 8; ++i) rv[i] = *reinterpret_cast<const half4*>(ph.R + (size_t)min(m0 + i * 16 + r16, M - 1) * ph.ldr + n);
      }
      if (ph.P) {
#pragma unroll
        for (int i = 0; i < 8; ++i) pv[i] = *reinterpret_cast<const half4*>(ph.P + (size_t)(min(m0 + i * 16 + r16, M - 1) % ph.period) * ph.N + n);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        floatx4 v = acc[i] + bias;
        if (ph.relu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        if (ph.R) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += (float)rv[i][r];
        }
        if (ph.P) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += (float)pv[i][r];
        }
        acc[i] = v;
      }
      const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(ph.Y, 0, -1, 0x00020000);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int m = m0 + i * 16 + r16;
        if (m >= M || (dbg & 8)) continue;
        const half4 o = {(half_t)acc[i][0], (half_t)acc[i][1], (half_t)acc[i][2], (half_t)acc[i][3]};
        const u32x2 bits = __builtin_bit_cast(u32x2, o);
        __builtin_amdgcn_raw_buffer_store_b64(bits, yrs, (int)(gv_out_off(a.fold, ph.fold, ph.ldy, m, n) * 2), 0, 16);   // sc1: write-through
      }
    }
    if (p + 1 < a.nph && !(dbg & 16)) gv_grid_barrier(a.bar, (unsigned)T * (unsigned)(p + 1), a.err);
  }
}

}  // namespace

// host side ------------------------------------------------------------------------------------------------------------------------------
size_t cfen_gvit_chain_part_bytes(int M, int maxN, int max_nsplit) {
  return (size_t)((M + 127) / 128) * (maxN / 128) * max_nsplit * 8 * 512 * 16;
}

int cfen_gvit_chain_impl_g(int dtype, int ng, const CfenChainArgs* ca, int team, hipStream_t s) {
  CFEN_CHECK_ARG(dtype == 1, "gvit_chain: fp16 only");
  CFEN_CHECK_ARG(ng >= 1 && ng <= CFEN_MAX_GROUPS && ca, "gvit_chain: 1..%d problems per launch", CFEN_MAX_GROUPS);
  CFEN_CHECK_ARG(team >= 1 && team * ng <= 256, "gvit_chain: %d x %d workgroups do not fit the chip (the grid barrier needs them all resident)", team, ng);
  Grouped<GvArgs> ga;
  memset(&ga, 0, sizeof(ga));
  for (int g = 0; g < ng; ++g) {
    const CfenChainArgs& c = ca[g];
    GvArgs& a = ga.g[g];
    CFEN_CHECK_ARG(c.nph >= 1 && c.nph <= GV_MAX_PHASES && c.M >= 1 && c.M == ca[0].M && c.nph == ca[0].nph, "gvit_chain: bad phase count / token count");
    CFEN_CHECK_ARG(c.bar && c.cnt && c.err && cfen_aligned16(c.part), "gvit_chain: synchronisation words missing");
    a.nph = c.nph; a.M = c.M; a.bar = c.bar; a.cnt = c.cnt; a.part = c.part; a.err = c.err; a.eps = cfen_gemm_lnf_eps(); a.dbg = cfen_tune_gvit_debug(); a.release = cfen_tune_gemm_splitk_release();
    a.fold = GvFold{c.fH, c.fW, c.fcs, c.fC, c.fp};
    const int tbs = (c.M + 127) / 128;
    for (int p = 0; p < c.nph; ++p) {
      const CfenChainPhase& q = c.ph[p];
      CFEN_CHECK_ARG(q.X && q.W && q.Y && cfen_aligned16(q.X) && cfen_aligned16(q.W) && cfen_aligned16(q.Y) && cfen_aligned16(q.R) && cfen_aligned16(q.P) &&
                     cfen_aligned16(q.bias) && cfen_aligned16(q.lnf_s), "gvit_chain: phase %d: null or misaligned operand", p);
      CFEN_CHECK_ARG(q.N > 0 && q.N % 128 == 0 && q.K > 0 && q.K % 64 == 0, "gvit_chain: phase %d: N (%d) must be a multiple of 128, K (%d) of 64", p, q.N, q.K);
      CFEN_CHECK_ARG(q.nsplit >= 1 && (q.K / 64) % q.nsplit == 0, "gvit_chain: phase %d: %d K-steps do not split %d ways", p, q.K / 64, q.nsplit);
      CFEN_CHECK_ARG(q.ldx % 8 == 0 && q.ldx >= q.K && (q.fold || (q.ldy % 4 == 0 && q.ldy >= q.N)) && (!q.R || (q.ldr % 4 == 0 && q.ldr >= q.N)),
                     "gvit_chain: phase %d: bad leading dimension", p);
      CFEN_CHECK_ARG(!q.P || q.period > 0, "gvit_chain: phase %d: position table needs a period", p);
      CFEN_CHECK_ARG(!(q.lnf_s && q.nsplit > 1), "gvit_chain: phase %d: a LayerNorm-folded phase cannot be split over K", p);
      CFEN_CHECK_ARG(q.nsplit == 1 || (c.part && (size_t)tbs * (q.N / 128) * q.nsplit * 8 * 512 * 16 <= c.part_bytes && tbs * (q.N / 128) <= c.ncnt),
                     "gvit_chain: phase %d: split-K scratch too small", p);
      if (q.fold)
        CFEN_CHECK_ARG(c.fC % 4 == 0 && c.fcs % 4 == 0 && c.fp > 0 && c.fH % c.fp == 0 && c.fW % c.fp == 0 && q.N == c.fp * c.fp * c.fC &&
                       c.M % ((c.fH / c.fp) * (c.fW / c.fp)) == 0, "gvit_chain: phase %d: bad fold geometry", p);
      a.ph[p] = GvPhase{(const half_t*)q.X, (const half_t*)q.W, q.bias, q.lnf_s, (const half_t*)q.R, (const half_t*)q.P, (half_t*)q.Y,
                        q.ldx, q.ldr, q.ldy, q.period, q.N, q.K, q.relu, q.nsplit, q.fold};
    }
  }
  CFEN_LAUNCH(k_gvit_chain, dim3(team, ng), dim3(512), 0, s, ga);
  CFEN_CHECK_LAUNCH("gvit_chain");
  return CFEN_OK;
}
