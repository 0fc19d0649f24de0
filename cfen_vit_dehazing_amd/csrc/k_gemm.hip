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
#include "cfen_internal.hpp"

int& cfen_tune_gemm_nt();
int& cfen_tune_gemm_defer_refill();
int& cfen_tune_gemm_splitk_release();
int& cfen_tune_gemm_mid();

namespace {

// XCD-aware block -> tile map.  Blocks are dealt round-robin to the 8 XCDs (own L2 each); XCD x = block & 7 works on one
// cell of a pn x pm partition of the tile grid (pn * pm == 8), so a weight tile is fetched from HBM by pm XCDs and a token
// tile by pn XCDs instead of by all 8.  The host picks the partition that minimises  W bytes * pm + X bytes * pn.
struct TileMap {
  int gn, gm;   // tile grid (feature tiles x token tiles)
  int pn, pm;   // XCD partition
  int cn, cm;   // cell size in tiles: ceil(gn / pn) x ceil(gm / pm); the launch has 8 * cn * cm blocks
};
CFEN_DEV bool tile_of_block(const TileMap& t, unsigned bid, int& tn, int& tm) {
  const int x = (int)(bid & 7u), l = (int)(bid >> 3);
  const int ln = l % t.cn, lm = l / t.cn;   // feature tile fastest: neighbours in time share the token tile
  tn = (x % t.pn) * t.cn + ln;
  tm = (x / t.pn) * t.cm + lm;
  return tn < t.gn && tm < t.gm;
}
static TileMap make_tile_map(int gn, int gm, double w_bytes, double x_bytes) {
  TileMap best{gn, gm, 1, 8, gn, (gm + 7) / 8};
  double best_cost = -1.0;
  for (int pn = 1; pn <= 8; pn *= 2) {
    const int pm = 8 / pn, cn = (gn + pn - 1) / pn, cm = (gm + pm - 1) / pm;
    const double waste = 8.0 * cn * cm / ((double)gn * gm);
    const double cost = (w_bytes * pm + x_bytes * pn) * waste * waste;   // padded cells are idle XCD slots: penalise twice
    if (best_cost < 0 || cost < best_cost) {
      best_cost = cost;
      best = TileMap{gn, gm, pn, pm, cn, cm};
    }
  }
  return best;
}

template <typename T> struct GemmArgs {
  const T* X; const T* W; const float* bias; const T* R; const T* P; T* Y;
  int M, N, K, ldx, ldw, ldr, ldy, period, relu;
  TileMap map;
  // optional token gather (LViT embedding, v3:1140-1143,1166): X -- and R, which is the same tokens -- are not a
  // token-major matrix but windows of patches read straight from the NHWC map (what k_patchify would have written):
  // token m = (image, window, patch), k = (i, j, c) -> pixel (y + i, x + j), channel c.
  const T* gmap;
  int gH, gW, gcs, gC, gws, gp;
  // optional fold (v3:1186 F.fold + Join2x2): Y is not a token-major matrix but the NHWC map the tokens tile -- feature n = (i, j, c) of
  // token m is stored at pixel (y + i, x + j), channel c of `ymap` (same geometry fields; never together with the gather)
  T* ymap;
  // optional split-K (k_gemm_dma only): blockIdx.y = K slice.  Every slice parks its fp32 accumulator tile (and its share of the LayerNorm
  // row sums) in `part` [tile][slice][vector][thread]; the workgroup whose arrival ticket on cnt[tile] is the last one adds the nsplit
  // slabs in slice order -- its own included, from memory -- and runs the epilogue: deterministic, no float atomics, no second launch
  float* part;
  unsigned* cnt;   // one arrival counter per tile, zero between launches (the reducing workgroup puts its tile's back to zero)
  int nsplit;
  int release;       // split-K: agent-scope release fence in every slice (A/B knob "gemm.splitk_release")
  int nt;            // weight rows by non-temporal LDS-DMA
  // optional LayerNorm fold (k_gemm_dma, nsplit == 1, K = the whole row): see CfenGemmPtrs::lnf_s
  const float* lnf_s;
  float lnf_eps;
  int wtile;   // W is tile-major (CfenGemmPtrs::wtile)
  int defer;   // k_gemm_dma: the K-step's refill goes out behind its first fragment reads instead of right behind the barrier ("gemm.defer_refill", round 6)
};

// pointer to channel 0 of the top-left pixel of token m's patch
template <typename T> CFEN_DEV size_t gather_pixoff(const GemmArgs<T>& a, int m) {
  const int tw = a.gws / a.gp, S = tw * tw, nwx = a.gW / a.gws, nwy = a.gH / a.gws;
  const int t = m % S, wi = m / S;
  const int wx = wi % nwx, wy = (wi / nwx) % nwy, b = wi / (nwx * nwy);
  const int y = wy * a.gws + (t / tw) * a.gp, x = wx * a.gws + (t % tw) * a.gp;
  return (((size_t)b * a.gH + y) * a.gW + x) * a.gcs;
}
template <typename T> CFEN_DEV const T* gather_pix(const GemmArgs<T>& a, int m) { return a.gmap + gather_pixoff(a, m); }
// element offset of feature k = (i, j, c) from that pixel
template <typename T> CFEN_DEV int gather_off(const GemmArgs<T>& a, int k) {
  const int ij = k / a.gC, c = k - ij * a.gC;
  return ((ij / a.gp) * a.gW + (ij % a.gp)) * a.gcs + c;
}
// where Y[m][n .. n+3] goes: the token-major row, or (fold) 4 consecutive channels of one pixel of the map
template <typename T> CFEN_DEV T* out_ptr(const GemmArgs<T>& a, long long m, int n) {
  return a.ymap ? a.ymap + gather_pixoff(a, (int)m) + gather_off(a, n) : a.Y + (size_t)m * a.ldy + n;
}

// Epilogue of the 3 x 4 tile block of one wave: the lane owns Y[m + 16 j][n + 16 i .. +3].  All residual / position
// loads are issued before the first store (R may alias Y element for element -- in-place residual -- so the compiler
// must not be left to order them: it would wait for every load separately), bias is read once.
// LN-fold: acc <- rstd_m * (acc - mean_m * s_n) for the lane's (token m + 16 j, features n + 16 i .. +3); stats[row] = (mean, rstd)
template <typename T, int TM, int TN = 3>
CFEN_DEV void gemm_lnfold(const GemmArgs<T>& a, floatx4 (&acc)[TN][TM], int n, const float* stats, int mloc) {
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const floatx4 sl = *reinterpret_cast<const floatx4*>(a.lnf_s + (n + 16 * i < a.N ? n + 16 * i : 0));   // unconditional (clamped), see gemm_epilogue
    const floatx4 sn = n + 16 * i < a.N ? sl : floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const float mean = stats[2 * (mloc + 16 * j)], rstd = stats[2 * (mloc + 16 * j) + 1];
      acc[i][j] = (acc[i][j] - sn * mean) * rstd;
    }
  }
}

template <typename T, int TM, int TN = 3>
CFEN_DEV void gemm_epilogue(const GemmArgs<T>& a, floatx4 (&acc)[TN][TM], int n, int m) {
  typedef typename Mma<T>::out4 out4;
  // Every load below is UNCONDITIONAL per lane, at an address clamped into the operand, inside wave-uniform `if`s; validity is applied when the
  // value is used.  `ok ? load : 0` (and a residual taken from one of two sources by a per-lane select) makes hipcc merge the loaded register
  // with the alternative right after the load: an exec-masked branch and an s_waitcnt vmcnt(0) per load -- the epilogue of the few-token GEMMs
  // was 6-8 serial memory round trips in a kernel of 7-15 us.
  floatx4 bias[TN];
  bool nok[TN];
  int nc[TN];
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    nok[i] = n + 16 * i < a.N;
    nc[i] = nok[i] ? n + 16 * i : 0;
    bias[i] = floatx4{0.f, 0.f, 0.f, 0.f};
  }
  if (a.bias) {
#pragma unroll
    for (int i = 0; i < TN; ++i) bias[i] = *reinterpret_cast<const floatx4*>(a.bias + nc[i]);
  }
  if constexpr (TN > 3) {
    // big wave tiles (96 accumulator registers): one token column at a time, so only TN residual vectors are live beside the accumulators
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int mj = m + 16 * j;
      if (mj >= a.M) continue;
      out4 rv[TN], pv[TN];
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        if (a.gmap && nok[i]) rv[i] = *reinterpret_cast<const out4*>(gather_pix(a, mj) + gather_off(a, n + 16 * i));
        else if (a.R && nok[i]) rv[i] = *reinterpret_cast<const out4*>(a.R + (size_t)mj * a.ldr + n + 16 * i);
        if (a.P && nok[i]) pv[i] = *reinterpret_cast<const out4*>(a.P + (size_t)(mj % a.period) * a.N + n + 16 * i);
      }
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        if (!nok[i]) continue;
        floatx4 v = acc[i][j] + bias[i];
        if (a.relu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        if (a.R || a.gmap) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += (float)rv[i][r];
        }
        if (a.P) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += (float)pv[i][r];
        }
        store4<T>(out_ptr(a, mj, n + 16 * i), v);
      }
    }
    return;
  }
  out4 rv[TN][TM], pv[TN][TM];
  if (a.gmap) {
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const T* px = gather_pix(a, min(m + 16 * j, a.M - 1));
#pragma unroll
      for (int i = 0; i < TN; ++i) rv[i][j] = *reinterpret_cast<const out4*>(px + gather_off(a, nc[i]));
    }
  } else if (a.R) {
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const T* rp = a.R + (size_t)min(m + 16 * j, a.M - 1) * a.ldr;
#pragma unroll
      for (int i = 0; i < TN; ++i) rv[i][j] = *reinterpret_cast<const out4*>(rp + nc[i]);
    }
  }
  if (a.P) {
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const T* pp = a.P + (size_t)(min(m + 16 * j, a.M - 1) % a.period) * a.N;
#pragma unroll
      for (int i = 0; i < TN; ++i) pv[i][j] = *reinterpret_cast<const out4*>(pp + nc[i]);
    }
  }
  // every loaded value is consumed before the first store: a store between two uses makes the next use wait for vmcnt(0), i.e. for the store
#pragma unroll
  for (int j = 0; j < TM; ++j)
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      floatx4 v = acc[i][j] + bias[i];
      if (a.relu) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
      }
      if (a.R || a.gmap) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += (float)rv[i][j][r];
      }
      if (a.P) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += (float)pv[i][j][r];
      }
      acc[i][j] = v;
    }
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int mj = m + 16 * j;
    if (mj >= a.M) continue;
#pragma unroll
    for (int i = 0; i < TN; ++i)
      if (nok[i]) store4<T>(out_ptr(a, mj, nc[i]), acc[i][j]);
  }
}

constexpr int G_BN = 96, G_BM = 128, G_ROWS = G_BN + G_BM;
constexpr int G_BKB = 128;             // bytes of K per row per stage
constexpr int G_ROWB = G_BKB + 16;     // padded LDS row (a 32-byte pad would be conflict-free but overflows 64 KB of static LDS)
constexpr int G_PIECES = G_BKB / 16;   // 16-byte pieces per row
constexpr int G_LOADS = G_ROWS * G_PIECES / 256;   // 7 pieces per thread

template <typename T>
__global__ __launch_bounds__(256) void k_gemm_nt(Grouped<GemmArgs<T>> ga) {
  const GemmArgs<T>& a = ga.g[blockIdx.z];
  constexpr int EPL = Mma<T>::EPL;
  constexpr int BK = G_BKB / (int)sizeof(T);
  constexpr int NCH = BK / Mma<T>::KC;   // fragment chunks per stage (2)
  typedef typename Mma<T>::frag frag;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][G_ROWS * G_ROWB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, h = lane >> 4;
  int tn, tm;
  if (!tile_of_block(a.map, blockIdx.x, tn, tm)) return;
  const int n0 = tn * G_BN, m0 = tm * G_BM;
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
      gptr[i] = a.gmap ? gather_pix(a, m) : a.X + (size_t)m * a.ldx;
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
      const bool xrow = (tid + i * 256) / G_PIECES >= G_BN;
      stage[i] = (k < a.K) ? load_frag<T>(gptr[i] + ((a.gmap && xrow) ? gather_off(a, k) : k)) : Mma<T>::zero();
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

  gemm_epilogue<T>(a, acc, n0 + wn * 48 + 4 * h, m0 + wm * 64 + r16);
}

// Same 128 x 96 block tile, staged by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write_b128 -- the
// 13-cycle wide LDS store of the register-staged kernel above costs as much LDS-path time per K step as all its
// fragment reads).  One wave-instruction lands 1 KiB = 8 rows x 128 B lane-linearly, so the image cannot be padded;
// bank conflicts are avoided by XOR-swizzling the 16-byte piece index with (row & 7) on the SOURCE address and on
// the fragment read address.  Two LDS stages; the DMA of stage k+1 stays in flight across the barriers of stage k
// (counted vmcnt, raw s_barrier).  Needs K * sizeof(T) % 128 == 0.
// one global_load_lds_dwordx4: this lane's 16 bytes at `g` land at (wave-uniform) `l` + 16 * lane
CFEN_DEV void dma16(const void* g, unsigned char* l) {
  __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
// ... with the non-temporal policy (aux = 2): weight rows that one launch reads once ("gemm.nt")
CFEN_DEV void dma16_nt(const void* g, unsigned char* l) {
  __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)l, 16, 0, 2);
}

// wait until at most `younger` K-steps of LOADS LDS-DMAs each are still in flight (younger <= Y)
template <int LOADS, int Y>
CFEN_DEV void gemm_wait_steps(int younger) {
  if constexpr (Y == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    if (younger >= Y) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Y * LOADS) : "memory");
    else gemm_wait_steps<LOADS, Y - 1>(younger);
  }
}

// NS LDS stages form a ring: NS - 1 K-steps of DMA are in flight while one is consumed, one barrier per K-step.
// TN = 6, TM = 4 (192 x 128 block, 96 x 64 per wave): per K-step a wave reads (6 + 4) fragments for 24 MFMAs instead of (3 + 2) for 6
// and the L2 -> LDS traffic per flop halves -- and it is slower on every GEMM of this network, see cfen_tune_gemm_big().
template <typename T, int TM, int NS, int TN = 3>   // block tile = 32*TN features x 32*TM tokens; a wave owns TN x TM MFMA tiles
__global__ __launch_bounds__(256, (TN > 3 && NS == 2) ? 2 : 1) void k_gemm_dma(Grouped<GemmArgs<T>> ga) {
  const GemmArgs<T> a = ga.g[blockIdx.z];   // by value: one bulk scalar load into SGPRs instead of a reload of each field where it is used
  constexpr int EPL = Mma<T>::EPL;
  constexpr int BK = G_BKB / (int)sizeof(T);
  constexpr int NCH = BK / Mma<T>::KC;   // 2
  constexpr int G_BN = 32 * TN;          // (shadows the file-level 96 of the other kernels)
  constexpr int BM = 32 * TM, ROWS = G_BN + BM, LOADS = ROWS / 32;
  constexpr int STAGE = ROWS * G_BKB;    // 28 KiB at TM = 4
  typedef typename Mma<T>::frag frag;
  static_assert(NS * STAGE <= 160 * 1024 && (NS - 2) * LOADS < 64, "LDS / vmcnt range");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[NS * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, h = lane >> 4;
  int tn, tm;
  if (!tile_of_block(a.map, blockIdx.x, tn, tm)) return;
  const int n0 = tn * G_BN, m0 = tm * BM;
  const int wn = wave & 1, wm = wave >> 1;

  // DMA assignment: instruction i of wave w fills rows i*32 + w*8 .. +8; lane -> (row, 16-byte slot)
  const T* gptr[LOADS];
  int gpc[LOADS];
#pragma unroll
  for (int i = 0; i < LOADS; ++i) {
    const int row = i * 32 + wave * 8 + (lane >> 3), slot = lane & 7;
    const int piece = slot ^ (row & 7);
    const bool gx = a.gmap && row >= G_BN;   // gathered token row: gptr = patch origin, the piece offset is added per K-step
    const T* base = row < G_BN ? (a.wtile ? a.W + ((size_t)tn * (a.K / BK) * G_BN + row) * BK : a.W + (size_t)min(n0 + row, a.N - 1) * a.ldw)
                               : gx ? gather_pix(a, min(m0 + row - G_BN, a.M - 1)) : a.X + (size_t)min(m0 + row - G_BN, a.M - 1) * a.ldx;
    gptr[i] = gx ? base : base + piece * EPL;
    gpc[i] = gx ? piece * EPL : -1;
  }
#define CFEN_GEMM_DMA_ISSUE(kt_, buf_)                                                                              \
  _Pragma("unroll") for (int i_ = 0; i_ < LOADS; ++i_) {                                                            \
    if (i_ < TN && a.nt) dma16_nt(gptr[i_] + (kt_) * wstep, lds + (buf_) * STAGE + (i_ * 256 + wave * 64) * 16);      \
    else dma16(gptr[i_] + (i_ < TN ? (kt_) * wstep : gpc[i_] >= 0 ? gather_off(a, (kt_) * BK + gpc[i_]) : (kt_) * BK),         \
               lds + (buf_) * STAGE + (i_ * 256 + wave * 64) * 16);                                                 \
  }

  const int wstep = a.wtile ? G_BN * BK : BK;   // elements from one K-step of a weight row to the next
  floatx4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets: row r, chunk c, quarter h -> r*128 + (((c*4 + h) ^ (r & 7)) << 4)
  int aoff[TN], boff[TM];
#pragma unroll
  for (int i = 0; i < TN; ++i) aoff[i] = (wn * 16 * TN + i * 16 + r16) * G_BKB;
#pragma unroll
  for (int j = 0; j < TM; ++j) boff[j] = (G_BN + wm * 16 * TM + j * 16 + r16) * G_BKB;
  const int sw = r16 & 7;   // every fragment row of this lane has (row & 7) == (r16 & 7): all tile offsets are multiples of 16

  const int nk = a.K / BK / a.nsplit;          // K-steps of this slice
  {
    const int kbeg = (int)blockIdx.y * nk * BK;
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      if (gpc[i] >= 0) gpc[i] += kbeg;         // gathered rows recompute their offset from the absolute k
      else gptr[i] += i < TN ? (int)blockIdx.y * nk * wstep : kbeg;
    }
  }
#pragma unroll
  for (int st = 0; st < NS - 1; ++st)
    if (st < nk) CFEN_GEMM_DMA_ISSUE(st, st);
  // LN-fold: thread (row = tid / 8 (+ 32 per extra token tile), piece = tid % 8) sums x and x^2 of its 16 bytes of every staged X tile
  float ls[TM], lq[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) ls[j] = lq[j] = 0.f;
  const int lnoff = (G_BN + (tid >> 3)) * G_BKB + (tid & 7) * 16;
  int buf = 0, fill = NS - 1;   // ring slots: `buf` is consumed at this step, `fill` receives K-step kt + NS - 1
  for (int kt = 0; kt < nk; ++kt) {
    // K-step kt must have landed; the younger K-steps (at most NS - 2 groups of LOADS DMAs) may stay in flight
    // (round 3: counted for EVERY ring depth -- the round-2 chain stopped at two K-steps in flight, so rings deeper than four stages
    // only cost LDS and the "deeper rings do not help" measurement of DESIGN 4.2 never had more than two K-steps in flight)
    gemm_wait_steps<LOADS, NS - 2>(min(NS - 2, nk - 1 - kt));
    __builtin_amdgcn_s_barrier();   // K-step kt visible to all waves; all waves are done with the slot consumed at kt - 1
    // the refill: an LDS-DMA piece holds the issuing wave for 60-140 cycles (tools/dbg_mlp3_stamps.py).  Deferred, those pass while the wave's first fragment reads are
    // in flight instead of in front of them (k_lvit_window: 376 -> 315 us, round 6)
    const bool refill = kt + NS - 1 < nk;
    if (refill && !a.defer) CFEN_GEMM_DMA_ISSUE(kt + NS - 1, fill);
    const unsigned char* st = lds + buf * STAGE;
    if (a.lnf_s) {
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        float v[EPL];
        Vec16<T>::load(reinterpret_cast<const T*>(st + lnoff + j * 32 * G_BKB), v);
#pragma unroll
        for (int e = 0; e < EPL; ++e) { ls[j] += v[e]; lq[j] += v[e] * v[e]; }
      }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int po = ((c * 4 + h) ^ sw) << 4;
      frag af[TN], bf[TM];
#pragma unroll
      for (int i = 0; i < TN; ++i) af[i] = *reinterpret_cast<const frag*>(st + aoff[i] + po);
#pragma unroll
      for (int j = 0; j < TM; ++j) bf[j] = *reinterpret_cast<const frag*>(st + boff[j] + po);
      if (c == 0 && refill && a.defer) CFEN_GEMM_DMA_ISSUE(kt + NS - 1, fill);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = Mma<T>::mma(af[i], bf[j], acc[i][j]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of `buf` are complete before it reaches the next barrier
    fill = buf;
    buf = buf + 1 == NS ? 0 : buf + 1;
  }

  if (a.nsplit > 1) {
    // In-launch reduction (guide: "Projection GEMM at M = 256", item 2, write-through form): the slab goes out with sc1 (write-through)
    // 16-byte stores, whole 128-byte lines per wave-instruction -> every wave drains its stores -> workgroup barrier -> lane 0 takes a relaxed
    // agent-scope ticket (no release fence: a buffer_wbl2 in every one of ~1000 workgroups cost 3x the GEMM itself, measured) -> the last
    // arriver: agent-scope acquire, drain, barrier, plain loads.  Correct for any placement of a tile's slices over CUs / XCDs; nobody waits
    // for anybody (no spin), so it cannot hang.
    constexpr int NV = TN * TM;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int tile = tm * a.map.gn + tn;
    const size_t slab = (size_t)(NV * 4 + 2 * TM) * 256;   // floats per (tile, slice): NV accumulator vectors, then (sum, sum of squares) per token tile
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(a.part, 0, -1, 0x00020000);
    const unsigned pbase = (unsigned)(((size_t)tile * a.nsplit + blockIdx.y) * slab * 4);   // byte offset of this slice's slab (< 2^32: checked on the host)
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const floatx4 v = acc[i][j];
        const u32x4 bits = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
        __builtin_amdgcn_raw_buffer_store_b128(bits, prs, (int)(pbase + ((i * TM + j) * 256 + tid) * 16), 0, 16);   // aux 16 = sc1
      }
    if (a.lnf_s) {
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ls[j]), prs, (int)(pbase + (NV * 1024 + (2 * j) * 256 + tid) * 4), 0, 16);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(lq[j]), prs, (int)(pbase + (NV * 1024 + (2 * j + 1) * 256 + tid) * 4), 0, 16);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // "gemm.splitk_release" = 1 (A/B knob): the guide's first recipe -- an agent-scope release (buffer_wbl2) in every slice on top of the
    // write-through stores.  The shipped form (0) relies on what MI355X_MICROARCH.md documents for gfx950: sc1 stores leave the XCD's L2 as
    // they are issued, so the drain above is the whole publish.  Hardware-specific; measured 57.7 us against 22.0 for a 1024-workgroup launch.
    if (a.release) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    unsigned* flag = reinterpret_cast<unsigned*>(lds);     // the ring is free: every wave is past its last fragment read
    if (tid == 0) *flag = __hip_atomic_fetch_add(a.cnt + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*flag != (unsigned)(a.nsplit - 1)) return;
    // EVERY wave of the reducing workgroup acquires at agent scope before it touches another slice's slab (round 4): the
    // invalidate is per CU, so one lane's would do on gfx950 -- but then three waves would read other workgroups' data with no acquire of
    // their own in program order, a data race in the HIP memory model.  One buffer_inv per wave of ONE workgroup per tile costs nothing measurable.
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) __hip_atomic_store(a.cnt + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero again for the next launch that uses this counter
    __syncthreads();
    const float* p0 = a.part + (size_t)tile * a.nsplit * slab;
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] = *reinterpret_cast<const floatx4*>(p0 + ((i * TM + j) * 256 + tid) * 4);
    if (a.lnf_s) {
#pragma unroll
      for (int j = 0; j < TM; ++j) { ls[j] = p0[NV * 1024 + (2 * j) * 256 + tid]; lq[j] = p0[NV * 1024 + (2 * j + 1) * 256 + tid]; }
    }
    for (int sl = 1; sl < a.nsplit; ++sl) {
      const float* ps = p0 + (size_t)sl * slab;
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] += *reinterpret_cast<const floatx4*>(ps + ((i * TM + j) * 256 + tid) * 4);
      if (a.lnf_s) {
#pragma unroll
        for (int j = 0; j < TM; ++j) { ls[j] += ps[NV * 1024 + (2 * j) * 256 + tid]; lq[j] += ps[NV * 1024 + (2 * j + 1) * 256 + tid]; }
      }
    }
  }
  if (a.lnf_s) {   // row statistics: reduce the 8 threads of a row (8 consecutive lanes), publish (mean, rstd) per row through LDS
    __builtin_amdgcn_s_barrier();            // every wave is done reading the last stage: the ring is free
    float* stats = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      float sv = ls[j], qv = lq[j];
      sv += dpp_mov<0xB1>(sv); qv += dpp_mov<0xB1>(qv);
      sv += dpp_mov<0x4E>(sv); qv += dpp_mov<0x4E>(qv);
      sv += dpp_mov<0x141>(sv); qv += dpp_mov<0x141>(qv);
      if ((tid & 7) == 0) {
        const float mean = sv / (float)a.K;
        const float var = fmaxf(qv / (float)a.K - mean * mean, 0.f);
        stats[2 * ((tid >> 3) + 32 * j)] = mean;
        stats[2 * ((tid >> 3) + 32 * j) + 1] = rsqrtf(var + a.lnf_eps);
      }
    }
    __syncthreads();
    gemm_lnfold<T, TM, TN>(a, acc, n0 + wn * 16 * TN + 4 * h, stats, wm * 16 * TM + r16);
  }
  gemm_epilogue<T, TM, TN>(a, acc, n0 + wn * 16 * TN + 4 * h, m0 + wm * 16 * TM + r16);
#undef CFEN_GEMM_DMA_ISSUE
}

// Small-M variant (GViT: 128..2048 tokens per batch against weight matrices of up to 6144 x 1536): the
// problem is weight-bandwidth bound, so the grid is cut for parallelism instead of reuse -- one workgroup
// per 16 output features x 64 tokens, its 4 waves split K and reduce through LDS; operands go straight
// from global memory into MFMA fragments (32 contiguous bytes per lane per step), no staging, no barrier
// in the loop.  Same epilogue as k_gemm_nt.
template <typename T>
__global__ __launch_bounds__(256) void k_gemm_skinny(Grouped<GemmArgs<T>> ga) {
  const GemmArgs<T> a = ga.g[blockIdx.z];
  constexpr int EPL = Mma<T>::EPL, KS = 2 * Mma<T>::KC;
  typedef typename Mma<T>::frag frag;
  __shared__ floatx4 red[4][4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, h = lane >> 4;
  int tn, tm;
  if (!tile_of_block(a.map, blockIdx.x, tn, tm)) return;
  const int n0 = tn * 16, m0 = tm * 64;
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
  store4<T>(out_ptr(a, m, n), v);
}

template <typename T>
int launch_gemm(int ng, const CfenGemmPtrs* gp, int ldx, int ldw, int ldr, int period, int ldy, int M, int N, int K, int relu, hipStream_t s,
                const CfenTokGather* tg, float* const* splitk_ws, size_t splitk_ws_bytes, const CfenTokGather* yg, int force_nsplit) {
  constexpr int EPL = Mma<T>::EPL;
  CFEN_CHECK_ARG(ng >= 1 && ng <= CFEN_MAX_GROUPS && gp, "gemm: 1..%d problems per launch", CFEN_MAX_GROUPS);
  if (yg) {   // Y is folded into an NHWC map (gp[g].ymap): token m = (image, window, patch), feature n = (i, j, c)
    CFEN_CHECK_ARG(yg->C % 4 == 0 && yg->cs % 4 == 0 && yg->cs >= yg->C && yg->p > 0 && yg->ws % yg->p == 0 && yg->H % yg->ws == 0 && yg->W % yg->ws == 0,
                   "gemm (fold): bad token geometry");
    CFEN_CHECK_ARG(N == yg->p * yg->p * yg->C, "gemm (fold): needs N == p*p*C");
    const int tw = yg->ws / yg->p;
    CFEN_CHECK_ARG(M == yg->B * (yg->H / yg->ws) * (yg->W / yg->ws) * tw * tw, "gemm (fold): M does not match the map");
  }
  if (tg) {   // X and R are the patch tokens of an NHWC map (gp[g].gmap)
    CFEN_CHECK_ARG(tg->C % EPL == 0 && tg->cs % EPL == 0 && tg->cs >= tg->C && tg->p > 0 && tg->ws % tg->p == 0 && tg->H % tg->ws == 0 &&
                   tg->W % tg->ws == 0, "gemm (gather): bad token geometry");
    CFEN_CHECK_ARG(K == tg->p * tg->p * tg->C && N == K, "gemm (gather): needs N == K == p*p*C");
    const int tw = tg->ws / tg->p;
    CFEN_CHECK_ARG(M == tg->B * (tg->H / tg->ws) * (tg->W / tg->ws) * tw * tw, "gemm (gather): M does not match the map");
    ldx = K;
  }
  CFEN_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm: empty problem M=%d N=%d K=%d", M, N, K);
  CFEN_CHECK_ARG(N % 4 == 0 && K % EPL == 0, "gemm: N (%d) must be a multiple of 4 and K (%d) of %d", N, K, EPL);
  CFEN_CHECK_ARG(ldx % EPL == 0 && ldw % EPL == 0 && ldy % 4 == 0 && ldx >= K && ldw >= K && ldy >= N, "gemm: bad leading dimension");
  Grouped<GemmArgs<T>> ga;
  memset(&ga, 0, sizeof(ga));
  for (int g = 0; g < ng; ++g) {
    const CfenGemmPtrs& q = gp[g];
    CFEN_CHECK_ARG((tg ? (q.gmap && !q.R) : q.X != nullptr) && q.W && q.Y, "gemm: null operand (problem %d)", g);
    CFEN_CHECK_ARG(!q.R || (ldr % 4 == 0 && ldr >= N), "gemm: bad residual leading dimension");
    CFEN_CHECK_ARG(cfen_aligned16(q.X) && cfen_aligned16(q.W) && cfen_aligned16(q.Y) && cfen_aligned16(q.R) && cfen_aligned16(q.P) &&
                   cfen_aligned16(q.bias) && cfen_aligned16(q.gmap), "gemm: pointers must be 16-byte aligned");
    CFEN_CHECK_ARG(!q.P || period > 0, "gemm: position table needs a period");
    CFEN_CHECK_ARG((q.R == nullptr) == (gp[0].R == nullptr) && (q.P == nullptr) == (gp[0].P == nullptr) && (q.bias == nullptr) == (gp[0].bias == nullptr),
                   "gemm: grouped problems must use the same epilogue operands");
    GemmArgs<T>& a = ga.g[g];
    a.X = (const T*)(tg ? q.gmap : q.X); a.W = (const T*)q.W; a.bias = q.bias; a.R = (const T*)q.R; a.P = (const T*)q.P; a.Y = (T*)q.Y;
    a.M = M; a.N = N; a.K = K; a.ldx = ldx; a.ldw = ldw; a.ldr = ldr; a.ldy = ldy; a.period = period; a.relu = relu;
    a.nsplit = 1;
    a.nt = 0;
    a.lnf_s = q.lnf_s;
    a.lnf_eps = cfen_gemm_lnf_eps();
    a.wtile = q.wtile;
    a.defer = cfen_tune_gemm_defer_refill();
    CFEN_CHECK_ARG(q.wtile == gp[0].wtile, "gemm: grouped problems must share the weight layout");
    if (tg) {
      a.gmap = (const T*)q.gmap; a.gH = tg->H; a.gW = tg->W; a.gcs = tg->cs; a.gC = tg->C; a.gws = tg->ws; a.gp = tg->p;
    }
    if (yg) {
      CFEN_CHECK_ARG(!tg && q.ymap && cfen_aligned16(q.ymap), "gemm (fold): needs an aligned output map and no input gather");
      a.ymap = (T*)q.ymap; a.gH = yg->H; a.gW = yg->W; a.gcs = yg->cs; a.gC = yg->C; a.gws = yg->ws; a.gp = yg->p;
    }
  }
  const bool lnf = gp[0].lnf_s != nullptr;
  for (int g = 0; g < ng; ++g)
    CFEN_CHECK_ARG((gp[g].lnf_s != nullptr) == lnf && cfen_aligned16(gp[g].lnf_s), "gemm: grouped problems must all (or none) fold a LayerNorm");
  const int forced = lnf ? -1 : cfen_tune_gemm_kernel();
  const bool k128 = (K * (int)sizeof(T)) % G_BKB == 0;
  CFEN_CHECK_ARG(!lnf || (k128 && !tg), "gemm (LayerNorm fold): needs K * sizeof(T) %% 128 == 0 and a plain token matrix");
  CFEN_CHECK_ARG(forced <= 0 || k128, "gemm: k_gemm_skinny / k_gemm_dma need K * sizeof(T) %% 128 == 0");
  // Kernel choice from kernel times measured with COLD caches (tools/bench_gemm_cold.py: in the network the 540 MB of
  // weights stream from HBM, a warm-cache microbenchmark picks the wrong variants), MI355X, batch 8:
  //   K not a multiple of 128 bytes (LViT level 1, K = 96)            -> register-staged k_gemm_nt (zero-fills the K tail)
  //   <= 128 tokens against <= 2048 features (GViT-3 square / K-heavy) -> k_gemm_skinny (in-workgroup split-K)
  //   >= 1024 tiles of 96 x 64 per launch (LViT)                      -> k_gemm_dma 96 x 64: occupancy (4 WG/CU) wins
  //   otherwise (GViT)                                                -> k_gemm_dma 96 x 32
  const long long tiles64 = (long long)ng * ((N + G_BN - 1) / G_BN) * ((M + 63) / 64);
  const long long tiles32 = (long long)ng * ((N + G_BN - 1) / G_BN) * ((M + 31) / 32);
  int kern = forced < 0 ? -1 : forced % 10, stages = forced < 0 ? 2 : 2 + forced / 10;
  // Few-token GEMMs against big matrices (GViT levels 2 and 3: 128 / 512 tokens, 0.5 GB of weights per forward).  A workgroup's LDS-DMA
  // stream runs at ~22 GB/s whatever its ring depth (measured, round 3: 96 x 32 and 96 x 128 tiles, 3 to 8 stages), so what sets the time is
  // the bytes ONE workgroup moves and how many workgroups share the chip: K is cut into slices of >= 4 K-steps until the launch has several
  // workgroups per CU, 96 x 32 tiles so that a slice's partial tile is 12 KB and the in-launch reduction (k_gemm_dma) stays cheap.
  int nsplit = 1;
  if (force_nsplit > 1 || (force_nsplit == 0 && forced < 0 && k128 && splitk_ws && !tg && M <= 512 && cfen_tune_gemm_splitk())) {
    const int nk = K / (G_BKB / (int)sizeof(T));
    const long long base = (long long)ng * ((N + G_BN - 1) / G_BN) * ((M + 31) / 32);
    if (force_nsplit > 1) {
      CFEN_CHECK_ARG(k128 && !tg && splitk_ws && nk % force_nsplit == 0, "gemm (split-K): needs K steps (%d) divisible by nsplit (%d), scratch, no gather", nk, force_nsplit);
      nsplit = force_nsplit;
    } else {
      // measured (MI355X, batch 8, GViT-3 at 128 tokens): splitting pays where the unsplit launch leaves CUs idle (embed / proj 64 workgroups:
      // 17.1 -> 13.8 us, ffn2 / head2 27.1 -> 18.8 us against the round-2 split-K with its second launch) and costs where the launch already
      // has a workgroup per CU (ffn1 at 256 workgroups cut 4 ways: 17.6 -> 22.0 us: slab traffic, the acquire and the serial tail of the reducer)
      while (base * nsplit < 256 && nsplit < 8 && nk % (2 * nsplit) == 0 && nk / (2 * nsplit) >= 4 && base * 2 * nsplit <= 512) nsplit *= 2;
    }
    const size_t slab = (size_t)(3 * 4 + 2) * 256 * sizeof(float);
    const long long tiles = (long long)((N + G_BN - 1) / G_BN) * ((M + 31) / 32);
    // the kernel addresses its slab with a 32-bit byte offset from the scratch base
    CFEN_CHECK_ARG((unsigned long long)tiles * (unsigned long long)nsplit * slab < (1ull << 32), "gemm (split-K): %lld tiles x %d slices exceed the 4 GiB slab window", tiles, nsplit);
    if (nsplit > 1 && (tiles > CFEN_SPLITK_COUNTERS || CFEN_SPLITK_COUNTERS * sizeof(unsigned) + (size_t)tiles * nsplit * slab > splitk_ws_bytes)) {
      CFEN_CHECK_ARG(force_nsplit <= 1, "gemm (split-K): scratch too small (%zu bytes)", splitk_ws_bytes);
      nsplit = 1;
    }
    if (nsplit > 1) { kern = 5; stages = 2 + cfen_tune_gemm_small() / 10; }
  }
  // many tokens against >= 768 features (LViT-3 / GViT-1 qkv, ffn1, head1): 192 x 128 tiles when they still fill the chip
  const long long tiles_big = (long long)ng * ((N + 191) / 192) * ((M + 127) / 128);
  if (kern < 0 && k128 && cfen_tune_gemm_big() > 0 && N >= 768 && tiles_big >= cfen_tune_gemm_big_min_tiles()) {
    kern = 6;
    stages = 2 + cfen_tune_gemm_big() / 10;
  }
  if (kern < 0) {
    const int pick = !k128 ? 0 : (M <= 128 && N <= 2048 && !tg && !lnf) ? 1 : tiles64 >= 1024 ? cfen_tune_gemm_large() : tiles32 <= 512 ? cfen_tune_gemm_small() : cfen_tune_gemm_mid();
    kern = pick % 10;
    stages = 2 + pick / 10;
  }
  // experiment knob: few-token problems (M <= 128) take this tile id (2 = 96 x 128: every weight byte is pulled by ONE workgroup)
  if (cfen_tune_gemm_m128() > 0 && forced < 0 && nsplit == 1 && M <= 128 && k128 && !tg) {
    kern = cfen_tune_gemm_m128() % 10;
    stages = 2 + cfen_tune_gemm_m128() / 10;
  }
  if (gp[0].wtile) {   // tile-major weights: the LDS-DMA kernel with 96-feature tiles (the skinny / register-staged kernels read rows)
    CFEN_CHECK_ARG(k128, "gemm (tile-major weights): needs K * sizeof(T) %% 128 == 0");
    if (kern == 1) { kern = 5; stages = 2 + cfen_tune_gemm_small() / 10; }
    CFEN_CHECK_ARG(kern >= 2 && kern <= 5, "gemm (tile-major weights): kernel %d reads row-major weights", kern);
  }
  CFEN_CHECK_ARG(!(tg && kern == 1), "gemm (gather): k_gemm_skinny does not gather");
  if (kern < 2) stages = 2;
  const int bn = kern == 1 ? 16 : kern == 6 ? 192 : G_BN, bm = kern == 0 || kern == 2 || kern == 6 ? 128 : kern == 3 ? 96 : kern == 4 || kern == 1 ? 64 : 32;
  const TileMap map = make_tile_map((N + bn - 1) / bn, (M + bm - 1) / bm, (double)N * K * sizeof(T), (double)M * K * sizeof(T));
  for (int g = 0; g < ng; ++g) {
    ga.g[g].map = map;
    ga.g[g].nsplit = nsplit;
    ga.g[g].release = cfen_tune_gemm_splitk_release();
    ga.g[g].nt = (cfen_tune_gemm_nt() == 2 || (cfen_tune_gemm_nt() == 1 && M <= 512)) ? 1 : 0;
    CFEN_CHECK_ARG(nsplit == 1 || (splitk_ws[g] && cfen_aligned16(splitk_ws[g])), "gemm: split-K workspace missing");
    ga.g[g].cnt = nsplit > 1 ? reinterpret_cast<unsigned*>(splitk_ws[g]) : nullptr;                       // [CFEN_SPLITK_COUNTERS] arrival counters, zero
    ga.g[g].part = nsplit > 1 ? splitk_ws[g] + CFEN_SPLITK_COUNTERS * sizeof(unsigned) / sizeof(float) : nullptr;   // then the partial slabs
  }
  const long long blocks = 8LL * map.cn * map.cm;
  CFEN_CHECK_ARG(blocks < (1LL << 31), "gemm: problem too large for one launch");
  const dim3 grid((unsigned)blocks, (unsigned)nsplit, (unsigned)ng);
  switch (kern + 10 * (stages - 2)) {
    case 0: case 10: case 20: CFEN_LAUNCH(k_gemm_nt<T>, grid, dim3(256), 0, s, ga); break;
    case 1: case 11: case 21: CFEN_LAUNCH(k_gemm_skinny<T>, grid, dim3(256), 0, s, ga); break;
    case 2: case 12: case 22: CFEN_LAUNCH((k_gemm_dma<T, 4, 2>), grid, dim3(256), 0, s, ga); break;
    case 3: case 13: case 23: CFEN_LAUNCH((k_gemm_dma<T, 3, 2>), grid, dim3(256), 0, s, ga); break;
    case 4: CFEN_LAUNCH((k_gemm_dma<T, 2, 2>), grid, dim3(256), 0, s, ga); break;
    case 14: case 24: CFEN_LAUNCH((k_gemm_dma<T, 2, 3>), grid, dim3(256), 0, s, ga); break;
    case 5: CFEN_LAUNCH((k_gemm_dma<T, 1, 2>), grid, dim3(256), 0, s, ga); break;
    case 15: CFEN_LAUNCH((k_gemm_dma<T, 1, 3>), grid, dim3(256), 0, s, ga); break;
    case 25: CFEN_LAUNCH((k_gemm_dma<T, 1, 4>), grid, dim3(256), 0, s, ga); break;
    case 45: CFEN_LAUNCH((k_gemm_dma<T, 1, 6>), grid, dim3(256), 0, s, ga); break;
    case 65: CFEN_LAUNCH((k_gemm_dma<T, 1, 8>), grid, dim3(256), 0, s, ga); break;
    case 34: CFEN_LAUNCH((k_gemm_dma<T, 2, 5>), grid, dim3(256), 0, s, ga); break;
    case 32: CFEN_LAUNCH((k_gemm_dma<T, 4, 5>), grid, dim3(256), 0, s, ga); break;
    case 6: case 16: case 26: CFEN_LAUNCH((k_gemm_dma<T, 4, 2, 6>), grid, dim3(256), 0, s, ga); break;
    default: CFEN_LAUNCH((k_gemm_dma<T, 1, 4>), grid, dim3(256), 0, s, ga); break;
  }
  CFEN_CHECK_LAUNCH("gemm");
  return CFEN_OK;
}

}  // namespace

float& cfen_gemm_lnf_eps() {
  static float v = 1e-5f;
  return v;
}
int& cfen_tune_gemm_defer_refill() {   // k_gemm_dma: 1 = a K-step's refill behind its first fragment reads, 0 = right behind the barrier (rounds 2-5) ("gemm.defer_refill")
  static int v = 1;
  return v;
}
int& cfen_tune_gemm_nt() {   // weight rows of k_gemm_dma by non-temporal LDS-DMA: 0 never, 1 few-token GEMMs (M <= 512: GViT levels 2 and 3), 2 always ("gemm.nt")
  static int v = 0;
  return v;
}

int& cfen_tune_gemm_splitk() {   // 0 (default, round 4): with several forwards in flight the unsplit launches are faster (2.13 against 2.16 ms per step, and 2.80 with
  static int v = 0;               // the release fence below: a split launch is more workgroups, and CU-time is what a forward costs there); 1 = round 3's shape rule
  return v;
}
int& cfen_tune_gemm_splitk_release() {   // 1 (default): every K slice runs an agent-scope release fence before its arrival ticket -- the memory model's recipe;
  static int v = 1;                       // 0 = round 3's publish (write-through slab stores + a vmcnt drain, the guide's "measured, not an architectural guarantee" row):
  return v;                               // also right in every test here, 57.7 -> 22.0 us on a 1024-workgroup launch.  (Round 4 blamed this seam for wrong outputs with two
}                                         // forwards in flight; the cause was the counters' zeroing as a graph MEMSET node, cfen_api.cpp: cfen_zero_async.)
// 0 (default): off; 6: 192 x 128 tiles, 2-stage ring (80 KB of LDS, two workgroups a CU).  MEASURED (MI355X, B = 8, round 2): the
// LViT-3 / GViT-1 qkv, ffn1, head1 GEMMs are 5 - 25 % SLOWER on it (ln1_qkv x3 60 -> 76 us, ln2_ffn1 x3 75 -> 80 us; a 3-stage
// one-workgroup-a-CU variant 104 / 119 us): with K = 384 a tile is 6 dependent K-steps, the kernel is bound by the latency of that
// chain and what hides it is the number of workgroups a CU holds, not the bytes or LDS reads per flop.  Kept as a tested variant.
int& cfen_tune_gemm_big() {
  static int v = 0;
  return v;
}
int& cfen_tune_gemm_big_min_tiles() {   // the 192 x 128 tile is used when a launch has at least this many of them
  static int v = 256;
  return v;
}
int& cfen_tune_gemm_splitk_stages() {   // extra ring stages of the 96 x 128 split-K tile: 0 (two stages) or 3 (five stages, 140 KB)
  static int v = 0;
  return v;
}
int& cfen_tune_gemm_m128() {
  static int v = 0;
  return v;
}
int& cfen_tune_gemm_large() {
  static int v = 4;
  return v;
}
int& cfen_tune_gemm_mid() {   // more than 512 tiles of 96 x 32 and fewer than 1024 of 96 x 64 (the grouped GViT-2 decoder GEMMs, 3 x 512 tokens): kernel id as "gemm.small".
  static int v = 2;            // Round 5, four forwards in flight, same box: 96 x 128 tiles (2) 2.059 / 2.060 ms per step, 96 x 64 (4) 2.094, 96 x 96 (3) 2.106, 96 x 32 on
  return v;                    // 2 / 3 stages (5 / 15: rounds 2-4) 2.075-2.078 / 2.136 (profiles/r05_ab_gvit2_decoder_gemm_tiles.txt): a quarter of the weight bytes through the DMA path
}
int& cfen_tune_gemm_small() {
  static int v = 15;   // 96 x 32 tiles, 3-stage ring: the few-token GViT GEMMs are latency bound (one K-step per memory round trip with 2 stages)
  return v;
}
int& cfen_tune_gemm_kernel() {
  static int v = -1;
  return v;
}

int cfen_gemm_impl_g(int dtype, int ng, const CfenGemmPtrs* gp, int ldx, int ldw, int ldr, int period, int ldy, int M, int N, int K, int relu,
                     const CfenTokGather* tg, hipStream_t s, float* const* splitk_ws, size_t splitk_ws_bytes, const CfenTokGather* yg, int force_nsplit) {
  if (dtype == 1) return launch_gemm<half_t>(ng, gp, ldx, ldw, ldr, period, ldy, M, N, K, relu, s, tg, splitk_ws, splitk_ws_bytes, yg, force_nsplit);
  if (dtype == 0) return launch_gemm<float>(ng, gp, ldx, ldw, ldr, period, ldy, M, N, K, relu, s, tg, splitk_ws, splitk_ws_bytes, yg, force_nsplit);
  cfen_set_error("gemm: unknown dtype %d", dtype);
  return CFEN_ERR_ARG;
}

int cfen_gemm_impl(int dtype, const void* X, int ldx, const void* W, int ldw, const float* bias, const void* R, int ldr,
                   const void* P, int period, void* Y, int ldy, int M, int N, int K, int relu, hipStream_t s) {
  const CfenGemmPtrs q{X, W, bias, R, P, Y, nullptr, nullptr};
  return cfen_gemm_impl_g(dtype, 1, &q, ldx, ldw, ldr, period, ldy, M, N, K, relu, nullptr, s, nullptr, 0, nullptr);
}

int cfen_embed_gather_impl(int dtype, const CfenTokGather* tg, const void* W, int ldw, const float* bias, const void* P, int period,
                           void* Y, int ldy, int M, hipStream_t s) {
  CFEN_CHECK_ARG(tg != nullptr, "embed_gather: null geometry");
  const int D = tg->p * tg->p * tg->C;
  const CfenGemmPtrs q{nullptr, W, bias, nullptr, P, Y, tg->map, nullptr};
  return cfen_gemm_impl_g(dtype, 1, &q, D, ldw, 0, period, ldy, M, D, D, 0, tg, s, nullptr, 0, nullptr);
}
