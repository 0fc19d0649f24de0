// Fused token MLP block:   y1 = x  + W2 relu(W1 LN(x) + b1) + b2      (encoder FFN,  v3:1387-1389)
//                           y2 = y1 + Wh2 relu(Wh1 y1 + bh1) + bh2     (mlp_head,     v3:1173)
//                           fold(y2) -> NHWC feature map                (v3:1176,1186 + Join2x2 v3:1046-1056)
// for LViT instances with D in {96, 192} (levels 1 and 2).
//
// Each wave owns TM*16 tokens for the whole chain; tokens never leave its registers:
//   * the residual stream lives as fp32 MFMA accumulators acc[n-tile][m-tile] (rows = features, cols = tokens),
//     D/16 * TM = 24 tiles for every supported D (TM = 4 / 2);
//   * LayerNorm reduces over features = over a lane's registers + two xor-shuffles (lanes l^16, l^32);
//   * an accumulator tile pair IS the B operand of the next GEMM (k-slot (h,j) of a 32-deep chunk <- rows
//     4h+j of tile a, j<4, and of tile b, j>=4), so LN(x) feeds FFN1 and relu(FFN1) feeds FFN2 with no LDS
//     or HBM round trip; the hidden activation (M x 4D, the largest tensor of the block) is never stored.
//     The matching permutation of the weights' k axis is applied once on the host (packing.kperm32).
//   * weights stream through LDS in hidden-dim chunks shared by the workgroup's 4 waves, with the next
//     chunk's global loads in flight during the MFMAs (register prefetch, two barriers per chunk).
// HBM traffic per instance: read x once, write y2 once (+ weights from L2).
#include <stdlib.h>
#include <type_traits>
#include "cfen_common.hpp"
#include "cfen_mlp.hpp"
#include "cfen_internal.hpp"

namespace {

template <typename T> struct Pack;
template <> struct Pack<half_t> {
  static constexpr int NPC = 2;   // accumulator n-tiles per K chunk
  static CFEN_DEV half8 make(const floatx4* t) {   // t[0], t[1]: tiles a, b of the same token tile
    half8 f = {(half_t)t[0][0], (half_t)t[0][1], (half_t)t[0][2], (half_t)t[0][3],
               (half_t)t[1][0], (half_t)t[1][1], (half_t)t[1][2], (half_t)t[1][3]};
    return f;
  }
};
template <> struct Pack<float> {
  static constexpr int NPC = 1;
  static CFEN_DEV floatx4 make(const floatx4* t) { return t[0]; }
};

// ND = D/16 n-tiles, TM = token tiles per wave, NW = waves per workgroup, HCH = hidden units per LDS stage
// WPE = waves per SIMD the register allocation must leave room for
template <typename T, int ND, int TM, int NW, int HCH, int WPE>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_mlp(Grouped<MlpArgs> ga) {
  const MlpArgs& a = ga.g[blockIdx.z];
  constexpr int KC = Mma<T>::KC, EPL = Mma<T>::EPL, SZ = (int)sizeof(T);
  constexpr int NPC = Pack<T>::NPC;
  constexpr int NT = NW * 64;
  constexpr int D = ND * 16;
  constexpr int NCH = ND / NPC;                   // K chunks over D
  constexpr int NSUB = HCH / KC;                  // MFMA K chunks per LDS stage
  constexpr int W1ROW = D * SZ + 16;              // LDS row of the W1 slice (HCH rows)
  constexpr int W2ROW = HCH * SZ + 16;            // LDS row of the W2 slice (D rows)
  constexpr int W1BYTES = HCH * W1ROW;
  constexpr int STAGE = W1BYTES + D * W2ROW;
  constexpr int P1 = HCH * (D * SZ / 16);         // 16-byte pieces of the W1 slice
  constexpr int P2 = D * (HCH * SZ / 16);         // ... of the W2 slice
  constexpr int NPF = (P1 + P2) / NT;             // prefetch pieces per thread
  static_assert((P1 + P2) % NT == 0 && HCH % KC == 0, "slice pieces must split evenly over the workgroup");
  typedef typename Mma<T>::frag frag;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // 2 stages

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, h = lane >> 4;
  const long long tok0 = ((long long)blockIdx.x * NW + wave) * (TM * 16);

  // ---- load x^T into accumulator layout: acc[i][j][r] = x[token j*16+r16][feature i*16+4h+r] ----
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

  if (a.Wp) {   // x += Wp att: the attention block's out_proj + residual, operands straight from L1/L2 (natural k order)
    const T* Wp = (const T*)a.Wp + (size_t)r16 * D + h * EPL;
    constexpr int NKC = D / KC;
    const T* ap[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      long long t = tok0 + j * 16 + r16;
      if (t >= a.M) t = a.M - 1;
      ap[j] = (const T*)a.A + t * D + h * EPL;
    }
#pragma unroll
    for (int c = 0; c < NKC; ++c) {   // k chunk outermost: only TM token fragments are live at a time
      frag ab[TM];
#pragma unroll
      for (int j = 0; j < TM; ++j) ab[j] = load_frag<T>(ap[j] + c * KC);
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        const frag af = load_frag<T>(Wp + (size_t)i * 16 * D + c * KC);
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = Mma<T>::mma(af, ab[j], acc[i][j]);
      }
    }
  }

  frag xb[NCH][TM];
#pragma unroll 1
  for (int stage = 0; stage < 2; ++stage) {
    const T* W1 = (const T*)(stage ? a.W1b : a.W1a);
    const T* W2 = (const T*)(stage ? a.W2b : a.W2a);
    const float* b1 = stage ? a.b1b : a.b1a;
    const float* b2 = stage ? a.b2b : a.b2a;
    if (!W1) break;
    // ---- B fragments of the stage input (LayerNorm first for the FFN stage) ----
    if (stage == 0 && a.ln_g) {
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
            float d = acc[i][j][r] - mean;
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
          xb[c][j] = Pack<T>::make(t);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          floatx4 t[NPC];
#pragma unroll
          for (int u = 0; u < NPC; ++u) t[u] = acc[c * NPC + u][j];
          xb[c][j] = Pack<T>::make(t);
        }
    }
    // residual + output bias: acc <- x + b2, FFN2 accumulates on top
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const floatx4 bb = *reinterpret_cast<const floatx4*>(b2 + i * 16 + 4 * h);
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] += bb;
    }

    // ---- hidden-dim stage loop: double-buffered LDS, next stage's loads in flight during the MFMAs ----
    const int nhc = a.H / HCH;
    frag pf[NPF];
    auto prefetch = [&](int hc) {
#pragma unroll
      for (int u = 0; u < NPF; ++u) {
        const int id = tid + u * NT;
        if (id < P1) {   // W1 slice: rows hc*HCH.., D elements each
          const int row = id / (D * SZ / 16), pc = id % (D * SZ / 16);
          pf[u] = load_frag<T>(W1 + (size_t)(hc * HCH + row) * D + pc * EPL);
        } else {         // W2 slice: D rows, HCH elements at column hc*HCH
          const int id2 = id - P1;
          const int row = id2 / (HCH * SZ / 16), pc = id2 % (HCH * SZ / 16);
          pf[u] = load_frag<T>(W2 + (size_t)row * a.H + hc * HCH + pc * EPL);
        }
      }
    };
    auto commit = [&](unsigned char* buf) {
#pragma unroll
      for (int u = 0; u < NPF; ++u) {
        const int id = tid + u * NT;
        if (id < P1) {
          const int row = id / (D * SZ / 16), pc = id % (D * SZ / 16);
          *reinterpret_cast<frag*>(buf + row * W1ROW + pc * 16) = pf[u];
        } else {
          const int id2 = id - P1;
          const int row = id2 / (HCH * SZ / 16), pc = id2 % (HCH * SZ / 16);
          *reinterpret_cast<frag*>(buf + W1BYTES + row * W2ROW + pc * 16) = pf[u];
        }
      }
    };
    prefetch(0);
    __syncthreads();                 // previous MLP stage is done with both buffers
    commit(lds);
    __syncthreads();
#pragma unroll 1
    for (int hc = 0; hc < nhc; ++hc) {
      const unsigned char* buf = lds + (hc & 1) * STAGE;
      if (hc + 1 < nhc) prefetch(hc + 1);
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub) {
        // FFN1 slice: hidden[KC units][tokens] = W1 slice . LN(x)^T + b1
        floatx4 hacc[NPC][TM];
#pragma unroll
        for (int u = 0; u < NPC; ++u) {
          const floatx4 bb = *reinterpret_cast<const floatx4*>(b1 + hc * HCH + sub * KC + u * 16 + 4 * h);
#pragma unroll
          for (int j = 0; j < TM; ++j) hacc[u][j] = bb;
#pragma unroll
          for (int c = 0; c < NCH; ++c) {
            const frag af = *reinterpret_cast<const frag*>(buf + (sub * KC + u * 16 + r16) * W1ROW + c * 64 + h * 16);
#pragma unroll
            for (int j = 0; j < TM; ++j) hacc[u][j] = Mma<T>::mma(af, xb[c][j], hacc[u][j]);
          }
        }
        frag hb[TM];
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          floatx4 t[NPC];
#pragma unroll
          for (int u = 0; u < NPC; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) t[u][r] = fmaxf(hacc[u][j][r], 0.f);
          hb[j] = Pack<T>::make(t);
        }
        // FFN2 slice: y += W2[:, slice] . relu(hidden)
#pragma unroll
        for (int i = 0; i < ND; ++i) {
          const frag af = *reinterpret_cast<const frag*>(buf + W1BYTES + (i * 16 + r16) * W2ROW + sub * 64 + h * 16);
#pragma unroll
          for (int j = 0; j < TM; ++j) acc[i][j] = Mma<T>::mma(af, hb[j], acc[i][j]);
        }
      }
      if (hc + 1 < nhc) commit(lds + ((hc + 1) & 1) * STAGE);
      __syncthreads();
    }
  }

  // ---- epilogue: token-major store, or fold + window join into the NHWC map ----
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const long long t = tok0 + j * 16 + r16;
    if (t >= a.M) continue;
    if (!a.fmap) {
      T* yp = (T*)a.Y + t * D + 4 * h;
#pragma unroll
      for (int i = 0; i < ND; ++i) store4<T>(yp + i * 16, acc[i][j]);
    } else {
      // token -> (image, window, ty, tx); feature f = (pi*p + pj)*C + c  (token_perm order)
      const int tw = a.ws / a.p, S = tw * tw;
      const int nwx = a.mapW / a.ws, nwy = a.mapH / a.ws;
      const int tt = (int)(t % S);
      const long long wi = t / S;
      const int wx = (int)(wi % nwx), wy = (int)((wi / nwx) % nwy);
      const long long b = wi / ((long long)nwx * nwy);
      const int y0 = wy * a.ws + (tt / tw) * a.p, x0 = wx * a.ws + (tt % tw) * a.p;
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        const int f = i * 16 + 4 * h;
        const int ij = f / a.C, c = f - ij * a.C;
        const int pi = ij / a.p, pj = ij - pi * a.p;
        T* dst = (T*)a.fmap + ((b * a.mapH + y0 + pi) * a.mapW + x0 + pj) * a.cs + c;
        store4<T>(dst, acc[i][j]);
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------------------------------------
// k_mlp2 (fp16): the same chain, rebuilt around what k_mlp's disassembly showed to cost most of its time --
//   * every hidden chunk started with a dependent GLOBAL load of its bias slice in front of the first MFMA;
//   * the compiler issued each weight-fragment ds_read just before the MFMA pair that uses it (one exposed LDS latency per 2 MFMAs);
//   * the weight slices went global -> registers -> ds_write_b128 (13 LDS-path cycles per wave-instruction, 12+ VGPRs).
// Here a chunk of the hidden dimension (W1 slice, W2 slice and the bias slice) lands in LDS by LDS-DMA (global_load_lds_dwordx4:
// no VGPRs, no ds_write), two stages, one raw s_barrier per chunk with the next chunk's DMA in flight across it; the MFMAs run in
// groups of 3 weight fragments x TM token tiles with the NEXT group's fragments (and bias vector) already loading into a second
// register set (software pipeline pinned with sched_barrier), rows padded by 32 bytes (pitch = 32 mod 64: conflict-free
// ds_read_b128, the pad pieces are dead DMA lanes); ReLU runs packed on the fp16 pair (v_pk_max_f16) after the conversion.
template <int I, int N, class F>
CFEN_DEV void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

CFEN_DEV void mlp_dma16(const void* g, unsigned char* l) {
  __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int ND, int TM, int NW, int HCH>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_mlp2(Grouped<MlpArgs> ga) {
  typedef half_t T;
  const MlpArgs a = ga.g[blockIdx.z];
  constexpr int KC = 32, EPL = 8;
  constexpr int NT = NW * 64;
  constexpr int D = ND * 16;
  constexpr int NCH = ND / 2;                     // 32-deep K chunks over D
  constexpr int NSUB = HCH / KC;                  // 32-unit hidden sub-steps per staged chunk
  constexpr int P1 = D * 2 + 32, P2 = HCH * 2 + 32;   // LDS row pitches (bytes), = 32 (mod 64)
  constexpr int PP1 = P1 / 16, PP2 = P2 / 16;     // 16-byte pieces per LDS row (the last two of each row are padding)
  constexpr int N1 = HCH * PP1 / 64, N2 = D * PP2 / 64;   // DMA wave-instructions (64 pieces = 1 KiB) per region
  static_assert(HCH * PP1 % 64 == 0 && D * PP2 % 64 == 0 && HCH * 4 <= 1024, "regions must be whole DMA instructions");
  constexpr int NINS = N1 + N2 + 1;               // + 1 for the bias slice
  constexpr int STAGE = NINS * 1024;
  constexpr int NI = (NINS + NW - 1) / NW;        // DMA instructions per wave per chunk
  constexpr int R2 = N1 * 1024, R3 = (N1 + N2) * 1024;
  static_assert(2 * STAGE <= 65536, "static LDS");
  static_assert(NCH % 3 == 0 && ND % 3 == 0, "fragment groups of 3");
  typedef half8 frag;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, h = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long tok0 = ((long long)blockIdx.x * NW + wave) * (TM * 16);
  const int nhc = a.H / HCH;

  // ---- DMA plan: instruction i of this wave fills LDS block blk = i * NW + wave of the stage; this lane's piece of it ----
  // region of a block (wave-uniform): 0 = W1 slice, 1 = W2 slice, 2 = bias slice; off[i] = byte offset of the lane's source piece
  // inside the region's matrix for chunk 0 (pad pieces re-read a real piece: they are never read back from LDS)
  unsigned off[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int blk = i * NW + wave;
    const int pc = lane;                            // piece inside the block
    if (blk < N1) {
      const int id = blk * 64 + pc, row = id / PP1, col = min(id % PP1, PP1 - 3);
      off[i] = (unsigned)(row * D * 2 + col * 16);                       // W1[hc*HCH + row][col*8 ..]
    } else if (blk < N1 + N2) {
      const int id = (blk - N1) * 64 + pc, row = id / PP2, col = min(id % PP2, PP2 - 3);
      off[i] = (unsigned)row * (unsigned)(a.H * 2) + (unsigned)(col * 16);   // W2[row][hc*HCH + col*8 ..]
    } else {
      off[i] = (unsigned)(min(pc, HCH * 4 / 16 - 1) * 16);               // b1[hc*HCH + 4*pc ..]
    }
  }
  // Flattened chunk sequence T: [projection phase: D/32 chunks of 32 rows of Wp (W1-shaped region only)] [stage a: nhc chunks]
  // [stage b: nhc chunks]; LDS stage = T & 1.
  constexpr int NPC = D / 32;                       // projection chunks
  constexpr int N1P = 32 * PP1 / 64;                // DMA instructions of one projection chunk
  static_assert(32 * PP1 % 64 == 0 && N1P <= N1, "projection chunk must be whole DMA instructions inside the W1 region");
  const int npc = a.Wp ? NPC : 0;
  const int nchunks = npc + (a.W1b ? 2 * nhc : nhc);
  auto issue = [&](int T, int buf) {
    if (T < npc) {
      const unsigned char* Wp = (const unsigned char*)a.Wp + (size_t)T * 32 * D * 2;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int blk = i * NW + wave;
        if (blk < N1P) mlp_dma16(Wp + off[i], lds + buf * STAGE + blk * 1024);
      }
      return;
    }
    const int t = T - npc;
    const bool sb = t >= nhc;
    const int hc = sb ? t - nhc : t;
    const unsigned char* W1 = (const unsigned char*)(sb ? a.W1b : a.W1a) + (size_t)hc * HCH * D * 2;
    const unsigned char* W2 = (const unsigned char*)(sb ? a.W2b : a.W2a) + (size_t)hc * HCH * 2;
    const unsigned char* B1 = (const unsigned char*)(sb ? a.b1b : a.b1a) + (size_t)hc * HCH * 4;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int blk = i * NW + wave;
      if (blk < NINS) {
        const unsigned char* base = blk < N1 ? W1 : blk < N1 + N2 ? W2 : B1;
        mlp_dma16(base + off[i], lds + buf * STAGE + blk * 1024);
      }
    }
  };
  // chunk T has landed and is visible to every wave; the other stage is free: start filling it with chunk T + 1
  auto begin_chunk = [&](int T) -> const unsigned char* {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (T + 1 < nchunks) issue(T + 1, (T + 1) & 1);
    return lds + (T & 1) * STAGE;
  };
  issue(0, 0);

  // ---- load x^T into accumulator layout: acc[i][j][r] = x[token j*16+r16][feature i*16+4h+r] ----
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
  const int a1 = r16 * P1 + h * 16;            // lane part of a W1-region fragment address
  frag xb[NCH][TM];
  if (a.Wp) {   // x += Wp att (out_proj + residual of the attention block, v3:1386): Wp streams through the same DMA ring, 32 rows a chunk;
                // the attention output is this phase's B operand and borrows xb (natural k order on both sides)
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      long long t = tok0 + j * 16 + r16;
      if (t >= a.M) t = a.M - 1;
      const T* ap = (const T*)a.A + t * D + h * EPL;
#pragma unroll
      for (int c = 0; c < NCH; ++c) xb[c][j] = load_frag<T>(ap + c * KC);
    }
    constexpr int GP = 2 * (NCH / 3);   // fragment groups per projection chunk: (row tile u, chunk triple cg)
    static_for<0, NPC>([&](auto tpc) {
      constexpr int tp = decltype(tpc)::value;
      const unsigned char* buf = begin_chunk(tp);
      frag F[2][3];
      auto load_p = [&](auto gc, frag (&f)[3]) {
        constexpr int g = decltype(gc)::value, u = g / (NCH / 3), cg = g % (NCH / 3);
#pragma unroll
        for (int k = 0; k < 3; ++k) f[k] = *reinterpret_cast<const frag*>(buf + a1 + (u * 16) * P1 + (cg * 3 + k) * 64);
      };
      load_p(std::integral_constant<int, 0>{}, F[0]);
      static_for<0, GP>([&](auto gc) {
        constexpr int g = decltype(gc)::value, u = g / (NCH / 3), cg = g % (NCH / 3);
        if constexpr (g + 1 < GP) load_p(std::integral_constant<int, g + 1>{}, F[(g + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
          for (int j = 0; j < TM; ++j) acc[2 * tp + u][j] = Mma<T>::mma(F[g & 1][k], xb[cg * 3 + k][j], acc[2 * tp + u][j]);
        __builtin_amdgcn_sched_barrier(0);
      });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    });
  }

  // stage-a input: LayerNorm(x) (or x) as B fragments; residual + output bias go into the accumulators
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
        xb[c][j] = Pack<T>::make(t);
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        floatx4 t[2] = {acc[c * 2][j], acc[c * 2 + 1][j]};
        xb[c][j] = Pack<T>::make(t);
      }
  }
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    const floatx4 bb = *reinterpret_cast<const floatx4*>(a.b2a + i * 16 + 4 * h);
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] += bb;
  }

  // fragment groups of one 32-unit sub-step: 2 * NCH/3 FFN1 groups (u, chunk triple), then ND/3 FFN2 groups (feature-tile triple)
  constexpr int G1 = 2 * (NCH / 3), G2 = ND / 3, GS = G1 + G2, NG = NSUB * GS;
  const int a2 = R2 + r16 * P2 + h * 16;       // ... of a W2 fragment address
  const int a3 = R3 + 16 * h;                  // ... of a bias vector address

  auto chunk = [&](int t) {
    const unsigned char* buf = begin_chunk(npc + t);

    frag F[2][3];
    floatx4 bia[2];
    floatx4 hacc[2][TM];
    frag hb[TM];
    auto load_group = [&](auto gc, frag (&f)[3], floatx4& bv) {
      constexpr int g = decltype(gc)::value, sub = g / GS, gi = g % GS;
      if constexpr (gi < G1) {
        constexpr int u = gi / (NCH / 3), cg = gi % (NCH / 3);
#pragma unroll
        for (int k = 0; k < 3; ++k) f[k] = *reinterpret_cast<const frag*>(buf + a1 + (sub * KC + u * 16) * P1 + (cg * 3 + k) * 64);
        if constexpr (cg == 0) bv = *reinterpret_cast<const floatx4*>(buf + a3 + (sub * KC + u * 16) * 4);
      } else {
        constexpr int ig = gi - G1;
#pragma unroll
        for (int k = 0; k < 3; ++k) f[k] = *reinterpret_cast<const frag*>(buf + a2 + ((ig * 3 + k) * 16) * P2 + sub * 64);
      }
    };
    auto run_group = [&](auto gc, const frag (&f)[3], const floatx4& bv) {
      constexpr int g = decltype(gc)::value, gi = g % GS;
      if constexpr (gi < G1) {
        constexpr int u = gi / (NCH / 3), cg = gi % (NCH / 3);
        if constexpr (cg == 0) {
#pragma unroll
          for (int j = 0; j < TM; ++j) hacc[u][j] = bv;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
          for (int j = 0; j < TM; ++j) hacc[u][j] = Mma<T>::mma(f[k], xb[cg * 3 + k][j], hacc[u][j]);
      } else {
        constexpr int ig = gi - G1;
        if constexpr (ig == 0) {   // relu(hidden) -> B fragments: convert, then a packed max on the fp16 pairs
#pragma unroll
          for (int j = 0; j < TM; ++j) {
            floatx4 tt[2] = {hacc[0][j], hacc[1][j]};
            const half8 v = Pack<T>::make(tt);
            half8 z;
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = (half_t)0;
            hb[j] = __builtin_elementwise_max(v, z);
          }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
          for (int j = 0; j < TM; ++j) acc[ig * 3 + k][j] = Mma<T>::mma(f[k], hb[j], acc[ig * 3 + k][j]);
      }
    };
    load_group(std::integral_constant<int, 0>{}, F[0], bia[0]);
    static_for<0, NG>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      if constexpr (g + 1 < NG) load_group(std::integral_constant<int, g + 1>{}, F[(g + 1) & 1], bia[(g + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      run_group(gc, F[g & 1], bia[g & 1]);
      __builtin_amdgcn_sched_barrier(0);
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of the stage are complete before it reaches the next barrier
  };
#pragma unroll 1
  for (int t = 0; t < nhc; ++t) chunk(t);
  if (a.W1b) {   // stage b: its input is the stage-a result, which becomes the new residual
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        floatx4 tt[2] = {acc[c * 2][j], acc[c * 2 + 1][j]};
        xb[c][j] = Pack<T>::make(tt);
      }
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const floatx4 bb = *reinterpret_cast<const floatx4*>(a.b2b + i * 16 + 4 * h);
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] += bb;
    }
#pragma unroll 1
    for (int t = nhc; t < 2 * nhc; ++t) chunk(t);
  }

  // ---- epilogue: token-major store, or fold + window join into the NHWC map ----
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const long long t = tok0 + j * 16 + r16;
    if (t >= a.M) continue;
    if (!a.fmap) {
      T* yp = (T*)a.Y + t * D + 4 * h;
#pragma unroll
      for (int i = 0; i < ND; ++i) store4<T>(yp + i * 16, acc[i][j]);
    } else {
      const int tw = a.ws / a.p, S = tw * tw;
      const int nwx = a.mapW / a.ws, nwy = a.mapH / a.ws;
      const int tt = (int)(t % S);
      const long long wi = t / S;
      const int wx = (int)(wi % nwx), wy = (int)((wi / nwx) % nwy);
      const long long b = wi / ((long long)nwx * nwy);
      const int y0 = wy * a.ws + (tt / tw) * a.p, x0 = wx * a.ws + (tt % tw) * a.p;
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        const int f = i * 16 + 4 * h;
        const int ij = f / a.C, c = f - ij * a.C;
        const int pi = ij / a.p, pj = ij - pi * a.p;
        T* dst = (T*)a.fmap + ((b * a.mapH + y0 + pi) * a.mapW + x0 + pj) * a.cs + c;
        store4<T>(dst, acc[i][j]);
      }
    }
  }
}

template <int ND, int TM, int NW, int HCH>
int launch_mlp2_t(int ng, const MlpArgs* ap, hipStream_t s) {
  const MlpArgs& a = ap[0];
  Grouped<MlpArgs> ga;
  for (int g = 0; g < CFEN_MAX_GROUPS; ++g) ga.g[g] = ap[g < ng ? g : 0];
  CFEN_CHECK_ARG(a.H % HCH == 0, "mlp: hidden dim %d must be a multiple of %d", a.H, HCH);
  const long long per = (long long)NW * TM * 16;
  const long long blocks = (a.M + per - 1) / per;
  CFEN_CHECK_ARG(blocks > 0 && blocks < (1ll << 31), "mlp: bad grid");
  CFEN_LAUNCH((k_mlp2<ND, TM, NW, HCH>), dim3((unsigned)blocks, 1, ng), dim3(NW * 64), 0, s, ga);
  CFEN_CHECK_LAUNCH("mlp");
  return CFEN_OK;
}

template <typename T, int ND, int TM, int NW, int HCH, int WPE>
int launch_mlp_t(int ng, const MlpArgs* ap, hipStream_t s) {
  const MlpArgs& a = ap[0];
  Grouped<MlpArgs> ga;
  for (int g = 0; g < CFEN_MAX_GROUPS; ++g) ga.g[g] = ap[g < ng ? g : 0];
  constexpr int SZ = (int)sizeof(T), D = ND * 16;
  constexpr size_t smem = 2 * (size_t)(HCH * (D * SZ + 16) + D * (HCH * SZ + 16));
  CFEN_CHECK_ARG(a.H % HCH == 0, "mlp: hidden dim %d must be a multiple of %d", a.H, HCH);
  const long long per = (long long)NW * TM * 16;
  const long long blocks = (a.M + per - 1) / per;
  CFEN_CHECK_ARG(blocks > 0 && blocks < (1ll << 31), "mlp: bad grid");
  static bool attr_set[64] = {};
  if (cfen_first_use_on_device(attr_set)) {
    if (hipFuncSetAttribute((const void*)k_mlp<T, ND, TM, NW, HCH, WPE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {
      cfen_set_error("mlp: cannot reserve %zu bytes of LDS", smem);
      return CFEN_ERR_HIP;
    }
  }
  CFEN_LAUNCH((k_mlp<T, ND, TM, NW, HCH, WPE>), dim3((unsigned)blocks, 1, ng), dim3(NW * 64), smem, s, ga);
  CFEN_CHECK_LAUNCH("mlp");
  return CFEN_OK;
}

template <typename T>
int check_mlp(const MlpArgs& a) {
  constexpr int KC = Mma<T>::KC;
  CFEN_CHECK_ARG(a.M > 0 && a.H > 0 && a.H % (2 * KC) == 0, "mlp: hidden dim %d must be a positive multiple of %d", a.H, 2 * KC);
  CFEN_CHECK_ARG(a.X && a.W1a && a.W2a && a.b1a && a.b2a, "mlp: null pointer");
  CFEN_CHECK_ARG((a.W1b == nullptr) == (a.W2b == nullptr) && (!a.W1b || (a.b1b && a.b2b)), "mlp: incomplete second stage");
  CFEN_CHECK_ARG((a.ln_g == nullptr) == (a.ln_b == nullptr), "mlp: LayerNorm needs gamma and beta");
  CFEN_CHECK_ARG(a.Y || a.fmap, "mlp: no output");
  CFEN_CHECK_ARG((a.A == nullptr) == (a.Wp == nullptr) && cfen_aligned16(a.A) && cfen_aligned16(a.Wp), "mlp: projection prologue needs A and Wp");
  if (a.fmap) {
    CFEN_CHECK_ARG(a.C > 0 && a.C % 4 == 0 && a.cs >= a.C && a.cs % 4 == 0 && a.p > 0 && a.ws % a.p == 0 && a.mapH % a.ws == 0 &&
                   a.mapW % a.ws == 0 && a.p * a.p * a.C == a.D, "mlp: bad fold geometry");
    CFEN_CHECK_ARG(a.M % ((long long)(a.ws / a.p) * (a.ws / a.p)) == 0, "mlp: token count does not tile the map");
  }
  CFEN_CHECK_ARG(cfen_aligned16(a.X) && cfen_aligned16(a.Y) && cfen_aligned16(a.fmap) && cfen_aligned16(a.W1a) && cfen_aligned16(a.W2a) &&
                 cfen_aligned16(a.W1b) && cfen_aligned16(a.W2b) && cfen_aligned16(a.b1a) && cfen_aligned16(a.b2a) && cfen_aligned16(a.b1b) &&
                 cfen_aligned16(a.b2b) && cfen_aligned16(a.ln_g) && cfen_aligned16(a.ln_b), "mlp: pointers must be 16-byte aligned");
  CFEN_CHECK_ARG(a.D == 96 || a.D == 192, "mlp: fused kernel supports D in {96,192}, got %d", a.D);
  return CFEN_OK;
}

template <typename T>
int launch_mlp(int ng, const MlpArgs* ap, hipStream_t s) {
  constexpr int KC = Mma<T>::KC;
  CFEN_CHECK_ARG(ng >= 1 && ng <= CFEN_MAX_GROUPS && ap, "mlp: 1..%d problems per launch", CFEN_MAX_GROUPS);
  for (int g = 0; g < ng; ++g) {
    int rc = check_mlp<T>(ap[g]);
    if (rc) return rc;
    CFEN_CHECK_ARG(ap[g].M == ap[0].M && ap[g].D == ap[0].D && ap[g].H == ap[0].H && (ap[g].W1b == nullptr) == (ap[0].W1b == nullptr) &&
                   (ap[g].ln_g == nullptr) == (ap[0].ln_g == nullptr) && (ap[g].Wp == nullptr) == (ap[0].Wp == nullptr),
                   "mlp: grouped problems must have the same shape");
  }
  const int small = cfen_tune_mlp_small_tiles();
  if constexpr (sizeof(T) == 2) {
    // small >= 10: the LDS-DMA / software-pipelined kernel (k_mlp2).  Tens digit = D 96 variant, ones digit = D 192 variant:
    //   D = 96 : 1x 256 tokens / 4-wave WG (TM 4; default), 2x 512 tokens / 8-wave WG (TM 4), 3x 256 tokens / 8-wave WG (TM 2), 4x as 3x with 32-unit stages
    //   D = 192: x0 128 tokens / 4-wave WG (TM 2; default), x1 256 tokens / 8-wave WG (TM 2)
    // (MI355X, batch 8, cold caches, tools/bench_mlp_variants.py: 4-wave workgroups win -- two of them share a CU and their barriers interleave)
    if (small >= 10) {
      const int v96 = small / 10, v192 = small % 10;
      if (ap[0].D == 96) return v96 == 2 ? launch_mlp2_t<6, 4, 8, 64>(ng, ap, s) : v96 == 3 ? launch_mlp2_t<6, 2, 8, 64>(ng, ap, s)
                              : v96 == 4 ? launch_mlp2_t<6, 2, 8, 32>(ng, ap, s) : launch_mlp2_t<6, 4, 4, 64>(ng, ap, s);
      // x2: 192 tokens / 6-wave WG -- the grouped decoder launch (3 x 32768 tokens) is then exactly one round of 2 WGs per CU
      if (v192 == 2 || (v192 == 3 && (long long)ng * ap[0].M >= 3 * 32768)) return launch_mlp2_t<12, 2, 6, 32>(ng, ap, s);
      return v192 == 1 ? launch_mlp2_t<12, 2, 8, 32>(ng, ap, s) : launch_mlp2_t<12, 2, 4, 32>(ng, ap, s);
    }
  }
  switch (ap[0].D) {
    case 96: return small >= 3 ? launch_mlp_t<T, 6, 2, 8, 2 * KC, 2>(ng, ap, s)   // 256 tokens / WG in 8 waves: half the weight re-streaming
                  : small ? launch_mlp_t<T, 6, 2, 4, 2 * KC, 2>(ng, ap, s)     // 128 tokens / WG, 2 waves per SIMD
                          : launch_mlp_t<T, 6, 4, 4, 2 * KC, 1>(ng, ap, s);    // 256 tokens / WG, 24 KB stages
    default: return small == 4 ? launch_mlp_t<T, 12, 1, 8, KC, 2>(ng, ap, s)   // 128 tokens / WG in 8 waves, no spills
                  : small == 3 ? launch_mlp_t<T, 12, 2, 8, KC, 2>(ng, ap, s)   // 256 tokens / WG in 8 waves
                  : small == 2 ? launch_mlp_t<T, 12, 2, 4, KC, 2>(ng, ap, s)   // 128 tokens / WG, registers capped for 2 waves per SIMD
                  : small ? launch_mlp_t<T, 12, 1, 4, KC, 2>(ng, ap, s)        // 64 tokens / WG, 2 waves per SIMD
                          : launch_mlp_t<T, 12, 2, 4, KC, 1>(ng, ap, s);       // 128 tokens / WG, 24 KB stages
  }
}

}  // namespace

// D = 384 (LViT level 3, 8192 tokens per batch of 8) does not fill the chip with 128-token workgroups and its
// 4.7 MB of weights per instance exceed what one CU can stream per token tile: the tiled GEMM path is faster there.
int& cfen_tune_mlp_small_tiles() {
  static int v = 10;
  return v;
}

bool cfen_mlp_supported(int D, int H, int dtype) { return (D == 96 || D == 192) && H % (dtype == 1 ? 64 : 32) == 0; }

int cfen_mlp_impl_g(int dtype, int ng, const MlpArgs* a, hipStream_t s) {
  if (dtype == 1) return launch_mlp<half_t>(ng, a, s);
  if (dtype == 0) return launch_mlp<float>(ng, a, s);
  cfen_set_error("mlp: unknown dtype %d", dtype);
  return CFEN_ERR_ARG;
}
int cfen_mlp_impl(int dtype, const MlpArgs* a, hipStream_t s) { return cfen_mlp_impl_g(dtype, 1, a, s); }
