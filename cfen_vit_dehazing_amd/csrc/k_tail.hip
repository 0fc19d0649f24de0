// k_tail_fused: the whole output tail of a decoder in ONE launch (R, S and D grouped) --
//   us_conv_d01* = ConvTranspose2d(4, 2, 1) 24 -> 12 + ActNorm2d + ReLU                  (models/networks_iid_hlgvit_crs_gd4_cfs_v3.py:318-322)
//   tail_*       = Conv2d 3x3 12 -> 12 (+ ActNorm2d for R, D) + ReLU, ReflectionPad2d(3), Conv2d 7x7 12 -> 3 / 1 / 3, Tanh   (v3:348-383)
// Rounds 3-4 ran this as k_up_conv3_fused (ConvTranspose + 3x3, 127 us) and k_conv7_tz (92 us): the 12-channel full-resolution map between them was
// written (201 MB) and read back with a 22 x 70 halo per 16 x 64 tile.  Both are wide launches, and with several forwards in flight a wide launch
// costs its whole duration (profiles/r05_probe_skip_launches.txt: 0.12 + 0.065 ms of a 2.08 ms step).
//
// Here a workgroup owns a strip of 64 output columns and walks DOWN it in bands of 8 rows, so no stage is recomputed vertically and only the
// strip's 72 / 70-column halo horizontally.  Per band j (three stages, each the loop of the kernel it replaces, same accumulation order per output,
// intermediates rounded to fp16 as the stored maps were -> bitwise the unfused results):
//   A  ConvTranspose rows 8j+4 .. 8j+11, columns X0-4 .. X0+71   (k_up_conv3_fused's stage: one parity phase per wave) from 6 staged base rows -> t1
//   B  3x3 rows 8j+3 .. 8j+10, columns X0-3 .. X0+68 from t1 (its last two rows of band j-1 are carried over)           -> t2, a ring of 16 rows
//   C  7x7 + tanh rows 8j .. 8j+7 from t2 rows 8j-3 .. 8j+10 (reflected at the image border by row / by mirrored stores of stage B) -> output
// Three wave groups work on three different bands at once, one barrier per band: in step j waves 0..3 run A of band j, waves 4..7 run B of band j-1, waves
// 8..11 run C of band j-2 (their 35 Toeplitz fragments are 140 registers a lane: the reason for the roles -- a 4-wave version that did everything spilled 157
// registers), and the eight waves 0..7 finish band j-3 (scale, tanh, store: one row each) from the accumulators the C waves parked in LDS.  Every SIMD holds
// one wave of each group, so one group's MFMAs run under another's epilogue arithmetic.  LDS (one workgroup a CU): staged input 2 x 6 x 50 x 64 B, t1 a ring
// of 20 rows x 84 x 32 B, t2 a ring of 22 rows x 72 x 32 B, parked accumulators 2 x 8 KB = 155.5 KB.
#include "cfen_common.hpp"
#include "cfen_internal.hpp"
#include "cfen_conv.hpp"

namespace {

constexpr int TF_INROWS = 6, TF_INW = 50, TF_INS = 40, TF_INB = TF_INW * 64, TF_INBUF = TF_INROWS * TF_INB;   // staged base rows 4j+1 .. 4j+6, base columns xb0-3 .. (40 staged, 50 addressed)
constexpr int TF_T1W = 84, TF_T1B = TF_T1W * 32, TF_T1R = 20, TF_T1KEEP = 76;  // t1 columns X0-4 .., kept while < 76 (stage B's zero-weight fourth tap must read finite values)
constexpr int TF_ZW = 72, TF_ZB = TF_ZW * 32, TF_ZNCH = 5, TF_ZKPAD = 7 * TF_ZNCH * 32, TF_T2R = 22;
constexpr int TF_ZBUF = 8 * 4 * 16 * 16;                                       // the 7x7's raw accumulators of a band: [row 8][channel 4][pixel quad 16] fp32 x 4
constexpr int TF_LDS = 2 * TF_INBUF + TF_T1R * TF_T1B + TF_T2R * TF_ZB + 2 * TF_ZBUF;
static_assert(TF_LDS <= 160 * 1024, "one workgroup a CU");

struct TailArgs {
  const half_t* in; const half_t* wT; const float* sT; const float* tT; int actT;
  const half_t* w3; const float* s3; const float* t3; int act3;
  const half_t* w7; const float* s7; const float* t7; int act7; int Cout; void* out; int out_mode;   // out_mode 1: fp32 NCHW, 2: uint8 HWC x 3 (util.tensor2im), 3: fp16 NCHW (round 6: the sharded run's wire type, written directly)
  int B, Hin, Win, cs_in;
};

CFEN_DEV int tf_swz(int col) { return ((col >> 2) & 1) << 1; }            // k_fuse.hip uf_swz<64>
CFEN_DEV int tf_t1col(int col) { return col ^ ((col >> 2) & 1); }         // k_fuse.hip uf_t1col
CFEN_DEV int tf_zpiece(int P, int q) { const int slot = 2 * P + q; return (slot & ~7) | ((slot & 7) ^ ((P >> 2) & 6)); }   // k_conv7_tz's halo swizzle
CFEN_DEV int tf_t1slot(int y) { return (y + 8 * TF_T1R) % TF_T1R; }       // y >= -160
CFEN_DEV int tf_t2slot(int y) { return (y + 8 * TF_T2R) % TF_T2R; }

__global__ __launch_bounds__(768) void k_tail_fused(Grouped<TailArgs> ga, int nblk, int segb, int dbg, int bal, unsigned long long* stamps) {
  const TailArgs& a = ga.g[blockIdx.z];
  typedef half_t T;
  typedef half8 frag;
  extern __shared__ __attribute__((aligned(16))) unsigned char tf_lds[];
  unsigned char* const inl = tf_lds;
  unsigned char* const t1 = tf_lds + 2 * TF_INBUF;
  unsigned char* const t2 = t1 + TF_T1R * TF_T1B;
  unsigned char* const zbuf = t2 + TF_T2R * TF_ZB;

  const int blk = (int)xcd_chunked_block(blockIdx.x, gridDim.x);
  if (blk >= nblk) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, h = lane >> 4, n = 4 * h;
  const int Hf = 2 * a.Hin, Wf = 2 * a.Win;
  const int tiles_x = Wf / 64, nseg = (Hf / 8) / segb;
  const int tx = blk % tiles_x, seg = (blk / tiles_x) % nseg, b = blk / (tiles_x * nseg);
  const int X0 = tx * 64, xb0 = tx * 32, k0 = seg * segb, k1 = k0 + segb;
  // scalars of the argument block the band loop uses: read once (behind a barrier the compiler re-reads them from the kernel arguments, an s_load + lgkmcnt(0) per use)
  const int actT = a.actT, act3 = a.act3, act7 = a.act7, Cout = a.Cout, out_mode = a.out_mode, Hin = a.Hin, Win = a.Win;
  void* const outp = a.out;
  // in-kernel stamps (tail.debug & 64): workgroup 0 of problem 0, lane 0 of the first wave of each group: [step][group][after the barrier, own work done]
  const bool stamper = stamps && blk == 0 && blockIdx.z == 0 && lane == 0 && (wave == 0 || wave == 4 || wave == 8);
  auto stamp = [&](int step, int which) { if (stamper && step < 96) stamps[(step * 3 + (wave >> 2)) * 2 + which] = __builtin_amdgcn_s_memrealtime(); };
  auto band_live = [&](int j) { return j >= k0 - 2 && j < k1 && 8 * j + 11 >= 0 && 8 * j + 4 < Hf; };   // ConvTranspose rows 8j+4 .. 8j+11 are wanted and meet the image
  const int jend = k1 + 2;                             // the last step finishes band k1 - 1

  if (wave < 8) {
    // ---- staging (waves 0..7): the 6 rows x 40 columns x 4 pieces of 16 bytes of a band are dealt over 512 threads, two pieces each (all of it on the
    //      ConvTranspose waves, four pieces each: slower, 277 against 265 us) ----
    const int src_pixb = a.cs_in * 2, src_pieces = src_pixb / 16;
    const unsigned char* src = (const unsigned char*)a.in + (size_t)b * a.Hin * a.Win * src_pixb;
    int s_row[2], s_off[2], s_dst[2];
    bool s_use[2], s_ok[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int e = tid + 512 * k, row = e / (TF_INS * 4), rem = e % (TF_INS * 4), scol = rem >> 2, spiece = rem & 3, sgx = xb0 - 3 + scol;
      s_use[k] = row < TF_INROWS;
      s_row[k] = row;
      s_ok[k] = s_use[k] && spiece < src_pieces && sgx >= 0 && sgx < a.Win;
      s_off[k] = min(max(sgx, 0), a.Win - 1) * src_pixb + min(spiece, src_pieces - 1) * 16;
      s_dst[k] = (s_use[k] ? row : 0) * TF_INB + scol * 64 + ((spiece ^ tf_swz(scol)) << 4);
    }
    frag stg[2];
    auto fetch = [&](int j) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int gy = 4 * j + 1 + s_row[k];
        const frag v = *reinterpret_cast<const frag*>(src + (size_t)min(max(gy, 0), Hin - 1) * Win * src_pixb + s_off[k]);
        stg[k] = (s_ok[k] && gy >= 0 && gy < Hin) ? v : Mma<T>::zero();
      }
    };
    auto stage = [&](int j) {                          // band j's input into buffer j & 1
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (s_use[k]) *reinterpret_cast<frag*>(inl + (j & 1) * TF_INBUF + s_dst[k]) = stg[k];
    };
    const float sc7 = a.s7[h], sh7 = a.t7[h];            // (h = 3 may read a padding entry of the [16] table; its results are never stored)
    // finish band e (waves 0..7 = its rows): scale, tanh, store -- k_conv7_tz's epilogue on the accumulators the C waves parked
    auto finish = [&](int e, int row) {
      const int y = 8 * e + row;
      floatx4 v = *reinterpret_cast<const floatx4*>(zbuf + (e & 1) * TF_ZBUF + ((row * 4 + h) * 16 + r16) * 16) * sc7 + sh7;
      if (act7 == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
      } else if (act7 == 2 && !(dbg & 8)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = tanhf(v[q]);
      }
      if (out_mode == 2) {
        unsigned wr = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) wr |= (unsigned)(unsigned char)(int)((v[q] + 1.f) / 2.0f * 255.0f) << (8 * q);
        const unsigned wg = Cout >= 3 ? (unsigned)__shfl((int)wr, r16 + 16, 64) : wr;     // all 64 lanes take part in the exchange
        const unsigned wbl = Cout >= 3 ? (unsigned)__shfl((int)wr, r16 + 32, 64) : wr;
        if (h == 0) {
          // bytes R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
          const unsigned o0 = (wr & 0xffu) | ((wg & 0xffu) << 8) | ((wbl & 0xffu) << 16) | ((wr & 0xff00u) << 16);
          const unsigned o1 = ((wg >> 8) & 0xffu) | (((wbl >> 8) & 0xffu) << 8) | (((wr >> 16) & 0xffu) << 16) | (((wg >> 16) & 0xffu) << 24);
          const unsigned o2 = ((wbl >> 16) & 0xffu) | (((wr >> 24) & 0xffu) << 8) | (((wg >> 24) & 0xffu) << 16) | (((wbl >> 24) & 0xffu) << 24);
          unsigned* op = reinterpret_cast<unsigned*>((unsigned char*)outp + (((size_t)b * Hf + y) * Wf + X0 + 4 * r16) * 3);
          op[0] = o0; op[1] = o1; op[2] = o2;
        }
      } else if (h < Cout) {
        const size_t o = (((size_t)b * Cout + h) * Hf + y) * Wf + X0 + 4 * r16;
        if (out_mode == 3) {
          const half4 hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
          *reinterpret_cast<half4*>((half_t*)outp + o) = hv;
        } else {
          *reinterpret_cast<floatx4*>((float*)outp + o) = v;
        }
      }
    };
    if (band_live(k0 - 2)) { fetch(k0 - 2); stage(k0 - 2); }

    if (wave < 4) {
      // ============================ waves 0..3: ConvTranspose (A) of band j, one parity phase per wave ============================
      const int py = wave >> 1, px = wave & 1;
      frag wf[4];
      {
        const T* wp = a.wT + ((size_t)wave * 16 + r16) * 128 + h * 8;
#pragma unroll
        for (int t = 0; t < 4; ++t) wf[t] = load_frag<T>(wp + t * 32);
      }
      const floatx4 scT = *reinterpret_cast<const floatx4*>(a.sT + n), shT = *reinterpret_cast<const floatx4*>(a.tT + n);
      // LDS offset of the (tap column, column tile) fragment; epilogue: t1 column of (xq, lane) = 32 xq + 2 r16 + px = X - (X0 - 4)
      int a_ld[2][3], a_pix[3];
      bool a_keep[3];
      floatx4 a_sc[3], a_sh[3];
#pragma unroll
      for (int xq = 0; xq < 3; ++xq) {
#pragma unroll
        for (int txx = 0; txx < 2; ++txx) {
          const int col = xq * 16 + r16 + px + 1 - txx;
          a_ld[txx][xq] = col * 64 + ((h ^ tf_swz(col)) << 4);
        }
        const int idx = 32 * xq + 2 * r16 + px, X = X0 - 4 + idx;
        a_keep[xq] = idx < TF_T1KEEP;
        a_pix[xq] = tf_t1col(idx) * 32 + n * 2;
        const bool ximg = X >= 0 && X < Wf;            // a column outside the image is the 3x3's zero padding: exact zeros through a zero scale and shift
        a_sc[xq] = ximg ? scT : floatx4{0.f, 0.f, 0.f, 0.f};
        a_sh[xq] = ximg ? shT : floatx4{0.f, 0.f, 0.f, 0.f};
      }
      for (int j = k0 - 2; j <= jend; ++j) {
        stamp(j - (k0 - 2), 1);
        __syncthreads();
        stamp(j - (k0 - 2) + 1, 0);
        const bool a_live = band_live(j);              // wave-uniform
        if (band_live(j + 1) && !(dbg & 32)) fetch(j + 1);
        if (j < k1) {
          floatx4 acc[4][3];
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int xq = 0; xq < 3; ++xq) acc[r][xq] = floatx4{0.f, 0.f, 0.f, 0.f};
          if (a_live && !(dbg & 1)) {
            const unsigned char* lp = inl + (j & 1) * TF_INBUF + (1 + py) * TF_INB;
#pragma unroll
            for (int q = -1; q < 4; ++q) {
#pragma unroll
              for (int txx = 0; txx < 2; ++txx) {
#pragma unroll
                for (int xq = 0; xq < 3; ++xq) {
                  const frag bf = *reinterpret_cast<const frag*>(lp + q * TF_INB + a_ld[txx][xq]);
                  if (q >= 0) acc[q][xq] = Mma<T>::mma(wf[txx], bf, acc[q][xq]);
                  if (q + 1 < 4) acc[q + 1][xq] = Mma<T>::mma(wf[2 + txx], bf, acc[q + 1][xq]);
                }
              }
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int Y = 8 * j + 4 + 2 * r + py;      // wave-uniform
            const bool yimg = a_live && Y >= 0 && Y < Hf;
            unsigned char* const row = t1 + tf_t1slot(Y) * TF_T1B;
#pragma unroll
            for (int xq = 0; xq < 3; ++xq) {
              floatx4 v = floatx4{0.f, 0.f, 0.f, 0.f};
              if (yimg) {
                v = acc[r][xq] * a_sc[xq] + a_sh[xq];
                if (actT == 1) {
#pragma unroll
                  for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
              }
              const half4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};   // (ReLU on the packed halves -- v_pk_max_f16 -- is NOT bitwise the fp32 one: tested)
              if (a_keep[xq]) *reinterpret_cast<half4*>(row + a_pix[xq]) = o;
            }
          }
        }
        if (j - 3 >= k0 && j - 3 < k1 && !(dbg & 16)) {
          if (bal & 1) { finish(j - 3, 2 * wave); finish(j - 3, 2 * wave + 1); }   // all eight rows on the ConvTranspose waves (the 3x3 waves are the long pole)
          else finish(j - 3, wave);
        }
        if (band_live(j + 1)) stage(j + 1);
      }
    } else {
      // ============================ waves 4..7: 3x3 (B) of band j-1: rows 2w, 2w+1 of the band, five 16-column tiles ============================
      const int w = wave - 4;
      frag wb[3][2];
      {
        const T* pb = a.w3 + (size_t)r16 * (3 * 2 * 32) + h * 8;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int c = 0; c < 2; ++c) wb[dy][c] = load_frag<T>(pb + (dy * 2 + c) * 32);
      }
      const floatx4 sc3 = *reinterpret_cast<const floatx4*>(a.s3 + n), sh3 = *reinterpret_cast<const floatx4*>(a.t3 + n);
      // column tile ct of the lane = t2 pixel P = 16 ct + r16 = X - (X0 - 3); the reflected copy of a padding column of the 7x7 is stored by its source's owner
      int b_ld[2][5], b_off[5], b_moff[5];
      bool b_keep[5], b_mir[5];
#pragma unroll
      for (int ct = 0; ct < 5; ++ct) {
        const int P = 16 * ct + r16, X = X0 - 3 + P;
#pragma unroll
        for (int c = 0; c < 2; ++c) b_ld[c][ct] = tf_t1col(16 * ct + r16 + 2 * c + (h >> 1)) * 32 + (h & 1) * 16;
        b_keep[ct] = P < TF_ZW && X >= 0 && X < Wf;
        b_off[ct] = (tf_zpiece(P & 127, h >> 1) << 4) + (h & 1) * 8;
        const int Xm = (X >= 1 && X <= 3) ? -X : (X >= Wf - 4 && X <= Wf - 2) ? 2 * (Wf - 1) - X : -100000;
        const int Pm = Xm - (X0 - 3);
        b_mir[ct] = b_keep[ct] && Pm >= 0 && Pm < TF_ZW;
        b_moff[ct] = (tf_zpiece(Pm & 127, h >> 1) << 4) + (h & 1) * 8;
      }
      for (int j = k0 - 2; j <= jend; ++j) {
        stamp(j - (k0 - 2), 1);
        __syncthreads();
        stamp(j - (k0 - 2) + 1, 0);
        const int bj = j - 1;
        if (band_live(j + 1) && !(dbg & 32)) fetch(j + 1);
        if (bj >= k0 - 1 && bj < k1) {
          floatx4 bc[2][5];
#pragma unroll
          for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int ct = 0; ct < 5; ++ct) bc[r][ct] = floatx4{0.f, 0.f, 0.f, 0.f};
          if (!(dbg & 2))
#pragma unroll
          for (int iy = 0; iy < 4; ++iy) {
            const unsigned char* lp = t1 + tf_t1slot(8 * bj + 2 + 2 * w + iy) * TF_T1B;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
              for (int ct = 0; ct < 5; ++ct) {
                const frag bf = *reinterpret_cast<const frag*>(lp + b_ld[c][ct]);
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                  const int dy = iy - r;
                  if (dy >= 0 && dy < 3) bc[r][ct] = Mma<T>::mma(wb[dy][c], bf, bc[r][ct]);
                }
              }
          }
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const int y = 8 * bj + 3 + 2 * w + r;      // wave-uniform
            if (y < 0 || y >= Hf) continue;            // the 7x7 reads reflected rows instead
            unsigned char* const row = t2 + tf_t2slot(y) * TF_ZB;
#pragma unroll
            for (int ct = 0; ct < 5; ++ct) {
              floatx4 v = bc[r][ct] * sc3 + sh3;
              if (act3 == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
              }
              const half4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
              if (b_keep[ct]) *reinterpret_cast<half4*>(row + b_off[ct]) = o;
              if (b_mir[ct]) *reinterpret_cast<half4*>(row + b_moff[ct]) = o;
            }
          }
        }
        if (j - 3 >= k0 && j - 3 < k1 && !(dbg & 16) && !(bal & 1)) finish(j - 3, wave);
        if (band_live(j + 1)) stage(j + 1);
      }
    }
  } else {
    // ============================ waves 8..11: 7x7 (C) of band j-2: rows 2w, 2w+1, accumulators parked in LDS for the finishing waves ============================
    const int w = wave - 8;
    frag w7[7][TF_ZNCH];
    {
      const T* pz = a.w7 + (size_t)r16 * TF_ZKPAD + h * 8;
#pragma unroll
      for (int dy = 0; dy < 7; ++dy)
#pragma unroll
        for (int c = 0; c < TF_ZNCH; ++c) w7[dy][c] = load_frag<T>(pz + (dy * TF_ZNCH + c) * 32);
    }
    for (int j = k0 - 2; j <= jend; ++j) {
      stamp(j - (k0 - 2), 1);
      __syncthreads();
      stamp(j - (k0 - 2) + 1, 0);
      const int cj = j - 2;
      if (cj < k0 || cj >= k1) continue;               // wave-uniform
      floatx4 zc[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};
      if (!(dbg & 4)) {
#pragma unroll
        for (int iy = 0; iy < 8; ++iy) {
          int yy = 8 * cj + 2 * w + iy - 3;
          yy = yy < 0 ? -yy : (yy >= Hf ? 2 * Hf - 2 - yy : yy);       // ReflectionPad2d(3)
          const unsigned char* lp = t2 + tf_t2slot(yy) * TF_ZB;
#pragma unroll
          for (int c = 0; c < TF_ZNCH; ++c) {
            const int G = r16 + (c >> 1);
            const frag bf = *reinterpret_cast<const frag*>(lp + ((8 * G + ((4 * (c & 1) + h) ^ (G & 6))) << 4));
#pragma unroll
            for (int r = 0; r < 2; ++r) {
              const int dy = iy - r;
              if (dy >= 0 && dy < 7) zc[r] = Mma<T>::mma(w7[dy][c], bf, zc[r]);
            }
          }
        }
      }
      // lane (r16, h): channel h, output pixels X0 + 4 r16 .. + 3 of rows 8 cj + 2 w + r -> zbuf[cj & 1][row][h][r16]
#pragma unroll
      for (int r = 0; r < 2; ++r) *reinterpret_cast<floatx4*>(zbuf + (cj & 1) * TF_ZBUF + (((2 * w + r) * 4 + h) * 16 + r16) * 16) = zc[r];
    }
  }
}

}  // namespace

int& cfen_tune_tail_balance() {   // work split between the wave groups ("tail.balance"): bit 0 = the 7x7's scale / tanh / store of all eight rows on the ConvTranspose waves.
  static int v = 0;              // In-kernel stamps (tools/dbg_tail_stamps.py): per band the ConvTranspose waves are busy 3.05 us, the 3x3 waves 4.2, the 7x7 waves 3.45 -> with bit 0
                                 // 3.3 / 3.8 / 3.45 and the launch 8 % shorter alone (265 against 287 us), but no faster with four forwards in flight (2.090-2.109 against 2.084-2.092 ms)
  return v;
}
int& cfen_tune_tail_debug() {   // timing experiments (results invalid): 1 no ConvTranspose MFMAs, 2 no 3x3 MFMAs, 4 no 7x7 MFMAs, 8 no tanh, 16 no output stores, 32 no input fetch
  static int v = 0;
  return v;
}
int& cfen_tune_tail_segments() {   // vertical segments a strip is cut into ("tail.segments"): more workgroups against 5 fill / drain steps per segment.  Measured, batch 8:
  static int v = 1;                // alone 4 segments are fastest (202 us against 235 for 2); with four forwards in flight fewer are: 2.043-2.048 ms per step on 4, 2.022-2.046 on 2, 2.023-2.035 on 1
  return v;
}

bool cfen_tail_fused_supported(int dtype, int cs_in, int Cup_pad, int cs_up, int C3_pad, int Hin, int Win, int Cout7, int out_mode) {
  return dtype == 1 && cs_in % 8 == 0 && cs_in * 2 <= 64 && Cup_pad == 16 && cs_up == 16 && C3_pad == 16 && Hin % 4 == 0 && Win % 32 == 0 && Hin >= 8 &&
         Cout7 >= 1 && Cout7 <= 4 && (out_mode == 1 || out_mode == 3 || (out_mode == 2 && (Cout7 == 1 || Cout7 == 3)));
}

int cfen_tail_fused_impl_g(int dtype, int ng, const CfenUpConv3* u, const ConvDesc* d7, hipStream_t s) {
  CFEN_CHECK_ARG(ng >= 1 && ng <= CFEN_MAX_GROUPS && u && d7, "tail (fused): 1..%d problems per launch", CFEN_MAX_GROUPS);
  Grouped<TailArgs> ga;
  memset(&ga, 0, sizeof(ga));
  for (int g = 0; g < ng; ++g) {
    const CfenUpConv3& q = u[g];
    const ConvDesc& z = d7[g];
    CFEN_CHECK_ARG(cfen_tail_fused_supported(dtype, q.cs_in, 16, 16, 16, q.Hin, q.Win, z.Cout, z.out_nchw_f32) && q.B > 0, "tail (fused): unsupported geometry");
    CFEN_CHECK_ARG(q.B == u[0].B && q.Hin == u[0].Hin && q.Win == u[0].Win && q.cs_in == u[0].cs_in, "tail (fused): grouped problems must share the geometry");
    CFEN_CHECK_ARG(z.Hin == 2 * q.Hin && z.Win == 2 * q.Win && z.B == q.B && z.pad_reflect && z.Kpad == TF_ZKPAD && z.cs_in == 16, "tail (fused): the 7x7 does not match the map");
    CFEN_CHECK_ARG(q.in && q.wT && q.sT && q.tT && q.w3 && q.s3 && q.t3 && z.weight && z.scale && z.shift && z.out, "tail (fused): null pointer");
    CFEN_CHECK_ARG(cfen_aligned16(q.in) && cfen_aligned16(q.wT) && cfen_aligned16(q.sT) && cfen_aligned16(q.tT) && cfen_aligned16(q.w3) && cfen_aligned16(q.s3) &&
                   cfen_aligned16(q.t3) && cfen_aligned16(z.weight) && cfen_aligned16(z.out), "tail (fused): pointers must be 16-byte aligned");
    CFEN_CHECK_ARG((q.actT == 0 || q.actT == 1) && (q.act3 == 0 || q.act3 == 1) && z.act >= 0 && z.act <= 2, "tail (fused): bad activation");
    ga.g[g] = TailArgs{(const half_t*)q.in, (const half_t*)q.wT, q.sT, q.tT, q.actT, (const half_t*)q.w3, q.s3, q.t3, q.act3,
                       (const half_t*)z.weight, z.scale, z.shift, z.act, z.Cout, z.out, z.out_nchw_f32, q.B, q.Hin, q.Win, q.cs_in};
  }
  for (int g = ng; g < CFEN_MAX_GROUPS; ++g) ga.g[g] = ga.g[0];
  const int bands = 2 * u[0].Hin / 8;
  int nseg = std::max(1, std::min(cfen_tune_tail_segments(), bands));
  while (bands % nseg) --nseg;
  const long long nblk = (long long)u[0].B * (2 * u[0].Win / 64) * nseg;
  CFEN_CHECK_ARG(nblk < (1ll << 31), "tail (fused): grid too large");
  // (per device: a process that drives several GPUs must raise the limit on each of them -- ADVICE r05)
  static bool attr_set[64] = {};
  if (cfen_first_use_on_device(attr_set)) {
    if (hipFuncSetAttribute((const void*)k_tail_fused, hipFuncAttributeMaxDynamicSharedMemorySize, TF_LDS) != hipSuccess) {
      cfen_set_error("tail (fused): cannot reserve %d bytes of LDS", TF_LDS);
      return CFEN_ERR_HIP;
    }
  }
  static unsigned long long* stamps = nullptr;
  const bool stamping = (cfen_tune_tail_debug() & 64) != 0;
  if (stamping && !stamps && hipMalloc(&stamps, 96 * 3 * 2 * sizeof(unsigned long long)) != hipSuccess) stamps = nullptr;
  if (stamping && stamps) (void)hipMemsetAsync(stamps, 0, 96 * 3 * 2 * sizeof(unsigned long long), s);
  CFEN_LAUNCH(k_tail_fused, dim3(cfen_grid8(nblk), 1, ng), dim3(768), TF_LDS, s, ga, (int)nblk, bands / nseg, cfen_tune_tail_debug(), cfen_tune_tail_balance(), stamping ? stamps : nullptr);
  if (stamping && stamps && !cfen_recorder()) {   // timing experiment: per-step stamps of workgroup 0 to stderr (100 MHz ticks -> us)
    unsigned long long hst[96 * 3 * 2];
    if (hipStreamSynchronize(s) == hipSuccess && hipMemcpy(hst, stamps, sizeof(hst), hipMemcpyDeviceToHost) == hipSuccess) {
      const unsigned long long t0 = hst[2];
      for (int st = 1; st < 96 && hst[(st * 3) * 2]; ++st) {
        fprintf(stderr, "tail step %2d:", st);
        for (int g = 0; g < 3; ++g) fprintf(stderr, "  %c start %7.2f busy %5.2f", "ABC"[g], (double)(hst[(st * 3 + g) * 2] - t0) * 0.01, hst[(st * 3 + g) * 2 + 1] ? (double)(hst[(st * 3 + g) * 2 + 1] - hst[(st * 3 + g) * 2]) * 0.01 : 0.0);
        fprintf(stderr, "\n");
      }
    }
  }
  CFEN_CHECK_LAUNCH("tail (fused)");
  return CFEN_OK;
}
