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
// The 35 Toeplitz fragments of the 7x7 (k_conv_tile.hip: k_conv7_tz) are 140 registers a lane, the other stages need ~130: one wave cannot hold both (a
// 4-wave version spilled 157 registers), so the workgroup has ROLES -- waves 0..3 stage the input and run A and B, waves 4..7 run C one band behind, a producer
// and a consumer wave on every SIMD.  LDS (one workgroup a CU): staged input 6 x 50 x 64 B, t1 a ring of 16 rows x 84 x 32 B (no carry copy), t2 a ring of
// 24 rows x 72 x 32 B (stage B of band j never touches what stage C of band j-1 reads) = 114.8 KB; two barriers per band.
#include "cfen_common.hpp"
#include "cfen_internal.hpp"
#include "cfen_conv.hpp"

namespace {

constexpr int TF_INROWS = 6, TF_INW = 50, TF_INS = 40, TF_INB = TF_INW * 64;   // staged base rows 4j+1 .. 4j+6, base columns xb0-3 .. (40 staged, 50 addressed)
constexpr int TF_T1W = 84, TF_T1B = TF_T1W * 32, TF_T1R = 16, TF_T1KEEP = 76;  // t1 columns X0-4 .., kept while < 76 (stage B's zero-weight fourth tap must read finite values)
constexpr int TF_ZW = 72, TF_ZB = TF_ZW * 32, TF_ZNCH = 5, TF_ZKPAD = 7 * TF_ZNCH * 32, TF_T2R = 24;
constexpr int TF_ZBUF = 8 * 4 * 16 * 16;                                           // the 7x7's raw accumulators of a band: [row 8][channel 4][pixel quad 16] fp32 x 4
constexpr int TF_LDS = TF_INROWS * TF_INB + TF_T1R * TF_T1B + TF_T2R * TF_ZB + TF_ZBUF;
static_assert(TF_LDS <= 160 * 1024, "one workgroup a CU");

struct TailArgs {
  const half_t* in; const half_t* wT; const float* sT; const float* tT; int actT;
  const half_t* w3; const float* s3; const float* t3; int act3;
  const half_t* w7; const float* s7; const float* t7; int act7; int Cout; void* out; int out_mode;   // out_mode 1: fp32 NCHW, 2: uint8 HWC x 3 (util.tensor2im)
  int B, Hin, Win, cs_in;
};

CFEN_DEV int tf_swz(int col) { return ((col >> 2) & 1) << 1; }            // k_fuse.hip uf_swz<64>
CFEN_DEV int tf_t1col(int col) { return col ^ ((col >> 2) & 1); }         // k_fuse.hip uf_t1col
CFEN_DEV int tf_zpiece(int P, int q) { const int slot = 2 * P + q; return (slot & ~7) | ((slot & 7) ^ ((P >> 2) & 6)); }   // k_conv7_tz's halo swizzle
CFEN_DEV int tf_t2slot(int y) { return (y + 4 * TF_T2R) % TF_T2R; }       // y >= -4 * 24

__global__ __launch_bounds__(768) void k_tail_fused(Grouped<TailArgs> ga, int nblk, int segb, int dbg) {
  const TailArgs& a = ga.g[blockIdx.z];
  typedef half_t T;
  typedef half8 frag;
  extern __shared__ __attribute__((aligned(16))) unsigned char tf_lds[];
  unsigned char* const inl = tf_lds;
  unsigned char* const t1 = tf_lds + TF_INROWS * TF_INB;
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

  if (wave < 8) {
    // ============================ producer waves: stage the input, ConvTranspose (A), 3x3 (B) ============================
    // A: wave = (parity phase, half): output rows 2 hf, 2 hf + 1 of the phase's four, all three column tiles.  B: wave = (row pair, half): rows 2 br, 2 br + 1
    // of the band, column tiles {0, 1, 2} / {3, 4}.
    const int ph = wave & 3, py = ph >> 1, px = ph & 1, hf = wave >> 2, br = wave & 3, ct0 = hf ? 3 : 0;
    const bool third = hf == 0;                        // wave-uniform: this wave has a third column tile in stage B
    frag wf[4], wb[3][2];
    {
      const T* wp = a.wT + ((size_t)ph * 16 + r16) * 128 + h * 8;
#pragma unroll
      for (int t = 0; t < 4; ++t) wf[t] = load_frag<T>(wp + t * 32);
      const T* pb = a.w3 + (size_t)r16 * (3 * 2 * 32) + h * 8;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int c = 0; c < 2; ++c) wb[dy][c] = load_frag<T>(pb + (dy * 2 + c) * 32);
    }
    const floatx4 scT = *reinterpret_cast<const floatx4*>(a.sT + n), shT = *reinterpret_cast<const floatx4*>(a.tT + n);
    const floatx4 sc3 = *reinterpret_cast<const floatx4*>(a.s3 + n), sh3 = *reinterpret_cast<const floatx4*>(a.t3 + n);
    const float sc7 = a.s7[h], sh7 = a.t7[h];            // (h = 3 may read a padding entry of the [16] table; its results are never stored)
    // staging: the 6 rows x 40 columns x 4 pieces of 16 bytes are dealt over the 512 producer threads, two pieces (rows) each
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
    // stage A: LDS offset of the (tap column, column tile) fragment; epilogue: t1 column of (xq, lane) = 32 xq + 2 r16 + px = X - (X0 - 4)
    int a_ld[2][3], a_pix[3];
    bool a_keep[3], a_ximg[3];
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
      a_ximg[xq] = X >= 0 && X < Wf;                   // a column outside the image is the 3x3's zero padding
    }
    // stage B: column tile ct0 + i of the lane = t2 pixel P = 16 ct + r16 = X - (X0 - 3); the reflected copy of a padding column of the 7x7 is stored by its source's owner
    int b_ld[2][3], b_off[3], b_moff[3];
    bool b_keep[3], b_mir[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int ct = ct0 + i, P = 16 * ct + r16, X = X0 - 3 + P;
#pragma unroll
      for (int c = 0; c < 2; ++c) b_ld[c][i] = tf_t1col((16 * ct + r16 + 2 * c + (h >> 1)) & 127) * 32 + (h & 1) * 16;
      b_keep[i] = ct < 5 && P < TF_ZW && X >= 0 && X < Wf;
      b_off[i] = (tf_zpiece(P & 127, h >> 1) << 4) + (h & 1) * 8;
      const int Xm = (X >= 1 && X <= 3) ? -X : (X >= Wf - 4 && X <= Wf - 2) ? 2 * (Wf - 1) - X : -100000;
      const int Pm = Xm - (X0 - 3);
      b_mir[i] = b_keep[i] && Pm >= 0 && Pm < TF_ZW;
      b_moff[i] = (tf_zpiece(Pm & 127, h >> 1) << 4) + (h & 1) * 8;
    }
    auto band_live = [&](int j) { return j < k1 && 8 * j + 11 >= 0 && 8 * j + 4 < Hf; };   // ConvTranspose rows 8j+4 .. 8j+11 meet the image
    frag stg[2];
    auto fetch = [&](int j) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int gy = 4 * j + 1 + s_row[k];
        const frag v = *reinterpret_cast<const frag*>(src + (size_t)min(max(gy, 0), Hin - 1) * Win * src_pixb + s_off[k]);
        stg[k] = (s_ok[k] && gy >= 0 && gy < Hin) ? v : Mma<T>::zero();
      }
    };
    if (band_live(k0 - 2)) fetch(k0 - 2);
    for (int j = k0 - 2; j <= k1 + 1; ++j) {
      const bool a_live = band_live(j);                // wave-uniform
      if (a_live) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
          if (s_use[k]) *reinterpret_cast<frag*>(inl + s_dst[k]) = stg[k];
      }
      __syncthreads();                                 // (A) the staged input of band j is in LDS; stage B of band j-1 is complete
      if (j - 2 >= k0 && !(dbg & 16)) {
        // epilogue of the 7x7 of band j-2 (its accumulators were parked by the consumer waves before this barrier): wave = row of the band, lane (r16, h) =
        // channel h of pixels X0 + 4 r16 .. + 3 -- k_conv7_tz's epilogue
        const int y = 8 * (j - 2) + wave;
        floatx4 v = *reinterpret_cast<const floatx4*>(zbuf + ((wave * 4 + h) * 16 + r16) * 16) * sc7 + sh7;
        if (act7 == 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        } else if (act7 == 2 && !(dbg & 8)) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = tanhf(v[e]);
        }
        if (out_mode == 2) {
          unsigned wr = 0;
#pragma unroll
          for (int e = 0; e < 4; ++e) wr |= (unsigned)(unsigned char)(int)((v[e] + 1.f) / 2.0f * 255.0f) << (8 * e);
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
          *reinterpret_cast<floatx4*>((float*)outp + (((size_t)b * Cout + h) * Hf + y) * Wf + X0 + 4 * r16) = v;
        }
      }
      if (j < k1) {
        floatx4 acc[2][3];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int xq = 0; xq < 3; ++xq) acc[r][xq] = floatx4{0.f, 0.f, 0.f, 0.f};
        if (a_live && !(dbg & 1)) {
          const unsigned char* lp = inl + (1 + py + 2 * hf) * TF_INB;
#pragma unroll
          for (int q = -1; q < 2; ++q) {
#pragma unroll
            for (int txx = 0; txx < 2; ++txx) {
#pragma unroll
              for (int xq = 0; xq < 3; ++xq) {
                const frag bf = *reinterpret_cast<const frag*>(lp + q * TF_INB + a_ld[txx][xq]);
                if (q >= 0) acc[q][xq] = Mma<T>::mma(wf[txx], bf, acc[q][xq]);
                if (q + 1 < 2) acc[q + 1][xq] = Mma<T>::mma(wf[2 + txx], bf, acc[q + 1][xq]);
              }
            }
          }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int Y = 8 * j + 4 + 2 * (2 * hf + r) + py;   // wave-uniform
          const bool yimg = a_live && Y >= 0 && Y < Hf;
          unsigned char* const row = t1 + ((Y + 4 * TF_T1R) & (TF_T1R - 1)) * TF_T1B;
#pragma unroll
          for (int xq = 0; xq < 3; ++xq) {
            floatx4 v = floatx4{0.f, 0.f, 0.f, 0.f};
            if (yimg) {
              v = acc[r][xq] * scT + shT;
              if (actT == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
              }
            }
            half4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            if (!a_ximg[xq]) o = half4{(half_t)0, (half_t)0, (half_t)0, (half_t)0};   // (two selects on the packed result)
            if (a_keep[xq]) *reinterpret_cast<half4*>(row + a_pix[xq]) = o;
          }
        }
      }
      __syncthreads();                                 // (B) t1 band j is complete
      if (band_live(j + 1) && !(dbg & 32)) fetch(j + 1);              // in flight behind stage B
      if (j < k1 && j >= k0 - 1) {
        // 3x3 rows 8j+3 .. 8j+10: rows 2 br, 2 br + 1 of the band, this wave's column tiles
        floatx4 bc[2][3];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int i = 0; i < 3; ++i) bc[r][i] = floatx4{0.f, 0.f, 0.f, 0.f};
        if (!(dbg & 2))
#pragma unroll
        for (int iy = 0; iy < 4; ++iy) {
          const unsigned char* lp = t1 + ((8 * j + 2 + 2 * br + iy + 4 * TF_T1R) & (TF_T1R - 1)) * TF_T1B;
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
              if (i == 2 && !third) continue;
              const frag bf = *reinterpret_cast<const frag*>(lp + b_ld[c][i]);
#pragma unroll
              for (int r = 0; r < 2; ++r) {
                const int dy = iy - r;
                if (dy >= 0 && dy < 3) bc[r][i] = Mma<T>::mma(wb[dy][c], bf, bc[r][i]);
              }
            }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int y = 8 * j + 3 + 2 * br + r;        // wave-uniform
          if (y < 0 || y >= Hf) continue;              // the 7x7 reads reflected rows instead
          unsigned char* const row = t2 + tf_t2slot(y) * TF_ZB;
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            if (i == 2 && !third) continue;
            floatx4 v = bc[r][i] * sc3 + sh3;
            if (act3 == 1) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            const half4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            if (b_keep[i]) *reinterpret_cast<half4*>(row + b_off[i]) = o;
            if (b_mir[i]) *reinterpret_cast<half4*>(row + b_moff[i]) = o;
          }
        }
      }
    }
  } else {
    // ============================ consumer waves: 7x7 + tanh of band j-1 (C), one band behind the producers ============================
    const int w = wave - 8;
    frag w7[7][TF_ZNCH];
    {
      const T* pz = a.w7 + (size_t)r16 * TF_ZKPAD + h * 8;
#pragma unroll
      for (int dy = 0; dy < 7; ++dy)
#pragma unroll
        for (int c = 0; c < TF_ZNCH; ++c) w7[dy][c] = load_frag<T>(pz + (dy * TF_ZNCH + c) * 32);
    }
    for (int j = k0 - 2; j <= k1 + 1; ++j) {
      const int cj = j - 1;
      const bool live = cj >= k0 && cj < k1;           // wave-uniform
      floatx4 zc[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};
      __syncthreads();                                 // (A) t2 bands cj-1, cj are complete; the producers' epilogue may read band cj-1's accumulators
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if (live && !(dbg & 4)) {
#pragma unroll
          for (int i4 = 0; i4 < 4; ++i4) {
            const int iy = 4 * half + i4;
            int yy = 8 * cj + 2 * w + iy - 3;
            yy = yy < 0 ? -yy : (yy >= Hf ? 2 * Hf - 2 - yy : yy);     // ReflectionPad2d(3)
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
        if (half == 0) __syncthreads();                // (B) the producers are done reading the previous band's accumulators
      }
      // lane (r16, h): channel h, output pixels X0 + 4 r16 .. + 3 of rows 8 cj + 2 w + r -> zbuf[row][h][r16]; the producer wave of that row finishes it
      if (live) {
#pragma unroll
        for (int r = 0; r < 2; ++r) *reinterpret_cast<floatx4*>(zbuf + (((2 * w + r) * 4 + h) * 16 + r16) * 16) = zc[r];
      }
    }
  }
}

}  // namespace

int& cfen_tune_tail_debug() {   // timing experiments (results invalid): 1 no ConvTranspose MFMAs, 2 no 3x3 MFMAs, 4 no 7x7 MFMAs, 8 no tanh, 16 no output stores, 32 no input fetch
  static int v = 0;
  return v;
}
int& cfen_tune_tail_segments() {   // vertical segments a strip is cut into ("tail.segments"): more workgroups against 2 + 1 warm-up bands per segment
  static int v = 4;
  return v;
}

bool cfen_tail_fused_supported(int dtype, int cs_in, int Cup_pad, int cs_up, int C3_pad, int Hin, int Win, int Cout7, int out_mode) {
  return dtype == 1 && cs_in % 8 == 0 && cs_in * 2 <= 64 && Cup_pad == 16 && cs_up == 16 && C3_pad == 16 && Hin % 4 == 0 && Win % 32 == 0 && Hin >= 8 &&
         Cout7 >= 1 && Cout7 <= 4 && (out_mode == 1 || (out_mode == 2 && (Cout7 == 1 || Cout7 == 3)));
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
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)k_tail_fused, hipFuncAttributeMaxDynamicSharedMemorySize, TF_LDS) != hipSuccess) {
      cfen_set_error("tail (fused): cannot reserve %d bytes of LDS", TF_LDS);
      return CFEN_ERR_HIP;
    }
    attr_set = true;
  }
  CFEN_LAUNCH(k_tail_fused, dim3(cfen_grid8(nblk), 1, ng), dim3(768), TF_LDS, s, ga, (int)nblk, bands / nseg, cfen_tune_tail_debug());
  CFEN_CHECK_LAUNCH("tail (fused)");
  return CFEN_OK;
}
