// Fused chains of the full-resolution 12-channel CNN layers: an intermediate map that one layer writes and the next one reads with a 3 x 3 halo
// stays in LDS (halo recomputed per workgroup) instead of making a 67 MB round trip through HBM per image batch.
//
//   k_resblock_fused : head.0.1 = ResBlock  x + conv3x3(ReLU(conv3x3(x) + b)) + b  (reference models/common.py:41-62 behind the 5x5 head conv,
//                      models/networks_iid_hlgvit_crs_gd4_cfs_v3.py:123-127): two k_conv_tile launches -> one; the hidden map never reaches HBM.
//                      (Stages B and C of round 3's k_head_fused with the conv5x5 output STAGED from HBM instead of recomputed: that kernel lost to
//                      the three launches because its 5x5 stage was MFMA-issue bound; k_head5 does that layer in 28 us now.)
//
// Same fragment conventions as k_conv_tile.hip: 32-byte NHWC pixels (16 channels, 12 real), a 64-byte MFMA chunk = 2 horizontally adjacent taps,
// weights in the "rows" layout ([16][3 dy][2 chunks][32]), a wave owns a 16-pixel column strip.  The accumulation order per output pixel is that of
// k_conv_tile, and intermediates are rounded to fp16 exactly as the stored maps were: results are bitwise those of the unfused launches.
#include "cfen_common.hpp"
#include "cfen_internal.hpp"

namespace {

constexpr int RF_R = 8;                            // output rows per workgroup
constexpr int RF_WT = 83;                          // tile columns: image columns x0 - 9 .. x0 + 73
constexpr int RF_R1 = RF_R + 4, RF_R2 = RF_R + 2;  // t1 (block input) rows y0 - 2 .., t2 (hidden map) rows y0 - 1 ..
constexpr int RF_RB = RF_WT * 32;
constexpr int RF_LDS = (RF_R1 + RF_R2) * RF_RB;
static_assert(2 * RF_LDS <= 160 * 1024, "two workgroups a CU");

struct ResblockArgs {
  const half_t* in; half_t* out; const half_t* wa; const half_t* wb; const float* sa; const float* ta; const float* sb; const float* tb;
  int B, H, W;
};

__global__ __launch_bounds__(320) void k_resblock_fused(ResblockArgs a, int nblk) {
  typedef half_t T;
  typedef half8 frag;
  __shared__ __attribute__((aligned(16))) unsigned char lds[RF_LDS];
  unsigned char* const t1 = lds;
  unsigned char* const t2 = lds + RF_R1 * RF_RB;

  const int blk = (int)xcd_chunked_block(blockIdx.x, gridDim.x);
  if (blk >= nblk) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, h = lane >> 4;
  const int tiles_x = a.W / 64, tiles_y = a.H / RF_R;
  const int tx = blk % tiles_x, ty = (blk / tiles_x) % tiles_y, b = blk / (tiles_x * tiles_y);
  const int x0 = tx * 64, y0 = ty * RF_R;
  const unsigned char* src = (const unsigned char*)a.in + (size_t)b * a.H * a.W * 32;

  // ---- stage the block input with its 2-pixel halo (zero outside the image): all loads in flight before the first LDS store ----
  constexpr int NPIECE = RF_R1 * RF_WT * 2, NIT = (NPIECE + 319) / 320;
  frag stg[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int idx = tid + i * 320;
    const int piece = idx & 1, col = (idx >> 1) % RF_WT, row = (idx >> 1) / RF_WT;
    const int gy = y0 - 2 + row, gx = x0 - 9 + col;
    const bool ok = idx < NPIECE && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    stg[i] = ok ? *reinterpret_cast<const frag*>(src + ((size_t)gy * a.W + gx) * 32 + piece * 16) : Mma<T>::zero();
  }
  frag wa[3][2], wb[3][2];
  {
    const T* pa = a.wa + (size_t)r16 * (3 * 2 * 32) + h * 8;
    const T* pb = a.wb + (size_t)r16 * (3 * 2 * 32) + h * 8;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        wa[dy][c] = load_frag<T>(pa + (dy * 2 + c) * 32);
        wb[dy][c] = load_frag<T>(pb + (dy * 2 + c) * 32);
      }
  }
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int idx = tid + i * 320;
    if (idx < NPIECE) *reinterpret_cast<frag*>(t1 + idx * 16) = stg[i];       // piece order == tile order
  }
  __syncthreads();

  const int n = 4 * h;
  const int gcol = x0 - 8 + wave * 16 + r16;         // image column of this lane's pixel in the 80-wide hidden strip
  const bool colin = gcol >= 0 && gcol < a.W;
  // ---- t2 = relu(conv3x3(t1) * scale + shift) on rows y0 - 1 .. y0 + R, columns x0 - 8 .. x0 + 71 (zero outside the image: the next conv's padding) ----
  {
    floatx4 acc[RF_R2];
#pragma unroll
    for (int r = 0; r < RF_R2; ++r) acc[r] = floatx4{0.f, 0.f, 0.f, 0.f};
    const unsigned char* lp = t1 + (wave * 16 + r16) * 32 + h * 16;
#pragma unroll
    for (int iy = 0; iy < RF_R1; ++iy)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const frag bf = *reinterpret_cast<const frag*>(lp + iy * RF_RB + c * 64);
#pragma unroll
        for (int r = 0; r < RF_R2; ++r) {
          const int dy = iy - r;
          if (dy >= 0 && dy < 3) acc[r] = Mma<T>::mma(wa[dy][c], bf, acc[r]);
        }
      }
    const floatx4 sc = *reinterpret_cast<const floatx4*>(a.sa + n), sh = *reinterpret_cast<const floatx4*>(a.ta + n);
#pragma unroll
    for (int r = 0; r < RF_R2; ++r) {
      const int gy = y0 - 1 + r;
      floatx4 v = acc[r] * sc + sh;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      if (!(colin && gy >= 0 && gy < a.H)) v = floatx4{0.f, 0.f, 0.f, 0.f};
      const half4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
      *reinterpret_cast<half4*>(t2 + r * RF_RB + (wave * 16 + r16 + 1) * 32 + n * 2) = o;
    }
  }
  __syncthreads();
  // ---- out = conv3x3(t2) * scale + shift + t1 on the 64 x R tile (waves 0..3) ----
  if (wave < 4) {
    floatx4 acc[RF_R];
#pragma unroll
    for (int r = 0; r < RF_R; ++r) acc[r] = floatx4{0.f, 0.f, 0.f, 0.f};
    const unsigned char* lp = t2 + (wave * 16 + r16 + 8) * 32 + h * 16;
#pragma unroll
    for (int iy = 0; iy < RF_R2; ++iy)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const frag bf = *reinterpret_cast<const frag*>(lp + iy * RF_RB + c * 64);
#pragma unroll
        for (int r = 0; r < RF_R; ++r) {
          const int dy = iy - r;
          if (dy >= 0 && dy < 3) acc[r] = Mma<T>::mma(wb[dy][c], bf, acc[r]);
        }
      }
    const floatx4 sc = *reinterpret_cast<const floatx4*>(a.sb + n), sh = *reinterpret_cast<const floatx4*>(a.tb + n);
    const int ox = x0 + wave * 16 + r16;
#pragma unroll
    for (int r = 0; r < RF_R; ++r) {
      floatx4 v = acc[r] * sc + sh;
      const half4 res = *reinterpret_cast<const half4*>(t1 + (r + 2) * RF_RB + (wave * 16 + r16 + 9) * 32 + n * 2);   // the block input at (y0 + r, ox)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += (float)res[e];
      store4<T>(a.out + (((size_t)b * a.H + y0 + r) * a.W + ox) * 16 + n, v);
    }
  }
}

}  // namespace

bool cfen_resblock_fused_supported(int dtype, int cs, int C, int H, int W) { return dtype == 1 && cs == 16 && C <= 16 && H % RF_R == 0 && W % 64 == 0; }

// in / out: (B, H, W, 16) fp16 maps; wa / wb: "<layer>.wr" of the two 3x3 convolutions, (sa, ta) / (sb, tb) their epilogue tables
int cfen_resblock_fused_impl(int dtype, const void* in, void* out, const void* wa, const float* sa, const float* ta, const void* wb, const float* sb,
                             const float* tb, int B, int H, int W, hipStream_t s) {
  CFEN_CHECK_ARG(cfen_resblock_fused_supported(dtype, 16, 12, H, W) && B > 0, "resblock (fused): fp16, image edges multiples of 8 / 64 (got dtype %d, %d x %d)", dtype, H, W);
  CFEN_CHECK_ARG(in && out && wa && wb && sa && ta && sb && tb && in != out, "resblock (fused): null pointer / in-place");
  CFEN_CHECK_ARG(cfen_aligned16(in) && cfen_aligned16(out) && cfen_aligned16(wa) && cfen_aligned16(wb) && cfen_aligned16(sa) && cfen_aligned16(ta) &&
                 cfen_aligned16(sb) && cfen_aligned16(tb), "resblock (fused): pointers must be 16-byte aligned");
  const ResblockArgs a{(const half_t*)in, (half_t*)out, (const half_t*)wa, (const half_t*)wb, sa, ta, sb, tb, B, H, W};
  const long long nblk = (long long)B * (H / RF_R) * (W / 64);
  CFEN_CHECK_ARG(nblk < (1ll << 31), "resblock (fused): grid too large");
  CFEN_LAUNCH(k_resblock_fused, dim3(cfen_grid8(nblk)), dim3(320), 0, s, a, (int)nblk);
  CFEN_CHECK_LAUNCH("resblock (fused)");
  return CFEN_OK;
}

// -------------------------------------------------------------------------------------------------------------------------------------------
//   k_up_conv3_fused : us_conv_d01* (ConvTranspose2d(4, 2, 1) 24 -> 12 + ActNorm2d + ReLU, v3:318-322) followed by the tail's first layer
//                      (Conv2d 3x3 12 -> 12 + ActNorm2d (R, D) + ReLU, v3:348-351, 360-363, 372-375), the R / S / D copies as one grouped launch.
//                      Unfused these were k_convT_tile (276 MB, 79 us) and k_conv_tile<32, 3> (402 MB, 90 us) -- both at the HBM rate of their own
//                      maps; here the 12-channel full-resolution map between them (201 MB written, 201 MB read back) stays in LDS: a workgroup
//                      computes the ConvTranspose on a 66 x 10 halo region of its 64 x 8 output tile (6 x 48 base pixels x 4 parity phases:
//                      2.25x the minimum MFMA work, on kernels that ran at 12 % MFMA-busy) and runs the 3x3 on that tile.
// The ConvTranspose stage is k_convT_tile's loop (one parity phase per wave, an input-row fragment feeds both output rows that use it), the 3x3
// stage is k_conv_tile's: same accumulation order per output, the intermediate rounded to fp16 as the stored map was -> bitwise the unfused results.
// `up_out` (optional): the ConvTranspose output of the tile interior is ALSO stored (the us_conv_d01* stage of SURVEY Appendix D, for parity tests).
namespace {

constexpr int UF_RYB = 6, UF_NX = 3;                       // base rows / 16-pixel base column tiles computed per workgroup
constexpr int UF_ROWS = UF_RYB + 2, UF_WT = 16 * UF_NX + 2; // input halo: 8 x 50 base pixels of 64 bytes
constexpr int UF_RB = UF_WT * 64;
constexpr int UF_T1W = 67, UF_T1R = 10, UF_T1B = UF_T1W * 32; // the 3x3's input tile: full-resolution columns X0 - 1 .. X0 + 65, rows Y0 - 1 .. Y0 + 8
constexpr int UF_LDS = UF_ROWS * UF_RB > UF_T1R * UF_T1B ? UF_ROWS * UF_RB : UF_T1R * UF_T1B;   // the 3x3's tile takes the halo's place (25.6 KB: five to six
static_assert(6 * UF_LDS <= 160 * 1024, "six workgroups a CU");                                // workgroups a CU -- the stage chain of a workgroup is latency bound)

struct UpConv3Args {
  const half_t* in; const half_t* wT; const float* sT; const float* tT; int actT;
  const half_t* w3; const float* s3; const float* t3; int act3;
  half_t* out; half_t* up_out;
  int B, Hin, Win, cs_in;
};

template <int PIXB> CFEN_DEV int uf_swz(int col) { return ((col >> 2) & 1) << 1; }   // k_conv_tile.hip convt_swz<64>
// Column -> pixel slot of the 3x3's input tile (32-byte pixels).  The ConvTranspose epilogue writes 8 bytes per lane at a pixel stride of 2
// (one parity phase per wave): ds_write_b64 is banked in groups of 16 lanes over 128 bytes, so at the identity map the 16 lanes of a group fall on
// two bank pairs -- 8-way, 32 LDS cycles per store against 4 (SQ_LDS_BANK_CONFLICT 1.4x the kernel's conflict-free LDS cycles, round 4).  Swapping
// the pixel pairs of columns 4..7 (mod 8) brings that to 3-way and keeps every ds_read_b128 of the 3x3 stage conflict-free for its column bases
// 16 w + 2 c (exhaustive search over per-column XOR flips under the guide's lane groups: no 2-way map with conflict-free reads exists).
CFEN_DEV int uf_t1col(int col) { return col ^ ((col >> 2) & 1); }

__global__ __launch_bounds__(256, 3) void k_up_conv3_fused(Grouped<UpConv3Args> ga, int nblk) {
  const UpConv3Args& a = ga.g[blockIdx.z];
  typedef half_t T;
  typedef half8 frag;
  __shared__ __attribute__((aligned(16))) unsigned char lds[UF_LDS];
  unsigned char* const t1 = lds;   // aliases the input halo: written only after every wave has finished reading it

  const int blk = (int)xcd_chunked_block(blockIdx.x, gridDim.x);
  if (blk >= nblk) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, h = lane >> 4;
  const int Hf = 2 * a.Hin, Wf = 2 * a.Win;
  const int tiles_x = Wf / 64, tiles_y = Hf / 8;
  const int tx = blk % tiles_x, ty = (blk / tiles_x) % tiles_y, b = blk / (tiles_x * tiles_y);
  const int X0 = tx * 64, Y0 = ty * 8, xb0 = tx * 32, yb0 = ty * 4;

  // ---- stage the input halo: base rows yb0 - 2 .. yb0 + 5, base columns xb0 - 2 .. xb0 + 47 (zero outside the map) ----
  // (one row of the halo per pass, 200 pieces = threads 0..199: piece and column of a thread are constants, the row is wave-uniform -- the flat
  // piece index of k_convT_tile costs a division by 50 and a 64-bit address per load; this kernel is vector-instruction bound, not HBM bound)
  const int src_pixb = a.cs_in * 2, src_pieces = src_pixb / 16;
  const unsigned char* src = (const unsigned char*)a.in + (size_t)b * a.Hin * a.Win * src_pixb;
  const int spiece = tid & 3, scol = tid >> 2;
  const int sgx = xb0 - 2 + scol;
  const bool sok = tid < UF_WT * 4 && spiece < src_pieces && sgx >= 0 && sgx < a.Win;
  const int soff = min(max(sgx, 0), a.Win - 1) * src_pixb + min(spiece, src_pieces - 1) * 16;     // clamped: the load is unconditional
  frag stg[UF_ROWS];
#pragma unroll
  for (int row = 0; row < UF_ROWS; ++row) {
    const int gy = yb0 - 2 + row;
    const frag v = *reinterpret_cast<const frag*>(src + (size_t)min(max(gy, 0), a.Hin - 1) * a.Win * src_pixb + soff);
    stg[row] = (sok && gy >= 0 && gy < a.Hin) ? v : Mma<T>::zero();
  }
  const int phase = wave, py = phase >> 1, px = phase & 1;
  frag wf[4];
  {
    const T* wp = a.wT + ((size_t)phase * 16 + r16) * 128 + h * 8;
#pragma unroll
    for (int t = 0; t < 4; ++t) wf[t] = load_frag<T>(wp + t * 32);
  }
  if (tid < UF_WT * 4) {
    unsigned char* sdst = lds + scol * 64 + ((spiece ^ uf_swz<64>(scol)) << 4);
#pragma unroll
    for (int row = 0; row < UF_ROWS; ++row) *reinterpret_cast<frag*>(sdst + row * UF_RB) = stg[row];
  }
  __syncthreads();

  const int n = 4 * h;
  // ---- ConvTranspose, one parity phase per wave: acc[r][xq] = base pixel (yb0 - 1 + r, xb0 - 1 + 16 xq + r16) -> output (2 yb + py, 2 xb + px) ----
  {
    floatx4 acc[UF_RYB][UF_NX];
#pragma unroll
    for (int r = 0; r < UF_RYB; ++r)
#pragma unroll
      for (int xq = 0; xq < UF_NX; ++xq) acc[r][xq] = floatx4{0.f, 0.f, 0.f, 0.f};
    const unsigned char* lp = lds + (1 + py) * UF_RB;
#pragma unroll
    for (int q = -1; q < UF_RYB; ++q) {
#pragma unroll
      for (int txx = 0; txx < 2; ++txx) {
#pragma unroll
        for (int xq = 0; xq < UF_NX; ++xq) {
          const int col = xq * 16 + r16 + px + 1 - txx;
          const frag bf = *reinterpret_cast<const frag*>(lp + q * UF_RB + col * 64 + ((h ^ uf_swz<64>(col)) << 4));
          if (q >= 0) acc[q][xq] = Mma<T>::mma(wf[txx], bf, acc[q][xq]);
          if (q + 1 < UF_RYB) acc[q + 1][xq] = Mma<T>::mma(wf[2 + txx], bf, acc[q + 1][xq]);
        }
      }
    }
    const floatx4 sc = *reinterpret_cast<const floatx4*>(a.sT + n), sh = *reinterpret_cast<const floatx4*>(a.tT + n);
    __syncthreads();                                        // every wave is done with the halo: its place becomes the 3x3's input tile
    if (tid < UF_T1R * 2) *reinterpret_cast<frag*>(t1 + (tid >> 1) * UF_T1B + uf_t1col(66) * 32 + (tid & 1) * 16) = Mma<T>::zero();   // column 66 meets zero weights only
    // column part of the epilogue, once per lane and column tile: the tile column of the lane's output, whether it belongs to the 66-wide tile, and
    // the (scale, shift) pair with the image-border mask folded in (a column outside the image becomes exact zeros: the 3x3's padding)
    int tcol[UF_NX], tpix[UF_NX];
    bool tin[UF_NX];
    floatx4 scx[UF_NX], shx[UF_NX];
#pragma unroll
    for (int xq = 0; xq < UF_NX; ++xq) {
      tcol[xq] = 2 * (xq * 16 + r16) + px - 1;              // X - (X0 - 1)
      tin[xq] = tcol[xq] >= 0 && tcol[xq] <= 65;
      tpix[xq] = uf_t1col(tcol[xq] & 127);
      const int X = X0 - 1 + tcol[xq];
      const bool ximg = X >= 0 && X < Wf;
      scx[xq] = ximg ? sc : floatx4{0.f, 0.f, 0.f, 0.f};
      shx[xq] = ximg ? sh : floatx4{0.f, 0.f, 0.f, 0.f};
    }
    unsigned char* const tw = t1 + n * 2;
#pragma unroll
    for (int r = 0; r < UF_RYB; ++r) {
      const int trow = 2 * r + py - 1;                      // tile row of the output: Y - (Y0 - 1); wave-uniform
      if (trow < 0 || trow >= UF_T1R) continue;
      const int Y = Y0 - 1 + trow;
      const bool yimg = Y >= 0 && Y < Hf;                   // wave-uniform: a row outside the image is written as zeros
#pragma unroll
      for (int xq = 0; xq < UF_NX; ++xq) {
        floatx4 v = acc[r][xq] * scx[xq] + shx[xq];
        if (a.actT == 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (!yimg) v = floatx4{0.f, 0.f, 0.f, 0.f};
        const half4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
        if (tin[xq]) *reinterpret_cast<half4*>(tw + trow * UF_T1B + tpix[xq] * 32) = o;
        if (a.up_out && tin[xq] && trow >= 1 && trow <= 8 && tcol[xq] >= 1 && tcol[xq] <= 64)
          *reinterpret_cast<half4*>(a.up_out + (((size_t)b * Hf + Y) * Wf + X0 - 1 + tcol[xq]) * 16 + n) = o;
      }
    }
  }
  __syncthreads();
  // ---- 3x3 on the tile: wave w = the 16-pixel column strip w, 8 rows ----
  {
    frag wb[3][2];
    const T* pb = a.w3 + (size_t)r16 * (3 * 2 * 32) + h * 8;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int c = 0; c < 2; ++c) wb[dy][c] = load_frag<T>(pb + (dy * 2 + c) * 32);
    floatx4 acc[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) acc[r] = floatx4{0.f, 0.f, 0.f, 0.f};
    // k quarter h of chunk c = 16 bytes of column 16 w + r16 + 2 c + (h >> 1), half h & 1
    const unsigned char* lp[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) lp[c] = t1 + uf_t1col(wave * 16 + r16 + 2 * c + (h >> 1)) * 32 + (h & 1) * 16;
#pragma unroll
    for (int iy = 0; iy < UF_T1R; ++iy)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const frag bf = *reinterpret_cast<const frag*>(lp[c] + iy * UF_T1B);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int dy = iy - r;
          if (dy >= 0 && dy < 3) acc[r] = Mma<T>::mma(wb[dy][c], bf, acc[r]);
        }
      }
    const floatx4 sc = *reinterpret_cast<const floatx4*>(a.s3 + n), sh = *reinterpret_cast<const floatx4*>(a.t3 + n);
    const int ox = X0 + wave * 16 + r16;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      floatx4 v = acc[r] * sc + sh;
      if (a.act3 == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      store4<T>(a.out + (((size_t)b * Hf + Y0 + r) * Wf + ox) * 16 + n, v);
    }
  }
}

}  // namespace

bool cfen_up_conv3_fused_supported(int dtype, int cs_in, int Cup_pad, int cs_up, int C3_pad, int Hin, int Win) {
  return dtype == 1 && cs_in % 8 == 0 && cs_in * 2 <= 64 && Cup_pad == 16 && cs_up == 16 && C3_pad == 16 && Hin % 4 == 0 && Win % 32 == 0;
}

int cfen_up_conv3_fused_impl_g(int dtype, int ng, const CfenUpConv3* u, hipStream_t s) {
  CFEN_CHECK_ARG(ng >= 1 && ng <= CFEN_MAX_GROUPS && u, "up_conv3 (fused): 1..%d problems per launch", CFEN_MAX_GROUPS);
  Grouped<UpConv3Args> ga;
  memset(&ga, 0, sizeof(ga));
  for (int g = 0; g < ng; ++g) {
    const CfenUpConv3& q = u[g];
    CFEN_CHECK_ARG(cfen_up_conv3_fused_supported(dtype, q.cs_in, 16, 16, 16, q.Hin, q.Win) && q.B > 0, "up_conv3 (fused): unsupported geometry");
    CFEN_CHECK_ARG(q.B == u[0].B && q.Hin == u[0].Hin && q.Win == u[0].Win && q.cs_in == u[0].cs_in, "up_conv3 (fused): grouped problems must share the geometry");
    CFEN_CHECK_ARG(q.in && q.wT && q.sT && q.tT && q.w3 && q.s3 && q.t3 && q.out, "up_conv3 (fused): null pointer");
    CFEN_CHECK_ARG(cfen_aligned16(q.in) && cfen_aligned16(q.wT) && cfen_aligned16(q.sT) && cfen_aligned16(q.tT) && cfen_aligned16(q.w3) && cfen_aligned16(q.s3) &&
                   cfen_aligned16(q.t3) && cfen_aligned16(q.out) && cfen_aligned16(q.up_out), "up_conv3 (fused): pointers must be 16-byte aligned");
    CFEN_CHECK_ARG((q.actT == 0 || q.actT == 1) && (q.act3 == 0 || q.act3 == 1), "up_conv3 (fused): activations 0 / 1 only");
    ga.g[g] = UpConv3Args{(const half_t*)q.in, (const half_t*)q.wT, q.sT, q.tT, q.actT, (const half_t*)q.w3, q.s3, q.t3, q.act3, (half_t*)q.out,
                          (half_t*)q.up_out, q.B, q.Hin, q.Win, q.cs_in};
  }
  for (int g = ng; g < CFEN_MAX_GROUPS; ++g) ga.g[g] = ga.g[0];
  const long long nblk = (long long)u[0].B * (2 * u[0].Hin / 8) * (2 * u[0].Win / 64);
  CFEN_CHECK_ARG(nblk < (1ll << 31), "up_conv3 (fused): grid too large");
  CFEN_LAUNCH(k_up_conv3_fused, dim3(cfen_grid8(nblk), 1, ng), dim3(256), 0, s, ga, (int)nblk);
  CFEN_CHECK_LAUNCH("up_conv3 (fused)");
  return CFEN_OK;
}
