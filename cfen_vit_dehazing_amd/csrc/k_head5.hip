// head.0.0 (default_conv 5x5, 3 -> n_feats / 2 = 12 channels, pad 2, bias; reference models/networks_iid_hlgvit_crs_gd4_cfs_v3.py:123-127 with
// models/common.py:11-14) read STRAIGHT from the network input -- fp32 NCHW as model.set_input hands it over (models/model_iid_dehazing.py:143), or
// the decoded uint8 HWC image with ToTensor + Normalize(0.5, 0.5) folded in (data/base_dataset.py:44-46).
//
// Replaces two launches of the round-3 plan: k_nchw_to_nhwc (fp32 planes -> an 8-channel fp16 NHWC map, 19 us) and k_conv_tile<16, 5> on that map
// (66 us: with 3 real channels in 16-byte pixels and 5 taps in chunks of 4 the MFMA work was 4.3x the algorithmic flops -- it was MFMA-issue bound
// on a layer that moves 92 MB).  Here a pixel is 8 bytes in LDS (3 channels + a zero), so the 64 bytes an MFMA chunk consumes per column are
// EIGHT horizontally adjacent taps: one chunk per kernel row, 5 MFMAs per 16 x 16 output tile instead of 10, and the 8-channel input map is
// never written.  Workgroup = 64 x 8 output pixels, wave w = the 16-pixel column strip w; halo of 12 x 68 pixels staged once.
//   B fragment of (input row iy, lane (pixel r16, h)): 16 bytes at  iy * RB + (16 w + r16 + 2 h) * 8  = taps 2h, 2h + 1 (8-byte aligned: two
//   ds_read_b64); weights "w5" [16][5 dy][8 taps][4 c] (taps 5..7 and channel 3 zero: packing.pack_head5).
#include "cfen_common.hpp"
#include "cfen_internal.hpp"

namespace {

constexpr int H5_R = 8, H5_WT = 72, H5_RB = H5_WT * 8, H5_ROWS = H5_R + 4;

struct Head5Args {
  const void* in; const half_t* w5; const float* scale; const float* shift; half_t* out;
  int B, H, W, cs_out, act;
};

template <int U8>
__global__ __launch_bounds__(256) void k_head5(Head5Args a, int nblk) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[H5_ROWS * H5_RB];
  const int blk = (int)xcd_chunked_block(blockIdx.x, gridDim.x);
  if (blk >= nblk) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, h = lane >> 4;
  const int tiles_x = a.W / 64, tiles_y = a.H / H5_R;
  const int tx = blk % tiles_x, ty = (blk / tiles_x) % tiles_y, b = blk / (tiles_x * tiles_y);
  const int x0 = tx * 64, y0 = ty * H5_R;

  // ---- stage the halo: every load in flight before the first LDS store ----
  constexpr int NPIX = H5_ROWS * H5_WT, NIT = (NPIX + 255) / 256;
  float v[NIT][3];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int idx = tid + i * 256;
    const int col = idx % H5_WT, row = idx / H5_WT;
    const int gy = y0 - 2 + row, gx = x0 - 2 + col;
    const bool ok = idx < NPIX && col < 68 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    const int cy = min(max(gy, 0), a.H - 1), cx = min(max(gx, 0), a.W - 1);   // unconditional loads at clamped addresses, validity applied after
    if (U8) {
      const unsigned char* p = (const unsigned char*)a.in + (((size_t)b * a.H + cy) * a.W + cx) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) v[i][c] = ok ? ((float)p[c] / 255.f - 0.5f) / 0.5f : 0.f;    // ToTensor, Normalize(0.5, 0.5): the reference's float32 arithmetic
    } else {
      const float* p = (const float*)a.in + ((size_t)b * 3 * a.H + cy) * a.W + cx;
#pragma unroll
      for (int c = 0; c < 3; ++c) { const float f = p[(size_t)c * a.H * a.W]; v[i][c] = ok ? f : 0.f; }
    }
  }
  half8 wf[5];
  {
    const half_t* wp = a.w5 + r16 * 160 + h * 8;
#pragma unroll
    for (int dy = 0; dy < 5; ++dy) wf[dy] = *reinterpret_cast<const half8*>(wp + dy * 32);
  }
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int idx = tid + i * 256;
    if (idx < NPIX) {
      const half4 o = {(half_t)v[i][0], (half_t)v[i][1], (half_t)v[i][2], (half_t)0};
      *reinterpret_cast<half4*>(&lds[idx * 8]) = o;
    }
  }
  __syncthreads();

  floatx4 acc[H5_R];
#pragma unroll
  for (int r = 0; r < H5_R; ++r) acc[r] = floatx4{0.f, 0.f, 0.f, 0.f};
  const unsigned char* lp = lds + (wave * 16 + r16 + 2 * h) * 8;
#pragma unroll
  for (int iy = 0; iy < H5_ROWS; ++iy) {
    const half4 lo = *reinterpret_cast<const half4*>(lp + iy * H5_RB), hi = *reinterpret_cast<const half4*>(lp + iy * H5_RB + 8);
    const half8 bf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
    for (int r = 0; r < H5_R; ++r) {
      const int dy = iy - r;
      if (dy >= 0 && dy < 5) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[dy], bf, acc[r], 0, 0, 0);
    }
  }

  // ---- epilogue: lane owns channels 4h .. 4h+3 of pixel (y0 + r, x0 + 16 w + r16): a wave instruction stores 512 contiguous bytes ----
  const int n = 4 * h, ox = x0 + wave * 16 + r16;
  if (n >= a.cs_out) return;
  const floatx4 sc = *reinterpret_cast<const floatx4*>(a.scale + n), sh = *reinterpret_cast<const floatx4*>(a.shift + n);
  half_t* op = a.out + (((size_t)b * a.H + y0) * a.W + ox) * a.cs_out + n;
#pragma unroll
  for (int r = 0; r < H5_R; ++r) {
    floatx4 o = acc[r] * sc + sh;
    if (a.act == 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
    }
    store4<half_t>(op + (size_t)r * a.W * a.cs_out, o);
  }
}

}  // namespace

bool cfen_head5_supported(int dtype, int Cout_pad, int cs_out, int H, int W) {
  return dtype == 1 && Cout_pad == 16 && cs_out == 16 && H % H5_R == 0 && W % 64 == 0;
}

int cfen_head5_impl(int dtype, int in_u8, const void* in, const void* w5, const float* scale, const float* shift, void* out, int B, int H, int W,
                    int cs_out, int act, hipStream_t s) {
  CFEN_CHECK_ARG(cfen_head5_supported(dtype, 16, cs_out, H, W) && B > 0, "head5: fp16, 16-channel output map, H %% 8 == 0, W %% 64 == 0 only");
  CFEN_CHECK_ARG(in && w5 && scale && shift && out && cfen_aligned16(w5) && cfen_aligned16(scale) && cfen_aligned16(shift) && cfen_aligned16(out) &&
                 (in_u8 || (reinterpret_cast<uintptr_t>(in) & 3) == 0), "head5: null or misaligned pointer");
  CFEN_CHECK_ARG(act == 0 || act == 1, "head5: activation 0 / 1 only");
  const Head5Args a{in, (const half_t*)w5, scale, shift, (half_t*)out, B, H, W, cs_out, act};
  const long long nblk = (long long)B * (H / H5_R) * (W / 64);
  CFEN_CHECK_ARG(nblk < (1ll << 31), "head5: problem too large");
  if (in_u8)
    CFEN_LAUNCH(k_head5<1>, dim3(cfen_grid8(nblk)), dim3(256), 0, s, a, (int)nblk);
  else
    CFEN_LAUNCH(k_head5<0>, dim3(cfen_grid8(nblk)), dim3(256), 0, s, a, (int)nblk);
  CFEN_CHECK_LAUNCH("head5");
  return CFEN_OK;
}
