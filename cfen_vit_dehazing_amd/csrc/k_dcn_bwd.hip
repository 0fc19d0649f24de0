// Deformable convolution backward (DCNv1 and modulated DCNv2): gradients w.r.t. input, offset, mask, weight, bias.
//
// Replaces, for the backward direction, the reference's CUDA extension:
//   deformable_col2im_gpu_kernel / _coord_gpu_kernel                 dcn/src/deform_conv_cuda_kernel.cu:278-421
//   modulated_deformable_col2im_gpu_kernel / _coord_gpu_kernel       dcn/src/deform_conv_cuda_kernel.cu:634-766
//   get_gradient_weight / get_coordinate_weight                      dcn/src/deform_conv_cuda_kernel.cu:116-187
//   deform_conv_backward_input_cuda / _parameters_cuda               dcn/src/deform_conv_cuda.cpp:260-484
//   modulated_deform_conv_cuda_backward                              dcn/src/deform_conv_cuda.cpp:566-679
//
// Arithmetic is fp32 whatever the tensor type (gradients are sums over up to B*Hout*Wout terms), tensors are NCHW like the reference
// API, scratch is the caller's `columns` buffer.  Every kernel maps ADJACENT LANES TO ADJACENT OUTPUT PIXELS and reads / adds to NCHW
// planes, so for smooth offset fields a wave's gathers and atomic adds fall into a few cache lines (a first version with NHWC copies and
// a channel loop per thread spent 28 ms at (8, 24, 256, 256): 64 cache lines per wave instruction).  Five passes:
//   1. k_dcnb_prep      layout copies: grad_out -> pixel-major rows (GEMM operand) and channel-major rows over all B*Hout*Wout pixels
//                       (MFMA operand of pass 5), weight -> per-group [tap*Cg + c][co] (transposed, tap-major)
//   2. token GEMM       column gradients, k-major:  gcol[g][tap*Cg + c][pixel] = sum_co W[g][co][c][tap] * grad_out[pixel][g][co]
//                       (cfen_gemm_impl, exact-fp32 MFMA: the reference's per-image addmm_ of W^T and grad_out, .cpp:332-337)
//   3. k_dcnb_col2im_lds  col2im: a workgroup per 16 x 16 output-pixel tile adds corner weight * gcol (* mask) into a fixed-point LDS
//                       tile of the input footprint (k_dcnb_col2im: every add a global fp32 atomic, the reference's .cu:278-328)
//   4. k_dcnb_sample    one thread per (pixel, tap, deformable group) walks the group's channels once: col2im_coord (d/d offset_h,
//                       d/d offset_w, d/d mask) and the forward's column matrix (masked bilinear samples), written over gcol
//   5. k_dcnb_weight    grad_W[g][co][k] += scale * sum_pixels grad_out[co][pixel] * column[k][pixel]: exact-fp32 MFMA over 16-pixel
//                       chunks, a wave owns 16 k x all co, partial sums of a workgroup's pixel range land with fp32 atomics
// then small epilogue kernels write the NCHW / (Cout, Cg, kh, kw) results in the tensor type.  grad_input / grad_offset / grad_mask
// are overwritten, grad_weight / grad_bias are accumulated into (the reference's addmm_ with beta = 1; its Python side passes zeros).
// Like the reference (atomicAdd, .cu:322) the scatter makes grad_input run-to-run reproducible only up to fp32 summation order.
#include <algorithm>
#include "cfen_common.hpp"
#include "cfen_internal.hpp"

namespace {

struct DcnBwd {
  const void* im; const void* offset; const void* mask; const void* weight; const void* gout;
  void* gin; void* goff; void* gmask; void* gweight; void* gbias;
  int B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, group, dg, Ho, Wo;
  float scale;
  // derived
  int Cg, Cog, Cogp, kk, Kg, Kgp, cpdg;
  long long P, Pp;          // B*Ho*Wo pixels; row pitch of the channel-major copies (multiple of 16)
  // scratch (fp32)
  float* gn;                // [Pp][group][Cogp]   (zero rows past P)
  float* gc;                // [Cout][Pp]
  float* wt;                // [group][Kgp][Cogp]
  float* col;               // pass 2/3: gcol [group][Kgp][Pp];  pass 4/5: forward columns [group][Kg][Pp]
  float* gi;                // [B][C][H][W]
  float* gw;                // [group][Cog][Kg]   (k = tap*Cg + c)
  float* gt;                // [B][tiles of 16 x 16 output pixels][C][E*E]: the LDS col2im tiles, stored whole (E <= DB_EMAX)
  float* gb;                // [Cout][DB_BIAS_PARTS]: partial sums of grad_bias
};

constexpr int DB_EMAX = 32;        // tile footprints up to 32 x 32 input pixels are parked in `gt` (3x3, stride 1: E = 31)
constexpr int DB_BIAS_PARTS = 64;

inline size_t up256(size_t v) { return (v + 255) / 256 * 256; }

struct DcnBwdLayout { size_t gn, gc, wt, col, gi, gw, gt, gb, total; };
DcnBwdLayout dcnb_layout(int B, int C, int H, int W, int Cout, int kk, int Ho, int Wo, int group) {
  const size_t Cg = C / group, Cog = Cout / group, Cogp = (Cog + 3) / 4 * 4, Kgp = (Cg * kk + 3) / 4 * 4;
  const size_t Pp = ((size_t)B * Ho * Wo + 15) / 16 * 16;
  DcnBwdLayout l;
  size_t o = 0;
  l.gn = o; o += up256(Pp * group * Cogp * 4);
  l.gc = o; o += up256((size_t)Cout * Pp * 4);
  l.wt = o; o += up256((size_t)group * Kgp * Cogp * 4);
  l.col = o; o += up256((size_t)group * Kgp * Pp * 4);
  l.gi = o; o += up256((size_t)B * H * W * C * 4);
  l.gw = o; o += up256((size_t)group * Cog * Cg * kk * 4);
  l.gt = o; o += up256((size_t)B * ((Ho + 15) / 16) * ((Wo + 15) / 16) * C * DB_EMAX * DB_EMAX * 4);
  l.gb = o; o += up256((size_t)Cout * DB_BIAS_PARTS * 4);
  l.total = o;
  return l;
}

template <typename T>
__global__ __launch_bounds__(256) void k_dcnb_prep(DcnBwd a, long long n2, long long n3, long long n4) {
  const long long HWo = (long long)a.Ho * a.Wo;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n2 + n3 + n4; i += (long long)gridDim.x * 256) {
    if (i < n2) {                                            // gn[pix][g][ol] (zero in the padding lanes ol >= Cog and rows pix >= P)
      const long long j = i;
      const int ol = (int)(j % a.Cogp), g = (int)((j / a.Cogp) % a.group);
      const long long pix = j / ((long long)a.Cogp * a.group), b = pix / HWo, p = pix % HWo;
      a.gn[j] = (ol < a.Cog && pix < a.P) ? (float)((const T*)a.gout)[(b * a.Cout + g * a.Cog + ol) * HWo + p] : 0.f;
    } else if (i < n2 + n3) {                                // gc[co][pix] (zero in the padding columns pix >= P)
      const long long j = i - n2, pix = j % a.Pp;
      const int co = (int)(j / a.Pp);
      const long long b = pix / HWo, p = pix % HWo;
      a.gc[j] = pix < a.P ? (float)((const T*)a.gout)[(b * a.Cout + co) * HWo + p] : 0.f;
    } else {                                                 // wt[g][k = t*Cg + c][ol] = weight[g*Cog + ol][c][t]
      const long long j = i - n2 - n3;
      const int ol = (int)(j % a.Cogp), k = (int)((j / a.Cogp) % a.Kgp), g = (int)(j / ((long long)a.Cogp * a.Kgp));
      float v = 0.f;
      if (ol < a.Cog && k < a.Kg) {
        const int t = k / a.Cg, c = k % a.Cg;
        v = (float)((const T*)a.weight)[((long long)(g * a.Cog + ol) * a.Cg + c) * a.kk + t];
      }
      a.wt[j] = v;
    }
  }
}

// sample geometry of (pixel, tap): position, validity, corners (the reference's .cu:83-114 / 116-187 rules)
struct DcnTap {
  float h, w, lh, lw, m;
  int hl, wl;
  bool inside, ok0, ok1, ok2, ok3;
};
template <typename T>
CFEN_DEV DcnTap dcnb_tap(const DcnBwd& a, long long b, long long p, int t, int dgi) {
  const long long HWo = (long long)a.Ho * a.Wo;
  const int ho = (int)p / a.Wo, wo = (int)p - ho * a.Wo, i = t / a.kw, j = t - i * a.kw;   // p < Ho*Wo < 2^31 (checked by the host)
  const T* off = (const T*)a.offset + (b * a.dg + dgi) * 2 * a.kk * HWo;
  DcnTap s;
  s.h = (float)(ho * a.sh - a.ph + i * a.dh) + (float)off[(long long)(2 * t) * HWo + p];
  s.w = (float)(wo * a.sw - a.pw + j * a.dw) + (float)off[(long long)(2 * t + 1) * HWo + p];
  s.m = a.mask ? (float)((const T*)a.mask)[((b * a.dg + dgi) * a.kk + t) * HWo + p] : 1.f;
  s.inside = s.h > -1.f && s.w > -1.f && s.h < (float)a.H && s.w < (float)a.W;
  s.hl = (int)floorf(s.h); s.wl = (int)floorf(s.w);
  s.lh = s.h - (float)s.hl; s.lw = s.w - (float)s.wl;
  const bool hlo = s.hl >= 0 && s.hl < a.H, hhi = s.hl + 1 >= 0 && s.hl + 1 < a.H;
  const bool wlo = s.wl >= 0 && s.wl < a.W, whi = s.wl + 1 >= 0 && s.wl + 1 < a.W;
  s.ok0 = hlo && wlo; s.ok1 = hlo && whi; s.ok2 = hhi && wlo; s.ok3 = hhi && whi;
  return s;
}

// pass 3a: col2im -- grad_input, fp32 atomics into an NCHW image.  One thread per (pixel, tap, channel), pixel fastest.
template <typename T>
__global__ __launch_bounds__(256) void k_dcnb_col2im(DcnBwd a) {
  const long long HWo = (long long)a.Ho * a.Wo, HW = (long long)a.H * a.W, n = a.P * a.kk * a.C;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long long)gridDim.x * 256) {
    const long long pix = idx % a.P;
    const int t = (int)((idx / a.P) % a.kk), c = (int)(idx / (a.P * a.kk));
    const long long b = pix / HWo, p = pix % HWo;
    const DcnTap s = dcnb_tap<T>(a, b, p, t, c / a.cpdg);
    if (!s.inside) continue;
    const int g = c / a.Cg, cl = c - g * a.Cg;
    const float tg = a.col[((long long)g * a.Kgp + t * a.Cg + cl) * a.Pp + pix] * s.m;
    float* gp = a.gi + (b * a.C + c) * HW + (long long)s.hl * a.W + s.wl;
    if (s.ok0) unsafeAtomicAdd(gp, (1.f - s.lh) * (1.f - s.lw) * tg);
    if (s.ok1) unsafeAtomicAdd(gp + 1, (1.f - s.lh) * s.lw * tg);
    if (s.ok2) unsafeAtomicAdd(gp + a.W, s.lh * (1.f - s.lw) * tg);
    if (s.ok3) unsafeAtomicAdd(gp + a.W + 1, s.lh * s.lw * tg);
  }
}

// pass 3a, LDS-privatised: a workgroup owns a 16 x 16 tile of OUTPUT pixels of one image for nch <= DB_CH channels of ONE deformable
// group (the sample geometry of a tap is then computed once per thread and tap).  Their samples land inside the tile's input footprint
// grown by a halo of R pixels (offsets beyond R are rare; those adds go straight to global memory), so the adds are LDS atomics into an
// E x E x nch tile.  A tap's nch column gradients are fetched together, ahead of the adds.
//
// The tile is 64-bit FIXED POINT, not fp32: tools/repro/lds_atomic_probe.hip measures ds_add_f32 at 0.33 lane-adds per clock and CU on
// MI355X (203 G/s over the chip, whatever the addresses) against 5-7 for ds_add_u64 / ds_add_u32 -- the fp32 LDS atomic alone was the
// kernel's 2.2 ms.  A first pass over the workgroup's column gradients (and masks) bounds every contribution by vmax; with
// e = 47 - ilogb(vmax) a contribution q * g scaled by 2^e is below 2^48 and, being a 24-bit float times a power of two, an exact integer
// (values below vmax * 2^-24 round to a grid of vmax * 2^-47), at most 256 * k*k <= 2^14 of them meet in one cell: no overflow, and the
// sum is exact -- grad_input no longer depends on the order of the adds (the reference's atomicAdd, .cu:322, does).
// The tile leaves the workgroup ONCE: with E <= DB_EMAX as plain coalesced fp32 stores into its own slot of `gt` -- k_dcnb_gin_out then
// sums, for every input pixel, the <= 4 tiles whose footprints cover it, in a fixed order -- otherwise (large strides / dilations) with
// global atomics on its non-zero entries.
constexpr int DB_CH = 8;
CFEN_DEV void lds_add_fixed(unsigned long long* cell, float v, int e) {
  atomicAdd(cell, (unsigned long long)(long long)__builtin_rintf(__builtin_ldexpf(v, e)));
}
template <typename T>
__global__ __launch_bounds__(256) void k_dcnb_col2im_lds(DcnBwd a, int E, int R, int nch_max, int tiles_x, int park) {
  extern __shared__ unsigned long long tile[];
  __shared__ float red[4];
  const long long HWo = (long long)a.Ho * a.Wo, HW = (long long)a.H * a.W;
  const int tid = threadIdx.x, b = blockIdx.z, c0 = blockIdx.y * nch_max, nch = min(nch_max, a.C - c0);
  const int dgi = c0 / a.cpdg;                      // host: nch_max divides cpdg, a chunk never straddles two deformable groups
  const int ho0 = (blockIdx.x / tiles_x) * 16, wo0 = (blockIdx.x % tiles_x) * 16;
  const int ry0 = ho0 * a.sh - a.ph - R, rx0 = wo0 * a.sw - a.pw - R;
  for (int i = tid; i < E * E * nch; i += 256) tile[i] = 0ull;
  const int ho = ho0 + tid / 16, wo = wo0 + tid % 16;
  const bool live = ho < a.Ho && wo < a.Wo;
  const long long p = (long long)ho * a.Wo + wo, pix = (long long)b * HWo + p;
  // ---- pass 0: bound of |mask * column gradient| over the workgroup
  float vmax = 0.f;
  bool bad = false;
  if (live) {
    for (int t = 0; t < a.kk; ++t) {
      const float m = a.mask ? fabsf((float)((const T*)a.mask)[(((long long)b * a.dg + dgi) * a.kk + t) * HWo + p]) : 1.f;
      float gl[DB_CH], gm = 0.f;
#pragma unroll
      for (int cl0 = 0; cl0 < DB_CH; ++cl0) {                 // unconditional loads (a lane past nch repeats channel c0): issued together
        const int c = c0 + (cl0 < nch ? cl0 : 0), g = c / a.Cg, cl = c - g * a.Cg;
        gl[cl0] = a.col[((long long)g * a.Kgp + t * a.Cg + cl) * a.Pp + pix];
      }
#pragma unroll
      for (int cl0 = 0; cl0 < DB_CH; ++cl0) {
        gm = fmaxf(gm, fabsf(gl[cl0]));
        bad |= !(fabsf(gl[cl0]) <= 3.0e38f);               // inf or nan (fmaxf drops a nan)
      }
      bad |= !(m <= 3.0e38f);
      vmax = fmaxf(vmax, gm * m);
    }
  }
  if (bad) vmax = __builtin_inff();
  for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
  if ((tid & 63) == 0) red[tid >> 6] = vmax;
  __syncthreads();
  vmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const bool finite = vmax > 0.f && vmax <= 3.0e38f;          // all-zero gradients: nothing to add; a non-finite one: the tile is poisoned below
  const int e = finite ? 48 - __builtin_amdgcn_frexp_expf(vmax) : 0;
  if (live && finite) {
    for (int t = 0; t < a.kk; ++t) {
      float gv[DB_CH];
#pragma unroll
      for (int cl0 = 0; cl0 < DB_CH; ++cl0) {
        const int c = c0 + (cl0 < nch ? cl0 : 0), g = c / a.Cg, cl = c - g * a.Cg;
        gv[cl0] = a.col[((long long)g * a.Kgp + t * a.Cg + cl) * a.Pp + pix];
      }
      const DcnTap s = dcnb_tap<T>(a, b, p, t, dgi);
      if (!s.inside) continue;
      const int ly = s.hl - ry0, lx = s.wl - rx0;
      const float q0 = (1.f - s.lh) * (1.f - s.lw) * s.m, q1 = (1.f - s.lh) * s.lw * s.m, q2 = s.lh * (1.f - s.lw) * s.m, q3 = s.lh * s.lw * s.m;
      const bool y0in = ly >= 0 && ly < E, y1in = ly + 1 >= 0 && ly + 1 < E, x0in = lx >= 0 && lx < E, x1in = lx + 1 >= 0 && lx + 1 < E;
      const bool l0 = s.ok0 && y0in && x0in, l1 = s.ok1 && y0in && x1in, l2 = s.ok2 && y1in && x0in, l3 = s.ok3 && y1in && x1in;
      const bool g0 = s.ok0 && !l0, g1 = s.ok1 && !l1, g2 = s.ok2 && !l2, g3 = s.ok3 && !l3;
      unsigned long long* lt = tile + ly * E + lx;
      float* gp = a.gi + ((long long)b * a.C + c0) * HW + (long long)s.hl * a.W + s.wl;
#pragma unroll
      for (int cl0 = 0; cl0 < DB_CH; ++cl0) {
        if (cl0 < nch) {
          const float tg = gv[cl0];
          if (l0) lds_add_fixed(lt, q0 * tg, e);
          if (l1) lds_add_fixed(lt + 1, q1 * tg, e);
          if (l2) lds_add_fixed(lt + E, q2 * tg, e);
          if (l3) lds_add_fixed(lt + E + 1, q3 * tg, e);
          if (g0 | g1 | g2 | g3) {
            if (g0) unsafeAtomicAdd(gp, q0 * tg);
            if (g1) unsafeAtomicAdd(gp + 1, q1 * tg);
            if (g2) unsafeAtomicAdd(gp + a.W, q2 * tg);
            if (g3) unsafeAtomicAdd(gp + a.W + 1, q3 * tg);
          }
        }
        lt += E * E;
        gp += HW;
      }
    }
  }
  __syncthreads();
  const float poison = vmax > 3.0e38f ? vmax - vmax : 0.f;     // inf - inf = nan: a non-finite gradient makes the whole tile nan, never a finite number
  if (park) {
    float* dst = a.gt + (((long long)b * gridDim.x + blockIdx.x) * a.C + c0) * (E * E);
    for (int i = tid; i < E * E * nch; i += 256) dst[i] = __builtin_ldexpf((float)(long long)tile[i], -e) + poison;
    return;
  }
  for (int i = tid; i < E * E * nch; i += 256) {
    const float v = __builtin_ldexpf((float)(long long)tile[i], -e) + poison;
    if (v == 0.f) continue;
    const int lx = i % E, ly = (i / E) % E, cl0 = i / (E * E);
    const int y = ry0 + ly, x = rx0 + lx;
    if (y >= 0 && y < a.H && x >= 0 && x < a.W) unsafeAtomicAdd(a.gi + ((long long)b * a.C + c0 + cl0) * HW + (long long)y * a.W + x, v);
  }
}

// passes 3b + 4 in one: one thread per (pixel, tap, deformable group) walks the group's channels ONCE and
//   COORD: sums d/d offset_h, d/d offset_w, d/d mask from the column gradients (the reference's col2im_coord, .cu:330-421, 689-766),
//   COL:   leaves the forward's column value (masked bilinear sample) where the column gradient was -- the k-major operand of pass 5
//          (rows at the padded pitch Kgp, zero in the padding columns pix >= P; the reference's im2col, .cu:189-242, 569-632),
// with one sample geometry and one fetch of the four corners for both.  The two corners of an image row come from ONE load of two
// adjacent elements (base clamped into the row, the wanted elements selected afterwards): two gathers per sample instead of four.
// Every load is UNCONDITIONAL at an address clamped into the tensor and the validity is applied by selects afterwards: a first version
// with `ok ? load : 0` compiled to one exec-masked branch and one s_waitcnt vmcnt(0) per load -- 72 serial memory round trips per thread.
template <typename T> struct DcnPair { T lo, hi; } __attribute__((packed, aligned(sizeof(T))));
template <typename T, bool COORD, bool COL, bool WIDE>     // WIDE: W >= 2 (a row holds a pair)
__global__ __launch_bounds__(256) void k_dcnb_sample(DcnBwd a) {
  const long long HWo = (long long)a.Ho * a.Wo, HW = (long long)a.H * a.W, n = a.Pp * a.kk * a.dg;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long long)gridDim.x * 256) {
    const long long pix = idx % a.Pp;
    const int t = (int)((idx / a.Pp) % a.kk), dgi = (int)(idx / (a.Pp * a.kk));
    const bool live = pix < a.P;
    const long long b = live ? pix / HWo : 0, p = live ? pix % HWo : 0;
    const DcnTap s = dcnb_tap<T>(a, b, p, t, dgi);
    const bool use = live && s.inside;
    const int hl = use ? s.hl : 0, wl = use ? s.wl : 0;
    const float w0 = (1.f - s.lh) * (1.f - s.lw), w1 = (1.f - s.lh) * s.lw, w2 = s.lh * (1.f - s.lw), w3 = s.lh * s.lw;
    const bool r0 = use && hl >= 0, r1 = use && hl + 1 < a.H;               // inside => -1 <= hl <= H - 1
    const int y0 = max(hl, 0), y1 = min(hl + 1, a.H - 1), xb = WIDE ? min(max(wl, 0), a.W - 2) : 0;
    // element `lo` of a pair is column xb, `hi` column xb + 1: which of them is the left (wl) / right (wl + 1) corner, if any
    const bool lo_l = wl == xb, hi_l = WIDE && wl == xb + 1, lo_r = wl + 1 == xb, hi_r = WIDE && wl == xb;
    const T* ip0 = (const T*)a.im + (b * a.C + (long long)dgi * a.cpdg) * HW + (long long)y0 * a.W + xb;
    const T* ip1 = (const T*)a.im + (b * a.C + (long long)dgi * a.cpdg) * HW + (long long)y1 * a.W + xb;
    float dH = 0.f, dW = 0.f, mv = 0.f;
    // channels in batches of four: all loads of a batch are issued before its first store (the in-place column store would otherwise
    // fence the next channel's loads behind it)
    for (int c = 0; c < a.cpdg; c += 4) {
      float lo0[4], hi0[4], lo1[4], hi1[4], gl[4];
      float* cp[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int cc = min(c + u, a.cpdg - 1), ch = dgi * a.cpdg + cc, g = ch / a.Cg, cl = ch - g * a.Cg;
        cp[u] = a.col + ((long long)g * a.Kgp + t * a.Cg + cl) * a.Pp + pix;
        if (WIDE) {
          const DcnPair<T> p0 = *reinterpret_cast<const DcnPair<T>*>(ip0 + cc * HW), p1 = *reinterpret_cast<const DcnPair<T>*>(ip1 + cc * HW);
          lo0[u] = (float)p0.lo; hi0[u] = (float)p0.hi; lo1[u] = (float)p1.lo; hi1[u] = (float)p1.hi;
        } else {
          lo0[u] = (float)ip0[cc * HW]; lo1[u] = (float)ip1[cc * HW]; hi0[u] = 0.f; hi1[u] = 0.f;
        }
        gl[u] = COORD ? *cp[u] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (c + u < a.cpdg) {
          const float v0 = r0 ? (lo_l ? lo0[u] : hi_l ? hi0[u] : 0.f) : 0.f, v1 = r0 ? (lo_r ? lo0[u] : hi_r ? hi0[u] : 0.f) : 0.f;
          const float v2 = r1 ? (lo_l ? lo1[u] : hi_l ? hi1[u] : 0.f) : 0.f, v3 = r1 ? (lo_r ? lo1[u] : hi_r ? hi1[u] : 0.f) : 0.f;
          if (COORD) {
            const float gv = use ? gl[u] : 0.f;
            dH += gv * (-(1.f - s.lw) * v0 - s.lw * v1 + (1.f - s.lw) * v2 + s.lw * v3);
            dW += gv * (-(1.f - s.lh) * v0 + (1.f - s.lh) * v1 - s.lh * v2 + s.lh * v3);
            mv += gv * (w0 * v0 + w1 * v1 + w2 * v2 + w3 * v3);
          }
          if (COL) *cp[u] = use ? (w0 * v0 + w1 * v1 + w2 * v2 + w3 * v3) * s.m : 0.f;
        }
      }
    }
    if (COORD && live) {
      if (a.goff) {
        T* go = (T*)a.goff + (b * a.dg + dgi) * 2 * a.kk * HWo;
        go[(long long)(2 * t) * HWo + p] = (T)(dH * s.m);
        go[(long long)(2 * t + 1) * HWo + p] = (T)(dW * s.m);
      }
      if (a.gmask) ((T*)a.gmask)[((b * a.dg + dgi) * a.kk + t) * HWo + p] = (T)mv;
    }
  }
}

// grad_input (NCHW, tensor type) = the fp32 image of the global adds + (E > 0) the parked tiles whose footprint covers the pixel:
// tile (ty, tx) starts at input row 16*sh*ty - ph - R, so it covers row y iff 0 <= y + ph + R - 16*sh*ty < E; summed in (ty, tx) order
template <typename T>
__global__ __launch_bounds__(256) void k_dcnb_gin_out(DcnBwd a, int E, int R, int tiles_x, int tiles_y) {
  const long long n = (long long)a.B * a.C * a.H * a.W;
  const int sy = 16 * a.sh, sx = 16 * a.sw;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float v = a.gi[i];
    if (E > 0) {
      const int x = (int)(i % a.W), y = (int)((i / a.W) % a.H), c = (int)((i / ((long long)a.W * a.H)) % a.C);
      const long long b = i / ((long long)a.W * a.H * a.C);
      const int yy = y + a.ph + R, xx = x + a.pw + R;
      const int ty1 = min(yy / sy, tiles_y - 1), ty0 = yy - E + 1 <= 0 ? 0 : (yy - E + sy) / sy;
      const int tx1 = min(xx / sx, tiles_x - 1), tx0 = xx - E + 1 <= 0 ? 0 : (xx - E + sx) / sx;
      for (int ty = ty0; ty <= ty1; ++ty)
        for (int tx = tx0; tx <= tx1; ++tx)
          v += a.gt[(((b * tiles_y + ty) * tiles_x + tx) * a.C + c) * (E * E) + (yy - ty * sy) * E + (xx - tx * sx)];
    }
    ((T*)a.gin)[i] = (T)v;
  }
}

// pass 5: gw[g][co][k] += scale * sum_pix gc[g*Cog + co][pix] * col[g][k][pix]  (col rows at the pitch Kgp per group).  grid (pixel ranges, groups of 4 k tiles, conv groups);
// wave w of a workgroup owns k tile 4*blockIdx.y + w and every co tile (<= 8 tiles: Cout / group <= 128).
constexpr int DB_MAXOT = 8;
__global__ __launch_bounds__(256) void k_dcnb_weight(DcnBwd a, int chunks_per_block) {
  typedef Mma<float>::frag frag;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, h = lane >> 4;
  const int g = blockIdx.z, kt = blockIdx.y * 4 + wave;
  if (kt * 16 >= a.Kg) return;
  const int not_ = (a.Cog + 15) / 16;
  const long long nchunks = a.Pp / 16;
  const long long c0 = (long long)blockIdx.x * chunks_per_block, c1 = min(nchunks, c0 + chunks_per_block);
  floatx4 acc[DB_MAXOT];
#pragma unroll
  for (int i = 0; i < DB_MAXOT; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
  const float* brow = a.col + ((long long)g * a.Kgp + min(kt * 16 + r16, a.Kg - 1)) * a.Pp + 4 * h;
  const float* arow[DB_MAXOT];
#pragma unroll
  for (int i = 0; i < DB_MAXOT; ++i) arow[i] = a.gc + ((long long)g * a.Cog + min(i * 16 + r16, a.Cog - 1)) * a.Pp + 4 * h;
  for (long long ch = c0; ch < c1; ++ch) {
    const frag bf = *reinterpret_cast<const frag*>(brow + ch * 16);
#pragma unroll
    for (int i = 0; i < DB_MAXOT; ++i)
      if (i < not_) acc[i] = Mma<float>::mma(*reinterpret_cast<const frag*>(arow[i] + ch * 16), bf, acc[i]);
  }
  // lane owns A rows (co) 4h .. 4h+3 of tile i for B row (k) r16
  const int k = kt * 16 + r16;
  if (k < a.Kg) {
#pragma unroll
    for (int i = 0; i < DB_MAXOT; ++i)
      if (i < not_) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = i * 16 + 4 * h + r;
          if (co < a.Cog) unsafeAtomicAdd(a.gw + ((long long)g * a.Cog + co) * a.Kg + k, a.scale * acc[i][r]);
        }
      }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_dcnb_weight_out(DcnBwd a) {   // grad_weight[co][c][t] += gw[g][col][t*Cg + c]
  const long long n = (long long)a.Cout * a.Cg * a.kk;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int t = (int)(i % a.kk), c = (int)((i / a.kk) % a.Cg), co = (int)(i / ((long long)a.kk * a.Cg));
    const int g = co / a.Cog, col = co % a.Cog;
    T* dst = (T*)a.gweight + i;
    *dst = (T)((float)*dst + a.gw[((long long)g * a.Cog + col) * a.Kg + t * a.Cg + c]);
  }
}

// grad_bias[co] += sum over all pixels of grad_out[co]: DB_BIAS_PARTS workgroups per channel leave partial sums, one thread per
// channel adds them in order (deterministic; one workgroup per channel took 0.5 ms over 524288 pixels)
__global__ __launch_bounds__(256) void k_dcnb_bias(DcnBwd a) {
  __shared__ float red[256];
  const int co = blockIdx.y, part = blockIdx.x;
  const long long per = (a.P + DB_BIAS_PARTS - 1) / DB_BIAS_PARTS, p0 = part * per, p1 = min(a.P, p0 + per);
  float s = 0.f;
  for (long long p = p0 + threadIdx.x; p < p1; p += 256) s += a.gc[(long long)co * a.Pp + p];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) a.gb[co * DB_BIAS_PARTS + part] = red[0];
}

template <typename T>
__global__ __launch_bounds__(256) void k_dcnb_bias_out(DcnBwd a) {
  const int co = blockIdx.x * 256 + threadIdx.x;
  if (co >= a.Cout) return;
  float s = 0.f;
  for (int i = 0; i < DB_BIAS_PARTS; ++i) s += a.gb[co * DB_BIAS_PARTS + i];
  ((T*)a.gbias)[co] = (T)((float)((T*)a.gbias)[co] + s);
}

int& cfen_dcn_bwd_lds() {   // 1 (default): LDS-privatised col2im, tiles parked and summed; 2: tiles flushed with global atomics; 0: every add a global atomic (A/B, tests)
  static int v = 1;
  return v;
}

unsigned grid_for(long long n) { return (unsigned)std::min<long long>((n + 255) / 256, 16384); }

template <typename T>
int run_dcn_backward(DcnBwd a, hipStream_t s) {
  const bool want_in = a.gin || a.goff || a.gmask;
  const long long nin = (long long)a.B * a.C * a.H * a.W;
  const long long n2 = a.Pp * a.group * a.Cogp, n3 = (long long)a.Cout * a.Pp, n4 = (long long)a.group * a.Kgp * a.Cogp;
  CFEN_LAUNCH(k_dcnb_prep<T>, dim3(grid_for(n2 + n3 + n4)), dim3(256), 0, s, a, n2, n3, n4);
  CFEN_CHECK_LAUNCH("deform_conv backward (layout pre-pass)");
  if (want_in) {
    if (a.gin && hipMemsetAsync(a.gi, 0, (size_t)nin * 4, s) != hipSuccess) { cfen_set_error("deform_conv backward: memset failed"); return CFEN_ERR_HIP; }
    for (int g = 0; g < a.group; ++g) {   // gcol[g][k][pix] = wt[g][k][:] . gn[pix][g][:]   ("tokens" = the Kgp rows of W^T, "features" = pixels)
      int rc = cfen_gemm_impl(0, a.wt + (size_t)g * a.Kgp * a.Cogp, a.Cogp, a.gn + (size_t)g * a.Cogp, a.group * a.Cogp, nullptr, nullptr, 0, nullptr, 0,
                              a.col + (size_t)g * a.Kgp * a.Pp, (int)a.Pp, a.Kgp, (int)a.Pp, a.Cogp, 0, s);
      if (rc) return rc;
    }
    if (a.gin) {
      // LDS tile: footprint of 16 output pixels per axis + halo R; CH channels as far as 64 KB go
      const int foot = 15 * std::max(a.sh, a.sw) + (std::max(a.kh, a.kw) - 1) * std::max(a.dh, a.dw) + 2;
      const int R = foot <= 20 ? (32 - foot) / 2 : 4, E = foot + 2 * R;
      int CH = DB_CH;                                   // channels per tile: the largest divisor of a deformable group's channels <= DB_CH ...
      while (CH > 1 && (a.cpdg % CH || (size_t)E * E * CH * 8 > 64 * 1024 - 64)) --CH;   // ... that fits 64 KB of LDS (8-byte cells)
      const int tiles_x = (a.Wo + 15) / 16, tiles_y = (a.Ho + 15) / 16;
      int parked = 0;
      if (cfen_dcn_bwd_lds() && (size_t)E * E * CH * 8 <= 64 * 1024 - 64 && a.C / CH <= 65535 && a.B <= 65535) {
        parked = E <= DB_EMAX && cfen_dcn_bwd_lds() == 1;
        CFEN_LAUNCH(k_dcnb_col2im_lds<T>, dim3((unsigned)(tiles_x * tiles_y), (unsigned)(a.C / CH), (unsigned)a.B), dim3(256),
                    (size_t)E * E * CH * 8, s, a, E, R, CH, tiles_x, parked);
      } else {
        CFEN_LAUNCH(k_dcnb_col2im<T>, dim3(grid_for(a.P * a.kk * a.C)), dim3(256), 0, s, a);
      }
      CFEN_CHECK_LAUNCH("deform_conv backward (col2im)");
      CFEN_LAUNCH(k_dcnb_gin_out<T>, dim3(grid_for(nin)), dim3(256), 0, s, a, parked ? E : 0, R, tiles_x, tiles_y);
      CFEN_CHECK_LAUNCH("deform_conv backward (grad_input)");
    }
  }
  // the sampling pass runs AFTER col2im: with a weight gradient wanted it overwrites the column gradients with the forward's columns
  const bool coord = a.goff || a.gmask;
  if (coord || a.gweight) {
    const dim3 grid(grid_for(a.Pp * a.kk * a.dg));
    const bool wide = a.W >= 2;
    if (coord && a.gweight) { if (wide) CFEN_LAUNCH((k_dcnb_sample<T, true, true, true>), grid, dim3(256), 0, s, a); else CFEN_LAUNCH((k_dcnb_sample<T, true, true, false>), grid, dim3(256), 0, s, a); }
    else if (coord) { if (wide) CFEN_LAUNCH((k_dcnb_sample<T, true, false, true>), grid, dim3(256), 0, s, a); else CFEN_LAUNCH((k_dcnb_sample<T, true, false, false>), grid, dim3(256), 0, s, a); }
    else { if (wide) CFEN_LAUNCH((k_dcnb_sample<T, false, true, true>), grid, dim3(256), 0, s, a); else CFEN_LAUNCH((k_dcnb_sample<T, false, true, false>), grid, dim3(256), 0, s, a); }
    CFEN_CHECK_LAUNCH("deform_conv backward (col2im_coord / im2col)");
  }
  if (a.gweight) {
    if (hipMemsetAsync(a.gw, 0, (size_t)a.group * a.Cog * a.Kg * 4, s) != hipSuccess) { cfen_set_error("deform_conv backward: memset failed"); return CFEN_ERR_HIP; }
    const long long nchunks = a.Pp / 16;
    const int kgroups = (a.Kg + 63) / 64;
    long long ranges = std::max<long long>(1, std::min<long long>(nchunks, 2048 / std::max(1, kgroups * a.group)));
    const int cpb = (int)((nchunks + ranges - 1) / ranges);
    ranges = (nchunks + cpb - 1) / cpb;
    CFEN_LAUNCH(k_dcnb_weight, dim3((unsigned)ranges, (unsigned)kgroups, (unsigned)a.group), dim3(256), 0, s, a, cpb);
    CFEN_CHECK_LAUNCH("deform_conv backward (grad_weight)");
    CFEN_LAUNCH(k_dcnb_weight_out<T>, dim3(grid_for((long long)a.Cout * a.Cg * a.kk)), dim3(256), 0, s, a);
    CFEN_CHECK_LAUNCH("deform_conv backward (grad_weight out)");
  }
  if (a.gbias) {
    CFEN_LAUNCH(k_dcnb_bias, dim3(DB_BIAS_PARTS, (unsigned)a.Cout), dim3(256), 0, s, a);
    CFEN_LAUNCH(k_dcnb_bias_out<T>, dim3((unsigned)(a.Cout + 255) / 256), dim3(256), 0, s, a);
    CFEN_CHECK_LAUNCH("deform_conv backward (grad_bias)");
  }
  return CFEN_OK;
}

int dcn_backward(int dtype, DcnBwd a, void* columns, size_t columns_bytes, hipStream_t s) {
  CFEN_CHECK_ARG(dtype == 0 || dtype == 1, "deform_conv backward: dtype %d unsupported (fp32, fp16)", dtype);
  CFEN_CHECK_ARG(a.im && a.offset && a.weight && a.gout, "deform_conv backward: null tensor");
  CFEN_CHECK_ARG(a.kw > 0 && a.kh > 0, "kernel size should be greater than zero, but got kH: %d kW: %d", a.kh, a.kw);
  CFEN_CHECK_ARG(a.sw > 0 && a.sh > 0, "stride should be greater than zero, but got dH: %d dW: %d", a.sh, a.sw);
  CFEN_CHECK_ARG(a.dw > 0 && a.dh > 0, "dilation should be greater than 0, but got dilationH: %d dilationW: %d", a.dh, a.dw);
  CFEN_CHECK_ARG(a.B > 0 && a.C > 0 && a.Cout > 0 && a.group > 0 && a.dg > 0, "deform_conv backward: empty problem");
  CFEN_CHECK_ARG(a.C % a.group == 0 && a.Cout % a.group == 0, "deform_conv backward: channels must be divisible by groups");
  CFEN_CHECK_ARG(a.C % a.dg == 0, "input channels must divide deformable group size");
  a.Ho = (a.H + 2 * a.ph - (a.dh * (a.kh - 1) + 1)) / a.sh + 1;
  a.Wo = (a.W + 2 * a.pw - (a.dw * (a.kw - 1) + 1)) / a.sw + 1;
  CFEN_CHECK_ARG(a.Ho >= 1 && a.Wo >= 1, "deform_conv backward: output size is too small");
  a.Cg = a.C / a.group; a.Cog = a.Cout / a.group; a.Cogp = (a.Cog + 3) / 4 * 4; a.kk = a.kh * a.kw;
  a.Kg = a.Cg * a.kk; a.Kgp = (a.Kg + 3) / 4 * 4; a.cpdg = a.C / a.dg;
  a.P = (long long)a.B * a.Ho * a.Wo; a.Pp = (a.P + 15) / 16 * 16;
  CFEN_CHECK_ARG(a.Cog <= 16 * DB_MAXOT, "deform_conv backward: more than %d output channels per group", 16 * DB_MAXOT);
  CFEN_CHECK_ARG(a.Pp < (1ll << 31) && (long long)a.group * a.Kgp < (1ll << 31), "deform_conv backward: problem too large");
  const DcnBwdLayout l = dcnb_layout(a.B, a.C, a.H, a.W, a.Cout, a.kk, a.Ho, a.Wo, a.group);
  CFEN_CHECK_ARG(columns && cfen_aligned16(columns) && columns_bytes >= l.total,
                 "deform_conv backward: `columns` scratch of at least %zu bytes (cfen_deform_conv_backward_bytes), 16-byte aligned, is required", l.total);
  unsigned char* base = (unsigned char*)columns;
  a.gn = (float*)(base + l.gn); a.gc = (float*)(base + l.gc); a.wt = (float*)(base + l.wt);
  a.col = (float*)(base + l.col); a.gi = (float*)(base + l.gi); a.gw = (float*)(base + l.gw);
  a.gt = (float*)(base + l.gt); a.gb = (float*)(base + l.gb);
  return dtype == 1 ? run_dcn_backward<half_t>(a, s) : run_dcn_backward<float>(a, s);
}

}  // namespace

extern "C" {

int cfen_deform_conv_backward_set_lds(int enabled) {   // A/B switch of the LDS-privatised col2im (tests, tools/bench_dcn.py); returns the old value
  const int old = cfen_dcn_bwd_lds();
  cfen_dcn_bwd_lds() = enabled;
  return old;
}

size_t cfen_deform_conv_backward_bytes(int B, int Cin, int H, int W, int Cout, int kH, int kW, int Hout, int Wout, int group) {
  if (B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || kH <= 0 || kW <= 0 || Hout <= 0 || Wout <= 0 || group <= 0 || Cin % group || Cout % group) return 0;
  return dcnb_layout(B, Cin, H, W, Cout, kH * kW, Hout, Wout, group).total;
}

int cfen_deform_conv_backward_input(int dtype, const void* input, const void* offset, const void* gradOutput, void* gradInput, void* gradOffset,
                                    const void* weight, int B, int Cin, int H, int W, int Cout, int kW, int kH, int dW, int dH, int padW,
                                    int padH, int dilationW, int dilationH, int group, int deformable_group, int im2col_step, void* columns,
                                    size_t columns_bytes, void* stream) {
  CFEN_CHECK_ARG(im2col_step > 0 && B % (im2col_step < B ? im2col_step : B) == 0, "im2col step must divide batchsize");
  CFEN_CHECK_ARG(gradInput && gradOffset, "deform_conv_backward_input: gradInput and gradOffset are required");
  DcnBwd a{};
  a.im = input; a.offset = offset; a.weight = weight; a.gout = gradOutput; a.gin = gradInput; a.goff = gradOffset;
  a.B = B; a.C = Cin; a.H = H; a.W = W; a.Cout = Cout; a.kh = kH; a.kw = kW; a.sh = dH; a.sw = dW; a.ph = padH; a.pw = padW;
  a.dh = dilationH; a.dw = dilationW; a.group = group; a.dg = deformable_group; a.scale = 1.f;
  return dcn_backward(dtype, a, columns, columns_bytes, (hipStream_t)stream);
}

int cfen_deform_conv_backward_parameters(int dtype, const void* input, const void* offset, const void* gradOutput, void* gradWeight, int B,
                                         int Cin, int H, int W, int Cout, int kW, int kH, int dW, int dH, int padW, int padH, int dilationW,
                                         int dilationH, int group, int deformable_group, float scale, int im2col_step, void* columns,
                                         size_t columns_bytes, void* stream) {
  CFEN_CHECK_ARG(im2col_step > 0 && B % (im2col_step < B ? im2col_step : B) == 0, "im2col step must divide batchsize");
  CFEN_CHECK_ARG(gradWeight, "deform_conv_backward_parameters: gradWeight is required");
  DcnBwd a{};
  a.im = input; a.offset = offset; a.weight = gradWeight /* shapes only: the pre-pass reads it, pass 5 does not use wt */; a.gout = gradOutput;
  a.gweight = gradWeight;
  a.B = B; a.C = Cin; a.H = H; a.W = W; a.Cout = Cout; a.kh = kH; a.kw = kW; a.sh = dH; a.sw = dW; a.ph = padH; a.pw = padW;
  a.dh = dilationH; a.dw = dilationW; a.group = group; a.dg = deformable_group; a.scale = scale;
  return dcn_backward(dtype, a, columns, columns_bytes, (hipStream_t)stream);
}

int cfen_modulated_deform_conv_backward(int dtype, const void* input, const void* weight, const void* bias, const void* offset, const void* mask,
                                        void* grad_input, void* grad_weight, void* grad_bias, void* grad_offset, void* grad_mask,
                                        const void* grad_output, int B, int Cin, int H, int W, int Cout, int kernel_h, int kernel_w, int stride_h,
                                        int stride_w, int pad_h, int pad_w, int dilation_h, int dilation_w, int group, int deformable_group,
                                        int with_bias, void* columns, size_t columns_bytes, void* stream) {
  (void)bias;
  CFEN_CHECK_ARG(mask != nullptr, "modulated_deform_conv_backward: mask is required");
  CFEN_CHECK_ARG(grad_input && grad_weight && grad_offset && grad_mask, "modulated_deform_conv_backward: gradient tensors are required");
  CFEN_CHECK_ARG(!with_bias || grad_bias, "modulated_deform_conv_backward: with_bias set but grad_bias is null");
  DcnBwd a{};
  a.im = input; a.offset = offset; a.mask = mask; a.weight = weight; a.gout = grad_output;
  a.gin = grad_input; a.goff = grad_offset; a.gmask = grad_mask; a.gweight = grad_weight; a.gbias = with_bias ? grad_bias : nullptr;
  a.B = B; a.C = Cin; a.H = H; a.W = W; a.Cout = Cout; a.kh = kernel_h; a.kw = kernel_w; a.sh = stride_h; a.sw = stride_w;
  a.ph = pad_h; a.pw = pad_w; a.dh = dilation_h; a.dw = dilation_w; a.group = group; a.dg = deformable_group; a.scale = 1.f;
  return dcn_backward(dtype, a, columns, columns_bytes, (hipStream_t)stream);
}

}  // extern "C"
