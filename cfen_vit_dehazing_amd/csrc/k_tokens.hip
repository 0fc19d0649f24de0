// Data-movement kernels between NHWC feature maps and token matrices.
//
// Feature maps live in HBM as NHWC with a channel stride `cs` >= C (12-channel maps are stored with
// cs = 16 so every pixel is a whole number of 16-byte vectors).  Token rows use the feature order
// (i, j, c) -- the four (LViT) or sixteen (GViT) pixels of a patch laid end to end -- instead of the
// reference's F.unfold order (c, i, j) (v3:1140); the host permutes every D-sized weight axis once at
// load time (packing.py), so a token row is p*p contiguous C-vectors of the map: pure 16-byte copies.
//
//   patchify   : Crop2x2 nesting (v3:403-428) + F.unfold (v3:1140), optionally fused with GViT's
//                avgpool2(avgpool2(x)) (v3:1274) as one 4x4 mean
//   unpatchify : F.fold (v3:1186) + Join2x2 (v3:1046-1056)
//   upsample4  : GViT's upsam(upsam(x)) (v3:1323): two successive bilinear x2, align_corners=False
//   nchw_to_nhwc: network input (B,3,H,W) fp32 -> NHWC T, channels zero-padded to `cs`
#include "cfen_common.hpp"

namespace {

struct TokGeom {
  int B, H, W;     // (pooled) map size the tokens tile
  int C, cs;       // channels / channel stride of the map in HBM
  int ws, p;       // window edge (pooled pixels), patch edge
  int pool;        // 1 or 4: map in HBM is pool x larger than (H, W)
};

template <typename T>
__global__ __launch_bounds__(256) void k_patchify(PtrG<const T> fmapg, PtrG<T> tokg, TokGeom g, long long nvec) {
  const T* __restrict__ fmap = fmapg.p[blockIdx.z];
  T* __restrict__ tok = tokg.p[blockIdx.z];
  constexpr int EPL = Vec16<T>::N;
  const int cv = g.C / EPL;               // vectors per pixel
  const int D = g.p * g.p * g.C;
  const int tw = g.ws / g.p;              // tokens per window row
  const int S = tw * tw;
  const int nwx = g.W / g.ws, nwy = g.H / g.ws;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < nvec; idx += (long long)gridDim.x * 256) {
    int v = (int)(idx % (D / EPL));
    long long m = idx / (D / EPL);
    int c = (v % cv) * EPL;
    int ij = v / cv;
    int i = ij / g.p, j = ij % g.p;
    int t = (int)(m % S);
    long long wi = m / S;
    int wx = (int)(wi % nwx);
    int wy = (int)((wi / nwx) % nwy);
    int b = (int)(wi / ((long long)nwx * nwy));
    int y = wy * g.ws + (t / tw) * g.p + i;
    int x = wx * g.ws + (t % tw) * g.p + j;
    float o[EPL];
    if (g.pool == 1) {
      Vec16<T>::load(fmap + (((size_t)b * g.H + y) * g.W + x) * g.cs + c, o);
    } else if (g.pool == 4) {
      // GViT's avgpool . avgpool (v3:1274): all 16 loads of the 4 x 4 mean are issued before the first add (with the run-time loop bounds
      // of the general branch they went out one dependent round trip at a time: 11-21 us per launch for a 25 MB read)
      const size_t HW = (size_t)g.W * 4;
      const T* src = fmap + (((size_t)b * g.H * 4 + (size_t)y * 4) * HW + (size_t)x * 4) * g.cs + c;
      typename Mma<T>::frag q[16];
#pragma unroll
      for (int dy = 0; dy < 4; ++dy)
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) q[dy * 4 + dx] = load_frag<T>(src + ((size_t)dy * HW + dx) * g.cs);
#pragma unroll
      for (int e = 0; e < EPL; ++e) o[e] = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k)
#pragma unroll
        for (int e = 0; e < EPL; ++e) o[e] += (float)q[k][e];
#pragma unroll
      for (int e = 0; e < EPL; ++e) o[e] *= 1.f / 16.f;
    } else {
      const int HW = g.W * g.pool;
#pragma unroll
      for (int e = 0; e < EPL; ++e) o[e] = 0.f;
      for (int dy = 0; dy < g.pool; ++dy)
        for (int dx = 0; dx < g.pool; ++dx) {
          float t8[EPL];
          Vec16<T>::load(fmap + (((size_t)b * g.H * g.pool + y * g.pool + dy) * HW + x * g.pool + dx) * g.cs + c, t8);
#pragma unroll
          for (int e = 0; e < EPL; ++e) o[e] += t8[e];
        }
      const float inv = 1.f / (float)(g.pool * g.pool);
#pragma unroll
      for (int e = 0; e < EPL; ++e) o[e] *= inv;
    }
    Vec16<T>::store(tok + m * D + v * EPL, o);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_unpatchify(PtrG<const T> tokg, PtrG<T> fmapg, TokGeom g, long long nvec) {
  const T* __restrict__ tok = tokg.p[blockIdx.z];
  T* __restrict__ fmap = fmapg.p[blockIdx.z];
  constexpr int EPL = Vec16<T>::N;
  const int cv = g.C / EPL;
  const int D = g.p * g.p * g.C;
  const int tw = g.ws / g.p;
  const int S = tw * tw;
  const int nwx = g.W / g.ws, nwy = g.H / g.ws;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < nvec; idx += (long long)gridDim.x * 256) {
    int v = (int)(idx % (D / EPL));
    long long m = idx / (D / EPL);
    int c = (v % cv) * EPL;
    int ij = v / cv;
    int i = ij / g.p, j = ij % g.p;
    int t = (int)(m % S);
    long long wi = m / S;
    int wx = (int)(wi % nwx);
    int wy = (int)((wi / nwx) % nwy);
    int b = (int)(wi / ((long long)nwx * nwy));
    int y = wy * g.ws + (t / tw) * g.p + i;
    int x = wx * g.ws + (t % tw) * g.p + j;
    typedef typename Mma<T>::frag vec;
    *reinterpret_cast<vec*>(fmap + (((size_t)b * g.H + y) * g.W + x) * g.cs + c) =
        *reinterpret_cast<const vec*>(tok + m * D + v * EPL);
  }
}

// Two successive bilinear x2 upsamples (align_corners=False, v3:1323) composed into ONE 3-tap filter per axis: output
// 4k + r reads inputs k-1, k, k+1 (indices clamped, which reproduces both levels of edge clamping) with weights
//   r = 0: .375 .625 0 | r = 1: .1875 .75 .0625 | r = 2: .0625 .75 .1875 | r = 3: 0 .625 .375
// (0.25 / 0.75 products, exact in binary).  9 loads and ~80 FMAs per 16-byte output vector instead of 16 loads and ~450
// VALU operations of the two-step evaluation: the kernel was VALU bound.
// One thread = one INPUT pixel x one 16-byte channel vector: it loads the 3 x 3 neighbourhood once (9 loads) and writes the 4 x 4 output
// pixels that neighbourhood determines.  (One thread per OUTPUT vector re-loaded the neighbourhood for each of the 16: 9 loads per 16-byte
// store, L1-bandwidth bound at 2.4x the time of the stores.)  Per output the sums run in the same order as before: bitwise the same values.
template <typename T>
__global__ __launch_bounds__(256) void k_upsample4(PtrG<const T> smallg, PtrG<T> outg, int B, int h, int w, int C,
                                                   int cs_in, int cs_out, long long nvec) {
  // One thread = (input pixel, output column rx of its 4 x 4 block, channel vector), channel vector fastest, then rx: the lanes of a wave
  // write (4 kx + rx) * cs + c = CONSECUTIVE bytes of an output row (round 2 had a thread write its whole 4 x 4 block: 16-byte pieces 4
  // pixels apart, a quarter of every line per store instruction; the map writes are what this kernel waits for).  The 3 x 3 neighbourhood is
  // loaded by the four rx threads of a pixel (same addresses: L1), the arithmetic per output value and its order are unchanged.
  const T* __restrict__ small = smallg.p[blockIdx.z];
  T* __restrict__ out = outg.p[blockIdx.z];
  constexpr int EPL = Vec16<T>::N;
  const int cv = C / EPL;
  const int H = 4 * h, W = 4 * w;
  const float w0 = 0.375f, w1 = 0.1875f, w2 = 0.0625f;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < nvec; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % cv) * EPL;
    const long long t = idx / cv;
    const int rx = (int)(t & 3);
    const long long pix = t >> 2;
    const int kx = (int)(pix % w), ky = (int)((pix / w) % h), b = (int)(pix / ((long long)w * h));
    const int xs[3] = {max(kx - 1, 0), kx, min(kx + 1, w - 1)};
    const int ys[3] = {max(ky - 1, 0), ky, min(ky + 1, h - 1)};
    const T* base = small + (size_t)b * h * w * cs_in + c;
    float p[3][3][EPL];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int bb = 0; bb < 3; ++bb) Vec16<T>::load(base + ((size_t)ys[a] * w + xs[bb]) * cs_in, p[a][bb]);
    const float wx[3] = {rx == 0 ? w0 : rx == 1 ? w1 : rx == 2 ? w2 : 0.f, rx == 0 || rx == 3 ? 0.625f : 0.75f, rx == 3 ? w0 : rx == 2 ? w1 : rx == 1 ? w2 : 0.f};
    float row[3][EPL];     // horizontal pass for output column 4 kx + rx, one row of the neighbourhood at a time
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int e = 0; e < EPL; ++e) row[a][e] = 0.f;
#pragma unroll
      for (int bb = 0; bb < 3; ++bb)
#pragma unroll
        for (int e = 0; e < EPL; ++e) row[a][e] += wx[bb] * p[a][bb][e];      // (a zero weight adds +0: the value a two-tap sum has)
    }
#pragma unroll
    for (int ry = 0; ry < 4; ++ry) {
      const float wy[3] = {ry == 0 ? w0 : ry == 1 ? w1 : ry == 2 ? w2 : 0.f, ry == 0 || ry == 3 ? 0.625f : 0.75f, ry == 3 ? w0 : ry == 2 ? w1 : ry == 1 ? w2 : 0.f};
      float acc[EPL];
#pragma unroll
      for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        if (wy[a] == 0.f) continue;       // compile-time after unrolling
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] += wy[a] * row[a][e];
      }
      Vec16<T>::store(out + (((size_t)b * H + 4 * ky + ry) * W + 4 * kx + rx) * cs_out + c, acc);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_nchw_to_nhwc(const float* __restrict__ in, T* __restrict__ out, int B, int C, int H, int W,
                                                      int cs, long long npix) {
  constexpr int EPL = Vec16<T>::N;
  for (long long pix = (long long)blockIdx.x * 256 + threadIdx.x; pix < npix; pix += (long long)gridDim.x * 256) {
    long long b = pix / ((long long)H * W), r = pix % ((long long)H * W);
    for (int c0 = 0; c0 < cs; c0 += EPL) {
      float o[EPL];
#pragma unroll
      for (int e = 0; e < EPL; ++e) o[e] = (c0 + e < C) ? in[(b * C + c0 + e) * (long long)H * W + r] : 0.f;
      Vec16<T>::store(out + pix * cs + c0, o);
    }
  }
}

// (x + 1) / 2 * 255 then a truncating cast, no clamp, 1-channel tensors tiled to 3: util/util.py:12-24 on device.
// in: (C,H,W) fp32, out: (H,W,3) uint8.  The reference computes in float32 (numpy keeps the tensor's dtype) -- so do we.
__global__ __launch_bounds__(256) void k_tensor2im_u8(const float* __restrict__ in, unsigned char* __restrict__ out, int C, long long npix) {
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long long)gridDim.x * 256) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = in[(C == 1 ? 0 : c) * npix + p];
      out[p * 3 + c] = (unsigned char)(int)((v + 1.f) / 2.0f * 255.0f);
    }
  }
}

// ToTensor + Normalize((0.5,)*3, (0.5,)*3) (data/base_dataset.py:44-46) fused with the NHWC layout change:
// in: (B,H,W,3) uint8, out: NHWC T with channel stride cs, value v / 255 * 2 - 1 evaluated as ((v / 255) - 0.5) / 0.5 in fp32.
template <typename T>
__global__ __launch_bounds__(256) void k_u8hwc_to_nhwc(const unsigned char* __restrict__ in, T* __restrict__ out, int cs, long long npix) {
  constexpr int EPL = Vec16<T>::N;
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long long)gridDim.x * 256) {
    for (int c0 = 0; c0 < cs; c0 += EPL) {
      float o[EPL];
#pragma unroll
      for (int e = 0; e < EPL; ++e) o[e] = (c0 + e < 3) ? ((float)in[p * 3 + c0 + e] / 255.0f - 0.5f) / 0.5f : 0.f;
      Vec16<T>::store(out + p * cs + c0, o);
    }
  }
}

inline unsigned grid_for(long long n) {
  long long g = (n + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

template <typename T>
int check_geom(const TokGeom& g, const char* what) {
  constexpr int EPL = Vec16<T>::N;
  CFEN_CHECK_ARG(g.B > 0 && g.H > 0 && g.W > 0 && g.C > 0, "%s: empty problem", what);
  CFEN_CHECK_ARG(g.C % EPL == 0 && g.cs % EPL == 0 && g.cs >= g.C, "%s: C=%d cs=%d must be multiples of %d", what, g.C, g.cs, EPL);
  CFEN_CHECK_ARG(g.ws > 0 && g.p > 0 && g.ws % g.p == 0 && g.H % g.ws == 0 && g.W % g.ws == 0,
                 "%s: window %d / patch %d do not tile the %dx%d map", what, g.ws, g.p, g.H, g.W);
  CFEN_CHECK_ARG(g.pool == 1 || g.pool == 4, "%s: pool must be 1 or 4", what);
  return CFEN_OK;
}

template <typename T>
int run_patchify(int ng, const void* const* fmap, void* const* tok, TokGeom g, int inverse, hipStream_t s) {
  int rc = check_geom<T>(g, inverse ? "unpatchify" : "patchify");
  if (rc) return rc;
  CFEN_CHECK_ARG(ng >= 1 && ng <= CFEN_MAX_GROUPS, "patchify: 1..%d problems per launch", CFEN_MAX_GROUPS);
  CFEN_CHECK_ARG(!(inverse && g.pool != 1), "unpatchify: pool must be 1");
  PtrG<const T> src{};
  PtrG<T> dst{};
  for (int k = 0; k < ng; ++k) {
    CFEN_CHECK_ARG(fmap[k] && tok[k] && cfen_aligned16(fmap[k]) && cfen_aligned16(tok[k]), "patchify: pointers must be non-null and 16-byte aligned");
    src.p[k] = (const T*)(inverse ? tok[k] : fmap[k]);
    dst.p[k] = (T*)(inverse ? const_cast<void*>(fmap[k]) : tok[k]);
  }
  const long long nvec = (long long)g.B * g.H * g.W * g.C / Vec16<T>::N;
  if (inverse)
    CFEN_LAUNCH(k_unpatchify<T>, dim3(grid_for(nvec), 1, ng), dim3(256), 0, s, src, dst, g, nvec);
  else
    CFEN_LAUNCH(k_patchify<T>, dim3(grid_for(nvec), 1, ng), dim3(256), 0, s, src, dst, g, nvec);
  CFEN_CHECK_LAUNCH("patchify");
  return CFEN_OK;
}

template <typename T>
int run_upsample4(int ng, const void* const* small, void* const* out, int B, int h, int w, int C, int cs_in, int cs_out, hipStream_t s) {
  constexpr int EPL = Vec16<T>::N;
  CFEN_CHECK_ARG(B > 0 && h > 0 && w > 0 && C > 0, "upsample4: empty problem");
  CFEN_CHECK_ARG(ng >= 1 && ng <= CFEN_MAX_GROUPS, "upsample4: 1..%d problems per launch", CFEN_MAX_GROUPS);
  CFEN_CHECK_ARG(C % EPL == 0 && cs_in % EPL == 0 && cs_out % EPL == 0, "upsample4: channels must be multiples of %d", EPL);
  PtrG<const T> src{};
  PtrG<T> dst{};
  for (int k = 0; k < ng; ++k) {
    CFEN_CHECK_ARG(small[k] && out[k] && cfen_aligned16(small[k]) && cfen_aligned16(out[k]), "upsample4: pointers must be non-null and 16-byte aligned");
    src.p[k] = (const T*)small[k]; dst.p[k] = (T*)out[k];
  }
  const long long nvec = (long long)B * h * w * 4 * (C / EPL);   // one thread per input pixel, output column of its 4 x 4 block and channel vector
  CFEN_LAUNCH(k_upsample4<T>, dim3(grid_for(nvec), 1, ng), dim3(256), 0, s, src, dst, B, h, w, C, cs_in, cs_out, nvec);
  CFEN_CHECK_LAUNCH("upsample4");
  return CFEN_OK;
}


// GViT's avgpool2(avgpool2(x)) (v3:1274) as a MAP: out[b][y][x][c] = mean of the 4 x 4 block of `in` -- what k_patchify(pool = 4) computes, kept in NHWC
// so that the fragment-stream front half (k_front3) can gather its 4 x 4 patches from it like an LViT block does from the level's map
template <typename T>
__global__ __launch_bounds__(256) void k_pool4(PtrG<const T> ing, PtrG<T> outg, int B, int h, int w, int C, int cs_in, int cs_out, long long nvec) {
  const T* __restrict__ in = ing.p[blockIdx.z];
  T* __restrict__ out = outg.p[blockIdx.z];
  constexpr int EPL = Vec16<T>::N;
  const int cv = C / EPL;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < nvec; idx += (long long)gridDim.x * 256) {
    const int v = (int)(idx % cv);
    const long long px = idx / cv;
    const int x = (int)(px % w), y = (int)((px / w) % h), b = (int)(px / ((long long)w * h));
    const size_t W4 = (size_t)w * 4;
    const T* src = in + (((size_t)b * h * 4 + (size_t)y * 4) * W4 + (size_t)x * 4) * cs_in + v * EPL;
    typename Mma<T>::frag q[16];
#pragma unroll
    for (int dy = 0; dy < 4; ++dy)
#pragma unroll
      for (int dx = 0; dx < 4; ++dx) q[dy * 4 + dx] = load_frag<T>(src + ((size_t)dy * W4 + dx) * cs_in);
    float o[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) o[e] = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k)            // the summation order of k_patchify's pooled branch: the two paths produce the same tokens bit for bit
#pragma unroll
      for (int e = 0; e < EPL; ++e) o[e] += (float)q[k][e];
#pragma unroll
    for (int e = 0; e < EPL; ++e) o[e] *= 1.f / 16.f;
    Vec16<T>::store(out + (size_t)px * cs_out + v * EPL, o);
  }
}

// What-if probe ("net.gvit_dummy_*"): hold `gridDim.x` CUs for `ticks` / 100 us the way a persistent GViT block kernel would (one 512-thread
// workgroup with 100 KB of LDS per CU), optionally streaming 16-byte loads meanwhile.  Timing experiments only: writes nothing but `sink`.
__global__ __launch_bounds__(512) void k_occupy(const uint4* __restrict__ src, size_t nvec, unsigned ticks, int do_stream, unsigned* sink) {
  __shared__ unsigned char lds[100 * 1024];
  lds[threadIdx.x] = (unsigned char)threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned acc = lds[(threadIdx.x * 7) & 1023];
  size_t i = ((size_t)blockIdx.x * 977 + blockIdx.y * 131) * 512 + threadIdx.x;
  for (int it = 0; it < (1 << 20); ++it) {
    if (__builtin_amdgcn_s_memrealtime() - t0 >= ticks) break;
    if (do_stream) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint4 v = src[(i + (size_t)u * 512) % nvec];
        acc += v.x ^ v.y ^ v.z ^ v.w;
      }
      i += 8 * 512 * 61;
    } else {
      __builtin_amdgcn_s_sleep(32);
    }
  }
  if (acc == 0x12345677u) sink[0] = acc;
}

}  // namespace

// the same with the footprint of a k_gemm_dma workgroup (256 threads, 36 KB of LDS: four fit a CU, and the dispatcher spreads them over every CU)
__global__ __launch_bounds__(256) void k_occupy_small(const uint4* __restrict__ src, size_t nvec, unsigned ticks, int do_stream, unsigned* sink) {
  __shared__ unsigned char lds[36 * 1024];
  lds[threadIdx.x] = (unsigned char)threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned acc = lds[(threadIdx.x * 7) & 1023];
  for (int it = 0; it < (1 << 20); ++it) {
    if (__builtin_amdgcn_s_memrealtime() - t0 >= ticks) break;
    __builtin_amdgcn_s_sleep(32);
  }
  if (acc == 0x12345677u) sink[0] = acc;
}

int cfen_pool4_impl_g(int dtype, int ng, const void* const* in, void* const* out, int B, int h, int w, int C, int cs_in, int cs_out, hipStream_t s) {
  CFEN_CHECK_ARG(ng >= 1 && ng <= CFEN_MAX_GROUPS && B > 0 && h > 0 && w > 0 && C > 0, "pool4: bad arguments");
  const int epl = dtype == 1 ? 8 : 4;
  CFEN_CHECK_ARG((dtype == 0 || dtype == 1) && C % epl == 0 && cs_in % epl == 0 && cs_out % epl == 0 && cs_in >= C && cs_out >= C, "pool4: channels must be multiples of %d", epl);
  const long long nvec = (long long)B * h * w * (C / epl);
  for (int k = 0; k < ng; ++k) CFEN_CHECK_ARG(in[k] && out[k] && cfen_aligned16(in[k]) && cfen_aligned16(out[k]), "pool4: null or misaligned pointer");
  if (dtype == 1) {
    PtrG<const half_t> a{}; PtrG<half_t> o{};
    for (int k = 0; k < ng; ++k) { a.p[k] = (const half_t*)in[k]; o.p[k] = (half_t*)out[k]; }
    CFEN_LAUNCH(k_pool4<half_t>, dim3(grid_for(nvec), 1, ng), dim3(256), 0, s, a, o, B, h, w, C, cs_in, cs_out, nvec);
  } else {
    PtrG<const float> a{}; PtrG<float> o{};
    for (int k = 0; k < ng; ++k) { a.p[k] = (const float*)in[k]; o.p[k] = (float*)out[k]; }
    CFEN_LAUNCH(k_pool4<float>, dim3(grid_for(nvec), 1, ng), dim3(256), 0, s, a, o, B, h, w, C, cs_in, cs_out, nvec);
  }
  CFEN_CHECK_LAUNCH("pool4");
  return CFEN_OK;
}

int cfen_occupy_impl(int wgs, int ng, int usec, int do_stream, const void* src, size_t src_bytes, void* sink, hipStream_t s) {
  CFEN_CHECK_ARG(wgs != 0 && wgs <= 256 && wgs >= -2048 && ng >= 1 && ng <= 3 && usec > 0 && usec <= 2000 && src && sink && src_bytes >= (1u << 20), "occupy: bad arguments");
  if (wgs < 0)      // negative: that many SMALL workgroups (k_gemm_dma's footprint)
    CFEN_LAUNCH(k_occupy_small, dim3(-wgs, ng), dim3(256), 0, s, (const uint4*)src, src_bytes / 16, (unsigned)usec * 100u, do_stream, (unsigned*)sink);
  else
    CFEN_LAUNCH(k_occupy, dim3(wgs, ng), dim3(512), 0, s, (const uint4*)src, src_bytes / 16, (unsigned)usec * 100u, do_stream, (unsigned*)sink);
  CFEN_CHECK_LAUNCH("occupy");
  return CFEN_OK;
}

// inverse: fmap[] are the maps written from tok[]
int cfen_patchify_impl_g(int dtype, int ng, const void* const* fmap, void* const* tok, int B, int H, int W, int C, int cs, int ws, int p, int pool,
                         int inverse, hipStream_t s) {
  TokGeom g{B, H, W, C, cs, ws, p, pool};
  if (dtype == 1) return run_patchify<half_t>(ng, fmap, tok, g, inverse, s);
  if (dtype == 0) return run_patchify<float>(ng, fmap, tok, g, inverse, s);
  cfen_set_error("patchify: unknown dtype %d", dtype);
  return CFEN_ERR_ARG;
}
int cfen_patchify_impl(int dtype, const void* fmap, void* tok, int B, int H, int W, int C, int cs, int ws, int p, int pool,
                       int inverse, hipStream_t s) {
  return cfen_patchify_impl_g(dtype, 1, &fmap, &tok, B, H, W, C, cs, ws, p, pool, inverse, s);
}

int cfen_upsample4_impl_g(int dtype, int ng, const void* const* small, void* const* out, int B, int h, int w, int C, int cs_in, int cs_out,
                          hipStream_t s) {
  if (dtype == 1) return run_upsample4<half_t>(ng, small, out, B, h, w, C, cs_in, cs_out, s);
  if (dtype == 0) return run_upsample4<float>(ng, small, out, B, h, w, C, cs_in, cs_out, s);
  cfen_set_error("upsample4: unknown dtype %d", dtype);
  return CFEN_ERR_ARG;
}
int cfen_upsample4_impl(int dtype, const void* small, void* out, int B, int h, int w, int C, int cs_in, int cs_out, hipStream_t s) {
  return cfen_upsample4_impl_g(dtype, 1, &small, &out, B, h, w, C, cs_in, cs_out, s);
}

// Zero a few KB of synchronisation words (arrival counters, barrier words) as a KERNEL: inside a recorded launch plan this is a kernel node like
// its neighbours (see cfen_zero_async in cfen_api.cpp for why it is not a memset node).
__global__ __launch_bounds__(256) void k_zero_words(unsigned* __restrict__ p, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = 0u;
}
int cfen_zero_words_impl(void* p, size_t nwords, hipStream_t s) {
  CFEN_LAUNCH(k_zero_words, dim3(grid_for((long long)nwords)), dim3(256), 0, s, (unsigned*)p, (long long)nwords);
  CFEN_CHECK_LAUNCH("zero_words");
  return CFEN_OK;
}

int cfen_tensor2im_u8_impl(const float* in, unsigned char* out, int C, int H, int W, hipStream_t s) {
  CFEN_CHECK_ARG(in && out && (C == 1 || C == 3) && H > 0 && W > 0, "tensor2im_u8: needs a (1|3,H,W) fp32 tensor");
  const long long npix = (long long)H * W;
  CFEN_LAUNCH(k_tensor2im_u8, dim3(grid_for(npix)), dim3(256), 0, s, in, out, C, npix);
  CFEN_CHECK_LAUNCH("tensor2im_u8");
  return CFEN_OK;
}

int cfen_u8hwc_to_nhwc_impl(int dtype, const unsigned char* in, void* out, int B, int H, int W, int cs, hipStream_t s) {
  CFEN_CHECK_ARG(in && out && B > 0 && H > 0 && W > 0 && cs >= 3 && cfen_aligned16(out), "u8hwc_to_nhwc: bad arguments");
  const long long npix = (long long)B * H * W;
  if (dtype == 1) {
    CFEN_CHECK_ARG(cs % 8 == 0, "u8hwc_to_nhwc: cs must be a multiple of 8");
    CFEN_LAUNCH(k_u8hwc_to_nhwc<half_t>, dim3(grid_for(npix)), dim3(256), 0, s, in, (half_t*)out, cs, npix);
  } else if (dtype == 0) {
    CFEN_CHECK_ARG(cs % 4 == 0, "u8hwc_to_nhwc: cs must be a multiple of 4");
    CFEN_LAUNCH(k_u8hwc_to_nhwc<float>, dim3(grid_for(npix)), dim3(256), 0, s, in, (float*)out, cs, npix);
  } else {
    cfen_set_error("u8hwc_to_nhwc: unknown dtype %d", dtype);
    return CFEN_ERR_ARG;
  }
  CFEN_CHECK_LAUNCH("u8hwc_to_nhwc");
  return CFEN_OK;
}

int cfen_nchw_to_nhwc_impl(int dtype, const float* in, void* out, int B, int C, int H, int W, int cs, hipStream_t s) {
  CFEN_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0 && cs >= C, "nchw_to_nhwc: bad shape");
  CFEN_CHECK_ARG(cfen_aligned16(out), "nchw_to_nhwc: output must be 16-byte aligned");
  long long npix = (long long)B * H * W;
  if (dtype == 1) {
    CFEN_CHECK_ARG(cs % 8 == 0, "nchw_to_nhwc: cs must be a multiple of 8");
    CFEN_LAUNCH(k_nchw_to_nhwc<half_t>, dim3(grid_for(npix)), dim3(256), 0, s, in, (half_t*)out, B, C, H, W, cs, npix);
  } else if (dtype == 0) {
    CFEN_CHECK_ARG(cs % 4 == 0, "nchw_to_nhwc: cs must be a multiple of 4");
    CFEN_LAUNCH(k_nchw_to_nhwc<float>, dim3(grid_for(npix)), dim3(256), 0, s, in, (float*)out, B, C, H, W, cs, npix);
  } else {
    cfen_set_error("nchw_to_nhwc: unknown dtype %d", dtype);
    return CFEN_ERR_ARG;
  }
  CFEN_CHECK_LAUNCH("nchw_to_nhwc");
  return CFEN_OK;
}
