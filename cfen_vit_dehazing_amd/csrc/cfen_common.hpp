// Shared device helpers for the CFEN-ViT gfx950 kernels.
//
// Every contraction in this library (token GEMMs, attention, implicit-GEMM convolutions, the
// deformable-conv GEMM) is built on ONE wave-level primitive, `Mma<T>`: a 16x16 output tile
// accumulated in fp32 from 16-byte per-lane operand fragments.
//
//   T = _Float16 : v_mfma_f32_16x16x32_f16   one MFMA per 32-deep chunk   (fp16 storage, fp32 accumulate)
//   T = float    : v_mfma_f32_16x16x4_f32 x4 four MFMAs per 16-deep chunk (exact fp32 fmaf chain)
//
// Fragment contract (lane l, r16 = l & 15, h = l >> 4, EPL = 16 / sizeof(T) elements per lane):
//   A fragment: A[row = r16][k = chunk*KC + h*EPL + j], j = 0..EPL-1   (16 contiguous bytes)
//   B fragment: B[k = chunk*KC + h*EPL + j][col = r16]                 (16 contiguous bytes, k-major source)
//   acc[r]    : C[row = 4*h + r][col = r16]
// For fp32 the four MFMAs use element j of both fragments in step j, i.e. the k order inside a
// chunk is permuted identically for A and B, which leaves the sum unchanged.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

typedef _Float16 half_t;
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(4))) _Float16 half4;
typedef __attribute__((ext_vector_type(4))) float floatx4;

#define CFEN_DEV __device__ __forceinline__

template <typename T> struct Mma;

template <> struct Mma<half_t> {
  static constexpr int KC = 32;   // k elements covered by one fragment chunk
  static constexpr int EPL = 8;   // elements per lane per fragment
  typedef half8 frag;
  typedef half4 out4;
  static CFEN_DEV frag zero() { frag z; for (int i = 0; i < 8; ++i) z[i] = (half_t)0; return z; }
  static CFEN_DEV floatx4 mma(frag a, frag b, floatx4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

template <> struct Mma<float> {
  static constexpr int KC = 16;
  static constexpr int EPL = 4;
  typedef floatx4 frag;
  typedef floatx4 out4;
  static CFEN_DEV frag zero() { frag z = {0.f, 0.f, 0.f, 0.f}; return z; }
  static CFEN_DEV floatx4 mma(frag a, frag b, floatx4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
    return c;
  }
};

template <typename T> CFEN_DEV typename Mma<T>::frag load_frag(const T* p) {
  return *reinterpret_cast<const typename Mma<T>::frag*>(p);
}

// 4 consecutive elements <-> fp32x4
template <typename T> CFEN_DEV floatx4 load4(const T* p);
template <> CFEN_DEV floatx4 load4<float>(const float* p) { return *reinterpret_cast<const floatx4*>(p); }
template <> CFEN_DEV floatx4 load4<half_t>(const half_t* p) {
  half4 v = *reinterpret_cast<const half4*>(p);
  floatx4 r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  return r;
}
template <typename T> CFEN_DEV void store4(T* p, floatx4 v);
template <> CFEN_DEV void store4<float>(float* p, floatx4 v) { *reinterpret_cast<floatx4*>(p) = v; }
template <> CFEN_DEV void store4<half_t>(half_t* p, floatx4 v) {
  half4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
  *reinterpret_cast<half4*>(p) = o;
}

// one 16-byte vector of T <-> fp32 registers (EPL values)
template <typename T> struct Vec16;
template <> struct Vec16<float> {
  static constexpr int N = 4;
  static CFEN_DEV void load(const float* p, float* o) {
    floatx4 v = *reinterpret_cast<const floatx4*>(p);
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
  }
  static CFEN_DEV void store(float* p, const float* o) {
    floatx4 v = {o[0], o[1], o[2], o[3]};
    *reinterpret_cast<floatx4*>(p) = v;
  }
};
template <> struct Vec16<half_t> {
  static constexpr int N = 8;
  static CFEN_DEV void load(const half_t* p, float* o) {
    half8 v = *reinterpret_cast<const half8*>(p);
    for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
  }
  static CFEN_DEV void store(half_t* p, const float* o) {
    half8 v;
    for (int i = 0; i < 8; ++i) v[i] = (half_t)o[i];
    *reinterpret_cast<half8*>(p) = v;
  }
};

// Cross-lane reductions stay in the VALU (DPP row operations + v_permlane16_swap / v_permlane32_swap) instead of
// ds_bpermute (__shfl*): no LDS-pipe round trip, and every lane of a group ends with the bitwise identical total.
// Sum / max over the 4 lanes {l, l^16, l^32, l^48} (the lanes that hold one MFMA column):
CFEN_DEV float col_sum(float v) {
  auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float a = __uint_as_float(q[0]) + __uint_as_float(q[1]);
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(a), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
CFEN_DEV float col_max(float v) {
  auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float a = fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(a), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// Two adjacent 16-feature tiles of the same 16 tokens -> ONE 16-byte store per lane (round 5).  A lane (token r16, row h of 16 lanes) holds features 4h .. 4h+3 of each
// tile; v_permlane16_swap exchanges the odd rows of tile A with the even rows of tile B, after which row 0 holds A's features 0..7, row 1 B's 0..7, row 2 A's 8..15,
// row 3 B's 8..15: every lane stores 8 consecutive features (feature offset from A's first: 16 (h & 1) + 8 (h >> 1)), a store instruction covers 64 bytes per token
// instead of 32, and there are half as many store instructions.  Pure data movement: the stored bits are those of the two 8-byte stores.
CFEN_DEV uint4 pair_tiles16(floatx4 a, floatx4 b) {
  const half4 ha = {(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3]}, hb = {(half_t)b[0], (half_t)b[1], (half_t)b[2], (half_t)b[3]};
  const uint2 ua = __builtin_bit_cast(uint2, ha), ub = __builtin_bit_cast(uint2, hb);
  auto lo = __builtin_amdgcn_permlane16_swap(ua.x, ub.x, false, false);
  auto hi = __builtin_amdgcn_permlane16_swap(ua.y, ub.y, false, false);
  return uint4{lo[0], hi[0], lo[1], hi[1]};
}

// Reductions inside a 16-lane row with DPP (quad_perm xor1 / xor2, row_half_mirror, row_mirror): pure VALU.
// Every lane of the row ends with the bitwise identical total.
template <int CTRL> CFEN_DEV float dpp_mov(float v) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xf, 0xf, false));
}
CFEN_DEV float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  return v;
}
// Sum over a group of G = 16, 32 or 64 consecutive lanes (group-aligned).
template <int G> CFEN_DEV float group_sum(float v) {
  v = row16_sum(v);
  if (G >= 32) {
    auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(q[0]) + __uint_as_float(q[1]);
  }
  if (G >= 64) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  return v;
}

// Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  Launch a 1-D grid of cfen_grid8(n) blocks and
// map the hardware block id to a logical one so that every XCD works on ONE contiguous range of logical blocks:
// neighbours that share cache lines (conv halo rows, the heads of one attention window) then share an L2.
CFEN_DEV unsigned xcd_chunked_block(unsigned bid, unsigned nblocks8) { return (bid & 7u) * (nblocks8 >> 3) + (bid >> 3); }
static inline unsigned cfen_grid8(long long n) { return (unsigned)((n + 7) / 8 * 8); }

// Grouped launches: up to CFEN_MAX_GROUPS independent problems of identical geometry (the R, S and D decoders run the
// same layer on different maps with different weights) go out as ONE launch, blockIdx.z selecting the problem: three
// times the workgroups per launch for the latency-bound small kernels, a third of the launches.
constexpr int CFEN_MAX_GROUPS = 3;
template <class A> struct Grouped { A g[CFEN_MAX_GROUPS]; };
template <class P> struct PtrG { P* p[CFEN_MAX_GROUPS]; };

// ---- host side ------------------------------------------------------------------------------
#define CFEN_OK 0
#define CFEN_ERR_ARG (-1)
#define CFEN_ERR_HIP (-2)
#define CFEN_ERR_STATE (-3)

void cfen_set_error(const char* fmt, ...);

#define CFEN_CHECK_ARG(cond, ...)                 \
  do {                                            \
    if (!(cond)) {                                \
      cfen_set_error(__VA_ARGS__);                \
      return CFEN_ERR_ARG;                        \
    }                                             \
  } while (0)

// ---- kernel launch: eager on a stream, or recorded as a node of an explicitly built hipGraph -------------
// The multi-stream plan (GViT beside LViT, S decoder beside R decoder) is turned into a graph by ADDING KERNEL
// NODES WITH EXPLICIT DEPENDENCIES, not by stream capture: hipStreamEndCapture of ROCm 7.2 recurses without bound
// on repeated fork/join between the same streams.  While a recorder is installed, "streams" are only lane ids.
#include <map>
#include <tuple>
#include <utility>
#include <vector>

struct CfenGraphRecorder {
  hipGraph_t graph = nullptr;
  std::map<hipStream_t, std::vector<hipGraphNode_t>> tail;   // nodes the next launch on a lane must follow
  size_t nodes = 0;
};
CfenGraphRecorder*& cfen_recorder();      // thread-local, null = launch eagerly
hipError_t& cfen_last_launch();           // thread-local status of the latest CFEN_LAUNCH

template <typename... KArgs, size_t... I>
static inline hipError_t cfen_add_node(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t smem, hipStream_t lane,
                                       std::tuple<KArgs...>& packed, std::index_sequence<I...>) {
  CfenGraphRecorder* rec = cfen_recorder();
  void* params[] = {(void*)&std::get<I>(packed)...};
  hipKernelNodeParams p;
  memset(&p, 0, sizeof(p));
  p.func = (void*)kernel; p.gridDim = grid; p.blockDim = block; p.sharedMemBytes = (unsigned)smem; p.kernelParams = params;
  std::vector<hipGraphNode_t>& deps = rec->tail[lane];
  hipGraphNode_t node;
  hipError_t e = hipGraphAddKernelNode(&node, rec->graph, deps.empty() ? nullptr : deps.data(), deps.size(), &p);
  if (e == hipSuccess) {
    deps.assign(1, node);
    ++rec->nodes;
  }
  return e;
}

template <typename... KArgs, typename... Args>
static inline hipError_t cfen_launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t smem, hipStream_t s, Args&&... args) {
  if (!cfen_recorder()) {
    hipLaunchKernelGGL(kernel, grid, block, smem, s, std::forward<Args>(args)...);
    return hipGetLastError();
  }
  std::tuple<KArgs...> packed(std::forward<Args>(args)...);
  return cfen_add_node(kernel, grid, block, smem, s, packed, std::index_sequence_for<KArgs...>{});
}

// every launch leaves the kernel's name in a thread-local log: cfen_net_profile attributes its per-launch times to DEVICE KERNELS with it (bench.py `by_symbol`), not to
// whatever the host code calls the step.  The name is the INSTANTIATION (round 6: "k_mlp3<24, 2, 3, 1536, 0, 1, 4, 0, 1, 0>", what rocprofv3 prints): the runtime's own
// record of the host stub -> device symbol registration (hipKernelNameRefByPtr), demangled, cached per stub.  Through round 5 it was the template expression as written at
// the launch site ("k_mlp3<ND, TM, R, ...>", still the fallback), which merged the D = 192 and D = 384 kernels into one `by_symbol` row.
#include <string>
std::string& cfen_kernel_log();
void cfen_log_kernel(const void* host_stub, const char* as_written);     // cfen_api.cpp
#define CFEN_LAUNCH(kernel, grid, block, smem, stream, ...) \
  (cfen_log_kernel((const void*)(kernel), #kernel), cfen_last_launch() = cfen_launch(kernel, grid, block, smem, stream, __VA_ARGS__))

#define CFEN_CHECK_LAUNCH(what)                                                    \
  do {                                                                             \
    hipError_t e__ = cfen_last_launch();                                           \
    if (e__ != hipSuccess) {                                                       \
      cfen_set_error("%s: launch failed: %s", what, hipGetErrorString(e__));       \
      return CFEN_ERR_HIP;                                                         \
    }                                                                              \
  } while (0)

// "has this launcher raised its kernel's dynamic-LDS limit on the CURRENT device yet?"  Per device: a process that drives several GPUs (one dec_ipt per device) must raise
// it on each; `flags` is the launcher's own static bool[64].
static inline bool cfen_first_use_on_device(bool (&flags)[64]) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  const bool first = !flags[dev];
  flags[dev] = true;
  return first;
}

static inline bool cfen_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
