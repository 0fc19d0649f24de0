// Stand-alone attempt to reproduce the defect DESIGN.md (round 1) attributed to packed-fp32 VALU instructions on MI355X:
// "the HIGH lane of v_pk_fma_f32 / v_pk_mul_f32 returns wrong values in a wave whose CU is shared with MFMA-heavy waves of ANOTHER
// kernel".  No buffer is shared between the two kernels; each checks its own arithmetic.
//
//   kernel A (victim):  every thread runs a chain of v_pk_fma_f32 (inline asm, both halves used) and the SAME chain with two scalar
//                       v_fma_f32 per step; any bit difference between the packed and the scalar result is counted.
//   kernel B (noise):   waves spinning on v_mfma_f32_16x16x32_f16 with operands in registers, launched on a second stream so that
//                       its waves share CUs with kernel A's (low per-block resources on both sides, grids of several waves per SIMD).
//
// Build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_fma_repro tools/repro/pk_fma_repro.hip && /tmp/pk_fma_repro
// Output: mismatch counts for A alone, A beside B (same process, two streams), for a few occupancies.  Exit code 0 always;
// tools/repro_pk_fma.py wraps it and records the result under profiles/.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(4))) float floatx4;
typedef __attribute__((ext_vector_type(2))) float float2_t;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

__global__ __launch_bounds__(256) void victim(unsigned long long* mismatches, float* sink, int iters, float seed) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  float2_t acc = {seed + tid * 1e-3f, seed - tid * 2e-3f};
  float s0 = acc[0], s1 = acc[1];
  float2_t m = {1.0000001f, 0.9999999f}, a = {1e-4f, -1e-4f};
  unsigned long long bad = 0;
  for (int i = 0; i < iters; ++i) {
    // packed: acc = acc * m + a on both halves in ONE instruction
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(acc) : "v"(acc), "v"(m), "v"(a));
    // scalar twin, one v_fma_f32 per half (asm so that the compiler cannot re-pack it)
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s0) : "v"(s0), "v"(m[0]), "v"(a[0]));
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s1) : "v"(s1), "v"(m[1]), "v"(a[1]));
    if ((i & 63) == 63) {
      bad += (__float_as_uint(acc[0]) != __float_as_uint(s0)) + (__float_as_uint(acc[1]) != __float_as_uint(s1));
      // a second flavour: packed multiply with swapped halves selected through op_sel must equal the scalar products
      float2_t p;
      asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "v"(acc), "v"(m));
      float q0, q1;
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(q0) : "v"(acc[0]), "v"(m[0]));
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(q1) : "v"(acc[1]), "v"(m[1]));
      bad += (__float_as_uint(p[0]) != __float_as_uint(q0)) + (__float_as_uint(p[1]) != __float_as_uint(q1));
    }
  }
  if (bad) atomicAdd(mismatches, bad);
  if (acc[0] == 123.456f) sink[tid] = acc[1] + s0 + s1;   // keep everything live
}

__global__ __launch_bounds__(256) void mfma_noise(float* sink, int iters) {
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
  floatx4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, b, c3, 0, 0, 0);
  }
  if (c0[0] + c1[1] + c2[2] + c3[3] == 123.456f) sink[blockIdx.x * blockDim.x + threadIdx.x] = c0[0];
}

int main() {
  unsigned long long* mm;
  float* sink;
  CHECK(hipMalloc(&mm, sizeof(*mm)));
  CHECK(hipMalloc(&sink, 1 << 24));
  hipStream_t sa, sb;
  CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  const int victim_blocks[] = {256, 1024, 4096};
  unsigned long long total_alone = 0, total_beside = 0;
  for (int rep = 0; rep < 20; ++rep) {
    for (int vb : victim_blocks) {
      unsigned long long h = 0;
      CHECK(hipMemsetAsync(mm, 0, sizeof(*mm), sa));
      hipLaunchKernelGGL(victim, dim3(vb), dim3(256), 0, sa, mm, sink, 4096, 0.5f + rep);
      CHECK(hipStreamSynchronize(sa));
      CHECK(hipMemcpy(&h, mm, sizeof(h), hipMemcpyDeviceToHost));
      total_alone += h;
      CHECK(hipMemsetAsync(mm, 0, sizeof(*mm), sa));
      CHECK(hipStreamSynchronize(sa));
      hipLaunchKernelGGL(mfma_noise, dim3(1024), dim3(256), 0, sb, sink + (1 << 20), 60000);   // ~ms of MFMA on every SIMD
      hipLaunchKernelGGL(victim, dim3(vb), dim3(256), 0, sa, mm, sink, 4096, 0.5f + rep);
      hipLaunchKernelGGL(victim, dim3(vb), dim3(256), 0, sa, mm, sink, 4096, 1.5f + rep);
      CHECK(hipStreamSynchronize(sa));
      CHECK(hipStreamSynchronize(sb));
      CHECK(hipMemcpy(&h, mm, sizeof(h), hipMemcpyDeviceToHost));
      total_beside += h;
    }
  }
  printf("{\"packed_vs_scalar_mismatches_alone\": %llu, \"packed_vs_scalar_mismatches_beside_mfma_kernel\": %llu, \"victim_launches\": %d}\n",
         total_alone, total_beside, 20 * 3 * 3);
  return 0;
}
