// LDS atomic throughput probe (MI355X): ds_add_f32 / ds_add_u32 / ds_add_rtn / plain read-modify-write, on random and on lane-linear
// addresses of a 31 x 31 x 8 fp32 tile -- the access pattern of k_dcnb_col2im_lds.   hipcc --offload-arch=gfx950 -O2 -o probe lds_atomic_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int N = 31 * 31 * 8, ITER = 1024;
template <int MODE, int RANDOM>
__global__ __launch_bounds__(256) void k(float* out, unsigned seed) {
  __shared__ float tile[N];
  for (int i = threadIdx.x; i < N; i += 256) tile[i] = 0.f;
  __syncthreads();
  unsigned h = seed + threadIdx.x * 2654435761u + blockIdx.x * 97u;
  float acc = 0.f;
  for (int i = 0; i < ITER; ++i) {
    h = h * 1664525u + 1013904223u;
    const int a = RANDOM ? (int)((h >> 8) % N) : (int)((threadIdx.x + i * 37) % N);
    const float v = (float)(h & 255) * 0.01f;
    if (MODE == 0) atomicAdd(&tile[a], v);                                                   // ds_add_f32
    else if (MODE == 1) atomicAdd((unsigned*)&tile[a], (unsigned)(h & 255));                 // ds_add_u32
    else if (MODE == 2) acc += atomicAdd(&tile[a], v);                                       // ds_add_rtn_f32
    else if (MODE == 3) tile[a] += v;                                                        // racy read-modify-write (throughput only)
    else if (MODE == 4) acc += tile[a];                                                      // plain read
    else if (MODE == 5) atomicAdd((unsigned long long*)&tile[a & ~1], (unsigned long long)(h & 255));   // ds_add_u64
  }
  __syncthreads();
  float s = acc;
  for (int i = threadIdx.x; i < N; i += 256) s += tile[i];
  if (s == 12345.678f) out[0] = s;
}
template <int MODE, int RANDOM>
int run(const char* name, float* out) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int blocks = 256 * 5 * 4;
  k<MODE, RANDOM><<<blocks, 256>>>(out, 1);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  k<MODE, RANDOM><<<blocks, 256>>>(out, 2);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  const double ops = (double)blocks * 256 * ITER;
  printf("{\"op\": \"%s\", \"addresses\": \"%s\", \"ms\": %.3f, \"G_lane_ops_per_s\": %.1f, \"lane_ops_per_clk_per_CU\": %.2f}\n", name,
         RANDOM ? "random" : "linear", ms, ops / ms * 1e-6, ops / (ms * 1e-3) / 256 / 2.4e9);
  return 0;
}
int main() {
  float* out;
  CK(hipMalloc(&out, 64));
  if (run<0, 1>("ds_add_f32", out) || run<0, 0>("ds_add_f32", out) || run<1, 1>("ds_add_u32", out) || run<1, 0>("ds_add_u32", out) ||
      run<2, 1>("ds_add_rtn_f32", out) || run<3, 1>("read+write", out) || run<3, 0>("read+write", out) || run<4, 1>("read", out) ||
      run<4, 0>("read", out) || run<5, 1>("ds_add_u64", out)) return 1;
  return 0;
}
