// Does a hipGraph kernel node accept more than 64 KiB of LDS on gfx950 / ROCm 7.2?  (round 1 saw hipGraphInstantiate fail for > 64 KiB of
// DYNAMIC LDS and kept every kernel <= 64 KiB.)  Probes static and dynamic LDS of 96 / 128 / 160 KiB, eager and as explicit kernel nodes.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/lds_probe tools/repro/lds_graph_probe.hip && /tmp/lds_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

template <int KB>
__global__ __launch_bounds__(256) void k_static(float* out) {
  __shared__ float buf[KB * 256];
  for (int i = threadIdx.x; i < KB * 256; i += 256) buf[i] = (float)i;
  __syncthreads();
  float s = 0.f;
  for (int i = threadIdx.x; i < KB * 256; i += 256) s += buf[(i * 7) % (KB * 256)];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_dynamic(float* out, int n) {
  extern __shared__ float dbuf[];
  for (int i = threadIdx.x; i < n; i += 256) dbuf[i] = (float)i;
  __syncthreads();
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += dbuf[(i * 7) % n];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

static const char* ok(hipError_t e) { return e == hipSuccess ? "ok" : hipGetErrorString(e); }

template <typename K>
static void probe(const char* what, K kernel, size_t dyn, float* out, int n) {
  hipError_t e1 = hipSuccess;
  if (dyn) e1 = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
  void* args_s[] = {&out};
  void* args_d[] = {&out, &n};
  hipLaunchKernelGGL(kernel, dim3(64), dim3(256), dyn, 0, out, n);
  hipError_t e2 = hipDeviceSynchronize();
  hipError_t e2b = hipGetLastError();
  hipGraph_t g; hipGraphExec_t ex = nullptr;
  hipGraphCreate(&g, 0);
  hipKernelNodeParams p; memset(&p, 0, sizeof(p));
  p.func = (void*)kernel; p.gridDim = dim3(64); p.blockDim = dim3(256); p.sharedMemBytes = (unsigned)dyn; p.kernelParams = args_d;
  (void)args_s;
  hipGraphNode_t node;
  hipError_t e3 = hipGraphAddKernelNode(&node, g, nullptr, 0, &p);
  hipError_t e4 = e3 == hipSuccess ? hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) : e3;
  hipError_t e5 = e4 == hipSuccess ? hipGraphLaunch(ex, 0) : e4;
  hipError_t e6 = hipDeviceSynchronize();
  printf("%-28s attr=%s eager=%s/%s addnode=%s instantiate=%s launch=%s sync=%s\n", what, ok(e1), ok(e2), ok(e2b), ok(e3), ok(e4), ok(e5), ok(e6));
  if (ex) hipGraphExecDestroy(ex);
  hipGraphDestroy(g);
  (void)hipGetLastError();
}

// static kernels take (float*) only: wrap with the same (float*, int) signature for the shared probe
template <int KB> __global__ __launch_bounds__(256) void k_static2(float* out, int) {
  __shared__ float buf[KB * 256];
  for (int i = threadIdx.x; i < KB * 256; i += 256) buf[i] = (float)i;
  __syncthreads();
  float s = 0.f;
  for (int i = threadIdx.x; i < KB * 256; i += 256) s += buf[(i * 7) % (KB * 256)];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float* out;
  hipMalloc(&out, 64 * 256 * 4);
  probe("static 64 KiB", k_static2<64>, 0, out, 0);
  probe("static 96 KiB", k_static2<96>, 0, out, 0);
  probe("static 128 KiB", k_static2<128>, 0, out, 0);
  probe("static 160 KiB", k_static2<160>, 0, out, 0);
  probe("dynamic 64 KiB", k_dynamic, 64 * 1024, out, 64 * 256);
  probe("dynamic 96 KiB", k_dynamic, 96 * 1024, out, 96 * 256);
  probe("dynamic 128 KiB", k_dynamic, 128 * 1024, out, 128 * 256);
  probe("dynamic 160 KiB", k_dynamic, 160 * 1024, out, 160 * 256);
  return 0;
}
