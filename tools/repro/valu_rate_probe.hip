// Issue rate of a few VALU instructions on one wave per SIMD (gfx950): cycles per wave-instruction from s_memtime around a loop of independent ops.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rate_probe tools/repro/valu_rate_probe.hip && /tmp/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define OPS8(STMT) STMT(0) STMT(1) STMT(2) STMT(3) STMT(4) STMT(5) STMT(6) STMT(7)

template <int KIND>
__global__ __launch_bounds__(256) void probe(float* out, unsigned long long* cyc, int iters) {
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = out[threadIdx.x + i * 256] * 0.001f - 1.0f;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#define S_EXP32(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
#define S_EXP16(i) asm volatile("v_exp_f16 %0, %0" : "+v"(v[i]));
#define S_FMA32(i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[i]));
#define S_PKFMA16(i) asm volatile("v_pk_fma_f16 %0, %0, %0, %0" : "+v"(v[i]));
#define S_CVT(i) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(v[i]));
#define S_MAX3(i) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(v[i]));
#define S_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
#define S_PKFMA32(i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(*(double*)&v[i & 6]));
#define S_LDEXP(i) asm volatile("v_ldexp_f32 %0, %0, 1" : "+v"(v[i]));
    if (KIND == 0) { OPS8(S_EXP32) }
    if (KIND == 1) { OPS8(S_EXP16) }
    if (KIND == 2) { OPS8(S_FMA32) }
    if (KIND == 3) { OPS8(S_PKFMA16) }
    if (KIND == 4) { OPS8(S_CVT) }
    if (KIND == 5) { OPS8(S_MAX3) }
    if (KIND == 6) { OPS8(S_RCP) }
    if (KIND == 7) { OPS8(S_LDEXP) }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i];
  out[threadIdx.x + blockIdx.x * 256] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[KIND] = t1 - t0;
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 64);
  hipMemset(out, 0, 1 << 22); hipMemset(cyc, 0, 64);
  const int iters = 4096;
  const char* names[8] = {"v_exp_f32", "v_exp_f16", "v_fma_f32", "v_pk_fma_f16", "v_cvt_f16_f32", "v_max3_f32", "v_rcp_f32", "v_ldexp_f32"};
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int waves = 1; waves <= 2; ++waves) {       // 1 or 2 waves per SIMD
    for (int k = 0; k < 8; ++k) {
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        switch (k) {
          case 0: probe<0><<<256 * waves, 256>>>(out, cyc, iters); break;
          case 1: probe<1><<<256 * waves, 256>>>(out, cyc, iters); break;
          case 2: probe<2><<<256 * waves, 256>>>(out, cyc, iters); break;
          case 3: probe<3><<<256 * waves, 256>>>(out, cyc, iters); break;
          case 4: probe<4><<<256 * waves, 256>>>(out, cyc, iters); break;
          case 5: probe<5><<<256 * waves, 256>>>(out, cyc, iters); break;
          case 6: probe<6><<<256 * waves, 256>>>(out, cyc, iters); break;
          case 7: probe<7><<<256 * waves, 256>>>(out, cyc, iters); break;
        }
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
      }
      std::vector<unsigned long long> h(8);
      hipMemcpy(h.data(), cyc, 64, hipMemcpyDeviceToHost);
      // one workgroup of 4 waves per CU (x waves): each SIMD issues iters * 8 instructions per resident wave
      printf("{\"instruction\": \"%s\", \"waves_per_simd\": %d, \"us\": %.1f, \"ns_per_wave_instruction_and_simd\": %.3f}\n", names[k], waves, ms * 1e3,
             ms * 1e6 / (double)(iters * 8 * waves));
    }
  }
  return 0;
}
