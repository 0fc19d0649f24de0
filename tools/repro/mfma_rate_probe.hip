// Issue rate of v_mfma_f32_16x16x32_f16 in the operand patterns of the fragment-stream kernels (gfx950), one wave per SIMD: cycles per MFMA from s_memtime around
// a loop of 96 MFMAs (one k_mlp3 double phase), on random data.  Round 6: the stamped k_mlp3 build with neither LDS reads nor refills took ~23 cycles per MFMA where
// MI355X_MICROARCH.md gives 16 -- which operand pattern costs the difference?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate_probe tools/repro/mfma_rate_probe.hip && /tmp/mfma_rate_probe
//   KIND 0: 48 independent accumulators x 2 sweeps, A rotating over 8 register quads, B two fixed quads (k_mlp3's second half: y += W2 relu(h))
//   KIND 1: 4 accumulation chains in rotation, A rotating over 8 quads, B rotating over 24 quads (k_mlp3's first half: h = W1 x)
//   KIND 2: KIND 1 then KIND 0 with the fp32 -> fp16 repack of the 4 chain results in between (the whole double phase, registers only)
//   KIND 3: 96 MFMAs on ONE accumulator (a single dependent chain)
//   KIND 4: KIND 0 with A and B the same registers in every MFMA (no operand rotation)
//   KIND 5: 16 independent accumulators x 6 sweeps            KIND 6: 8 x 12 sweeps          KIND 7: 4 x 24 sweeps (no A rotation inside a sweep: = KIND 1's chains without its B rotation)
//   KIND 8: 48 accumulators, each hit TWICE in a row (pairs)   KIND 9: 24 accumulators, each hit 4 times in a row
//   KIND 10: 48 independent accumulators, two MFMAs on other accumulators apart but walking the tiles with stride 2 (register distance)
// build a second binary with -DACC_IN_VGPR to see the same loops with the accumulators in VGPRs (they fit: 48 x 4 + 8 x 4 + 24 x 4 = 320 > 256 only for KIND 1's 24 B quads, which then
// share registers with the accumulators it does not use)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// the accumulate-in-place form the kernels' loops compile to (vDst = srcC), with the accumulator's register file chosen by the constraint: hipcc's own allocation of this
// probe's builtin MFMAs shuffled the AGPR accumulators through v_accvgpr_mov / read / write (25.5 cycles per "MFMA" were those copies)
#ifdef ACC_IN_VGPR
__device__ inline floatx4 mma(half8 a, half8 b, floatx4 c) {
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  return c;
}
#else
__device__ inline floatx4 mma(half8 a, half8 b, floatx4 c) {
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  return c;
}
#endif

template <int KIND>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void probe(const half8* in, float* out, unsigned long long* cyc, int iters) {
  const int lane = threadIdx.x;
  half8 A[8], B[24];
  for (int i = 0; i < 8; ++i) A[i] = in[lane + i * 256];
  for (int i = 0; i < 24; ++i) B[i] = in[lane + (8 + i) * 256];
  floatx4 acc[48], h[4];
  for (int i = 0; i < 48; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 4; ++i) h[i] = floatx4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0 || KIND == 4) {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int f = 0; f < 24; ++f) {
          acc[f * 2] = mma(A[KIND == 4 ? 0 : f % 8], B[0], acc[f * 2]);
          acc[f * 2 + 1] = mma(A[KIND == 4 ? 0 : f % 8], B[KIND == 4 ? 0 : 1], acc[f * 2 + 1]);
        }
    }
    if (KIND == 1) {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int f = 0; f < 24; ++f) {
          const int u = f % 2, c = f / 2;
          h[u * 2] = mma(A[f % 8], B[c * 2], h[u * 2]);
          h[u * 2 + 1] = mma(A[f % 8], B[c * 2 + 1], h[u * 2 + 1]);
        }
    }
    if (KIND == 2) {
#pragma unroll
      for (int f = 0; f < 24; ++f) {
        const int u = f % 2, c = f / 2;
        h[u * 2] = mma(A[f % 8], B[c * 2], h[u * 2]);
        h[u * 2 + 1] = mma(A[f % 8], B[c * 2 + 1], h[u * 2 + 1]);
      }
      half8 hb[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        half8 v = {(_Float16)h[j][0], (_Float16)h[j][1], (_Float16)h[j][2], (_Float16)h[j][3], (_Float16)h[2 + j][0], (_Float16)h[2 + j][1], (_Float16)h[2 + j][2], (_Float16)h[2 + j][3]};
        half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        hb[j] = __builtin_elementwise_max(v, z);
        h[j] = floatx4{0.1f, 0.f, 0.f, 0.f};
        h[2 + j] = floatx4{0.f, 0.1f, 0.f, 0.f};
      }
#pragma unroll
      for (int f = 0; f < 24; ++f) {
        acc[f * 2] = mma(A[f % 8], hb[0], acc[f * 2]);
        acc[f * 2 + 1] = mma(A[f % 8], hb[1], acc[f * 2 + 1]);
      }
    }
    if (KIND == 5 || KIND == 6 || KIND == 7) {
      constexpr int NA = KIND == 5 ? 16 : KIND == 6 ? 8 : 4;
#pragma unroll
      for (int s = 0; s < 96 / NA; ++s)
#pragma unroll
        for (int f = 0; f < NA; ++f) acc[f] = mma(A[(s + f) % 8], B[f % 2], acc[f]);
    }
    if (KIND == 8) {
#pragma unroll
      for (int f = 0; f < 48; ++f) {
        acc[f] = mma(A[f % 8], B[0], acc[f]);
        acc[f] = mma(A[(f + 1) % 8], B[1], acc[f]);
      }
    }
    if (KIND == 9) {
#pragma unroll
      for (int f = 0; f < 24; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[f] = mma(A[(f + r) % 8], B[r % 2], acc[f]);
    }
    if (KIND == 10) {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int f = 0; f < 48; ++f) acc[(f * 2) % 48 + (f * 2) / 48] = mma(A[f % 8], B[f % 2], acc[(f * 2) % 48 + (f * 2) / 48]);
    }
    if (KIND == 3) {
#pragma unroll
      for (int f = 0; f < 96; ++f) h[0] = mma(A[f % 8], B[f % 24], h[0]);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  floatx4 s = h[0] + h[1] + h[2] + h[3];
  for (int i = 0; i < 48; ++i) s += acc[i];
  out[threadIdx.x + blockIdx.x * 256] = s[0] + s[1] + s[2] + s[3];
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    cyc[KIND * 2] = t1 - t0;
    cyc[KIND * 2 + 1] = r1 - r0;
  }
}

int main(int argc, char** argv) {
  const int grid_arg = argc > 1 ? atoi(argv[1]) : 1;      // workgroups (one per CU): 1 = the issue-rate measurement; 256 = every CU busy (PMC calibration on a full chip)
  half8* in;
  float* out;
  unsigned long long* cyc;
  const int n = 32 * 256;
  std::vector<_Float16> hin((size_t)n * 8);
  srand(1);
  for (auto& v : hin) v = (_Float16)((rand() % 2001 - 1000) * 1e-3f);
  (void)hipMalloc(&in, n * sizeof(half8));
  (void)hipMalloc(&out, 1 << 22);
  (void)hipMalloc(&cyc, 256);
  (void)hipMemcpy(in, hin.data(), n * sizeof(half8), hipMemcpyHostToDevice);
  (void)hipMemset(cyc, 0, 256);
  const int iters = 2000;
  const char* names[11] = {"48 independent accumulators (second half)", "4 chains in rotation (first half)", "whole double phase, registers only", "one dependent chain", "independent, fixed A / B registers",
                           "16 independent accumulators", "8 independent accumulators", "4 independent accumulators", "48 accumulators, 2 MFMAs in a row each", "24 accumulators, 4 in a row each",
                           "48 independent accumulators, stride 2"};
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  for (int grid : {grid_arg}) {
    for (int k = 0; k < 11; ++k) {
      float ms = 0;
      for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(a);
        switch (k) {
          case 0: probe<0><<<grid, 256>>>(in, out, cyc, iters); break;
          case 1: probe<1><<<grid, 256>>>(in, out, cyc, iters); break;
          case 2: probe<2><<<grid, 256>>>(in, out, cyc, iters); break;
          case 3: probe<3><<<grid, 256>>>(in, out, cyc, iters); break;
          case 4: probe<4><<<grid, 256>>>(in, out, cyc, iters); break;
          case 5: probe<5><<<grid, 256>>>(in, out, cyc, iters); break;
          case 6: probe<6><<<grid, 256>>>(in, out, cyc, iters); break;
          case 7: probe<7><<<grid, 256>>>(in, out, cyc, iters); break;
          case 8: probe<8><<<grid, 256>>>(in, out, cyc, iters); break;
          case 9: probe<9><<<grid, 256>>>(in, out, cyc, iters); break;
          case 10: probe<10><<<grid, 256>>>(in, out, cyc, iters); break;
        }
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        (void)hipEventElapsedTime(&ms, a, b);
      }
      unsigned long long h[32];
      (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      const double per = (double)h[k * 2] / (96.0 * iters), ghz = (double)h[k * 2] / ((double)h[k * 2 + 1] * 10.0);
      printf("{\"workgroups\": %d, \"pattern\": \"%s\", \"cycles_per_mfma\": %.2f, \"clock_ghz\": %.2f, \"launch_us\": %.1f, \"tflops_chip_equiv\": %.0f}\n", grid, names[k], per, ghz, ms * 1e3,
             grid * 4.0 * 96.0 * iters * 16384.0 / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
