// Does a chain of v_mfma_f32_32x32x16_f16 into ONE accumulator issue back to back?  Patterns per wave, 1 or 2 waves per SIMD:
//   0: every MFMA into the same accumulator;  1: three in a row per accumulator, 7 accumulators round robin (the conv kernels'
//   tile_products order);  2: the three products of TWO tiles interleaved (distance 2);  3: 7 accumulators, one MFMA each, round robin
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mfma32_dep.hip -o tools/probe_mfma32_dep && tools/probe_mfma32_dep
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
template <int PAT>
__global__ __launch_bounds__(512) void k(float* out, int iters, unsigned long long* cyc) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.01f); }
  f32x16 acc[7];
  for (int j = 0; j < 7; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {   // 42 MFMAs per iteration in every pattern
    if (PAT == 0) {
#pragma unroll
      for (int i = 0; i < 42; ++i) acc[0] = MF(a, b, acc[0]);
    } else if (PAT == 1) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 7; ++j) { acc[j] = MF(a, b, acc[j]); acc[j] = MF(b, a, acc[j]); acc[j] = MF(a, a, acc[j]); }
    } else if (PAT == 2) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int j = 0; j < 6; j += 2) {
          acc[j] = MF(a, b, acc[j]); acc[j + 1] = MF(a, b, acc[j + 1]);
          acc[j] = MF(b, a, acc[j]); acc[j + 1] = MF(b, a, acc[j + 1]);
          acc[j] = MF(a, a, acc[j]); acc[j + 1] = MF(a, a, acc[j + 1]);
        }
        acc[6] = MF(a, b, acc[6]); acc[6] = MF(b, a, acc[6]); acc[6] = MF(a, a, acc[6]);
      }
    } else if (PAT == 4) {   // three tiles interleaved (distance 3): 2 groups of 3 + 1 single per round
#pragma unroll
      for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int j = 0; j < 6; j += 3) {
          acc[j] = MF(a, b, acc[j]); acc[j + 1] = MF(a, b, acc[j + 1]); acc[j + 2] = MF(a, b, acc[j + 2]);
          acc[j] = MF(b, a, acc[j]); acc[j + 1] = MF(b, a, acc[j + 1]); acc[j + 2] = MF(b, a, acc[j + 2]);
          acc[j] = MF(a, a, acc[j]); acc[j + 1] = MF(a, a, acc[j + 1]); acc[j + 2] = MF(a, a, acc[j + 2]);
        }
        acc[6] = MF(a, b, acc[6]); acc[6] = MF(b, a, acc[6]); acc[6] = MF(a, a, acc[6]);
      }
    } else if (PAT == 5) {   // four / three tiles interleaved (distance 4 / 3)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        acc[0] = MF(a, b, acc[0]); acc[1] = MF(a, b, acc[1]); acc[2] = MF(a, b, acc[2]); acc[3] = MF(a, b, acc[3]);
        acc[0] = MF(b, a, acc[0]); acc[1] = MF(b, a, acc[1]); acc[2] = MF(b, a, acc[2]); acc[3] = MF(b, a, acc[3]);
        acc[0] = MF(a, a, acc[0]); acc[1] = MF(a, a, acc[1]); acc[2] = MF(a, a, acc[2]); acc[3] = MF(a, a, acc[3]);
        acc[4] = MF(a, b, acc[4]); acc[5] = MF(a, b, acc[5]); acc[6] = MF(a, b, acc[6]);
        acc[4] = MF(b, a, acc[4]); acc[5] = MF(b, a, acc[5]); acc[6] = MF(b, a, acc[6]);
        acc[4] = MF(a, a, acc[4]); acc[5] = MF(a, a, acc[5]); acc[6] = MF(a, a, acc[6]);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int j = 0; j < 7; ++j) acc[j] = MF(a, b, acc[j]);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int j = 0; j < 7; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  auto run = [&](const char* name, auto kern, int threads) {
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
      hipExtLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, e0, e1, 0, out, iters, cyc);
      hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double per_simd = (double)iters * 42 * (threads / 256);   // MFMAs per SIMD
    printf("%-52s %d waves/SIMD  %8.1f us  %.1f ns per MFMA per SIMD  (s_memtime: %.1f ticks per MFMA of the wave)\n", name, threads / 256,
           best * 1e3, best * 1e6 / per_simd, (double)c / (iters * 42.0));
  };
  run("0: one accumulator", k<0>, 256); run("0: one accumulator", k<0>, 512);
  run("1: 3 in a row, 7 accumulators (conv kernels)", k<1>, 256); run("1: 3 in a row, 7 accumulators (conv kernels)", k<1>, 512);
  run("2: products of two tiles interleaved", k<2>, 256); run("2: products of two tiles interleaved", k<2>, 512);
  run("4: three tiles interleaved", k<4>, 256); run("4: three tiles interleaved", k<4>, 512);
  run("5: four + three tiles interleaved", k<5>, 256); run("5: four + three tiles interleaved", k<5>, 512);
  run("3: 7 accumulators round robin", k<3>, 256); run("3: 7 accumulators round robin", k<3>, 512);
  return 0;
}
