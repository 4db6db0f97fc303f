// Issue-rate probe for v_mfma_f32_4x4x1_16b_f32 (cbsz=4) streams as the GRU fast path uses them:
// which SIMD each wave of a 512-thread workgroup lands on, and cycles per MFMA for 1/2/4 accumulator chains
// with different sets of active waves.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(512) void rate(float* out, unsigned long long* cyc, unsigned* hwid, unsigned mask, int reps) {
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float b[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) b[i] = (float)(lane + i) * 1e-3f;
  float a = (float)lane * 1e-2f;
  f32x4 c[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) c[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (lane == 0) hwid[wv] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (mask & (1u << wv)) {
    for (int r = 0; r < reps; ++r) {
#pragma unroll
      for (int i = 0; i < 64; ++i) c[i % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b[i], c[i % NACC], 4, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  f32x4 s = c[0];
#pragma unroll
  for (int i = 1; i < NACC; ++i) s += c[i];
  out[threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (lane == 0) cyc[wv] = t1 - t0;
}
template <int NACC>
void run(unsigned mask, float* out, unsigned long long* cyc, unsigned* hw) {
  const int reps = 200;
  for (int it = 0; it < 2; ++it) {
    hipLaunchKernelGGL(rate<NACC>, dim3(1), dim3(512), 0, 0, out, cyc, hw, mask, reps);
    hipDeviceSynchronize();
  }
  unsigned long long h[8]; unsigned id[8];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  hipMemcpy(id, hw, sizeof(id), hipMemcpyDeviceToHost);
  printf("acc=%d mask=0x%02x :", NACC, mask);
  for (int w = 0; w < 8; ++w)
    if (mask & (1u << w)) printf("  w%d(simd%u) %.2f", w, (id[w] >> 4) & 3, (double)h[w] / (64.0 * reps));
  printf("  cyc/mfma\n");
}
int main() {
  float* out; unsigned long long* cyc; unsigned* hw;
  hipMalloc(&out, 512 * 4); hipMalloc(&cyc, 64); hipMalloc(&hw, 32);
  unsigned masks[] = {0x01, 0x11, 0x03, 0x0f, 0xff};
  for (unsigned m : masks) { run<1>(m, out, cyc, hw); run<2>(m, out, cyc, hw); run<4>(m, out, cyc, hw); }
  return 0;
}
