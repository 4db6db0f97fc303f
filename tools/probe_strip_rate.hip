// Cycles per MFMA of the GRU fast path's strip engine (mnexp_amd/csrc/gru4.hip) in its operand modes:
// resident strips, LDS-fed strip, with 1..4 waves of a 256-thread workgroup active.
#include <stdio.h>
#include "../mnexp_amd/csrc/gru4.hip"
using namespace lstur;
template <int M0, int M1>
__global__ __launch_bounds__(256) void rate(const float* img, float* out, unsigned long long* cyc, unsigned mask, int reps,
                                            int wl) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int KQ = 50, QR1 = 40;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 800 + 16 + 51 * wl * 4 + 4 * 10 * 256; i += 256) smem[i] = img[i % 4096];
  float4 u0[KQ], u1[QR1];
#pragma unroll
  for (int q = 0; q < KQ; ++q) u0[q] = ((const float4*)img)[q * 64 + lane];
#pragma unroll
  for (int q = 0; q < QR1; ++q) u1[q] = ((const float4*)img)[(q + 50) * 64 + lane];
  const float* aimg = smem;
  const float* uL = smem + 816 + (size_t)min(wv * 64 + lane, wl - 1) * kqp(KQ) * 4;
  const float* tail = smem + 816 + 51 * wl * 4 + wv * 10 * 256 + lane * 4;
  __syncthreads();
  f32x4 s0 = {0, 0, 0, 0}, s1 = s0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (mask & (1u << wv)) {
    for (int r = 0; r < reps; ++r) {
      f32x4 r0, r1;
      run_strips<KQ, QR1, M0, M1>(u0, u1, M1 == M_LDS ? uL : tail, aimg, lane, r0, r1);
      s0 += r0; s1 += r1;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = s0[0] + s0[1] + s0[2] + s0[3] + s1[0] + s1[1] + s1[2] + s1[3];
  if (lane == 0) cyc[wv] = t1 - t0;
}
template <int M0, int M1>
void run(const char* name, unsigned mask, const float* img, float* out, unsigned long long* cyc, int wl) {
  const int reps = 100;
  const size_t lds = (800 + 16 + 51 * wl * 4 + 4 * 10 * 256) * 4;
  hipFuncSetAttribute((const void*)rate<M0, M1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int it = 0; it < 2; ++it) {
    hipLaunchKernelGGL((rate<M0, M1>), dim3(1), dim3(256), lds, 0, img, out, cyc, mask, reps, wl);
    hipDeviceSynchronize();
  }
  unsigned long long h[4];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const int nm = ((M0 != M_NONE) + (M1 != M_NONE)) * 200;
  printf("%-22s mask=0x%x :", name, mask);
  for (int w = 0; w < 4; ++w)
    if (mask & (1u << w)) printf("  w%d %.2f", w, (double)h[w] / ((double)nm * reps));
  printf("  cyc/mfma\n");
}
int main() {
  float *img, *out; unsigned long long* cyc;
  hipMalloc(&img, 1 << 20); hipMalloc(&out, 1024); hipMalloc(&cyc, 64);
  hipMemset(img, 0, 1 << 20);
  for (unsigned m : {0x1u, 0x7u, 0xfu}) {
    run<M_RES, M_NONE>("slot0 resident", m, img, out, cyc, 136);
    run<M_NONE, M_RES>("slot1 resident+tail", m, img, out, cyc, 136);
    run<M_RES, M_RES>("two resident", m, img, out, cyc, 136);
    run<M_NONE, M_LDS>("LDS-fed wl=136", m, img, out, cyc, 136);
    run<M_NONE, M_LDS>("LDS-fed wl=8", m, img, out, cyc, 8);
  }
  return 0;
}
