// Issue-rate probe for the GRU recurrence's candidate instruction: v_mfma_f32_4x4x4_16b_f16 in A-broadcast mode (cbsz = 4),
// against the v_mfma_f32_4x4x1_16b_f32 the fast path runs today (tools/probe_mfma4_rate.hip).  One wave per SIMD, 1 / 2 / 4
// accumulator chains.  A step of the recurrence is h [4 x K] . U [K x N]: K MFMAs per 64 columns with 4x4x1 f32 (one k each),
// 3 K / 4 with 4x4x4 f16 on two-way fp16 splits of both operands (hh + hl + lh, four k each) -- it pays iff the f16
// instruction costs about what the f32 one does.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mfma4h_rate.hip -o tools/probe_mfma4h_rate && tools/probe_mfma4h_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool F16>
__global__ __launch_bounds__(256) void rate(float* out, unsigned long long* cyc, int reps) {
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x4 c[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) c[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bf[64];
  f16x4 bh[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    bf[i] = (float)(lane + i) * 1e-3f;
    bh[i] = (f16x4){(_Float16)(lane * 0.01f + i), (_Float16)(i * 0.5f), (_Float16)(lane * 0.1f), (_Float16)1.0f};
  }
  const float af = (float)lane * 1e-2f;
  const f16x4 ah = (f16x4){(_Float16)(lane * 0.02f), (_Float16)0.5f, (_Float16)2.0f, (_Float16)(lane * 0.001f)};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int i = 0; i < 64; i += 4) {   // (abid must be a literal: the four k-quads of one operand register, as in the GRU)
      if (F16) {
        c[i % NACC] = __builtin_amdgcn_mfma_f32_4x4x4f16(ah, bh[i], c[i % NACC], 4, 0, 0);
        c[(i + 1) % NACC] = __builtin_amdgcn_mfma_f32_4x4x4f16(ah, bh[i + 1], c[(i + 1) % NACC], 4, 1, 0);
        c[(i + 2) % NACC] = __builtin_amdgcn_mfma_f32_4x4x4f16(ah, bh[i + 2], c[(i + 2) % NACC], 4, 2, 0);
        c[(i + 3) % NACC] = __builtin_amdgcn_mfma_f32_4x4x4f16(ah, bh[i + 3], c[(i + 3) % NACC], 4, 3, 0);
      } else {
        c[i % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(af, bf[i], c[i % NACC], 4, 0, 0);
        c[(i + 1) % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(af, bf[i + 1], c[(i + 1) % NACC], 4, 1, 0);
        c[(i + 2) % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(af, bf[i + 2], c[(i + 2) % NACC], 4, 2, 0);
        c[(i + 3) % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(af, bf[i + 3], c[(i + 3) % NACC], 4, 3, 0);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  f32x4 s = c[0];
#pragma unroll
  for (int i = 1; i < NACC; ++i) s += c[i];
  out[threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (lane == 0) cyc[wv] = t1 - t0;
}
template <int NACC, bool F16>
void run(float* out, unsigned long long* cyc) {
  const int reps = 200;
  for (int it = 0; it < 2; ++it) {
    hipLaunchKernelGGL((rate<NACC, F16>), dim3(1), dim3(256), 0, 0, out, cyc, reps);
    hipDeviceSynchronize();
  }
  unsigned long long h[4];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-28s chains=%d :", F16 ? "v_mfma_f32_4x4x4_16b_f16" : "v_mfma_f32_4x4x1_16b_f32", NACC);
  for (int w = 0; w < 4; ++w) printf("  w%d %.2f", w, (double)h[w] / (64.0 * reps));
  printf("  cycles / MFMA (s_memtime ticks)\n");
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 4); hipMalloc(&cyc, 64);
  run<1, false>(out, cyc); run<2, false>(out, cyc); run<4, false>(out, cyc);
  run<1, true>(out, cyc); run<2, true>(out, cyc); run<4, true>(out, cyc);
  // fp16 subnormal operands: kept or flushed?  2^-20 * 2^-4 = 2^-24 must come out of the product
  return 0;
}
