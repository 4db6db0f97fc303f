// Probe of v_mfma_f32_4x4x1_16b_f32 with A-block broadcast (cbsz = 4): prints which (row, col) each
// output register/lane holds when lanes 0..3 carry A rows and lane l carries B column l.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out) {
  int l = threadIdx.x;
  // A: lane l<4 holds row value 10^l ... use distinct primes; other lanes hold garbage 1000
  float a = l < 4 ? (float)(l + 1) : 1000.f;       // row i -> i+1
  float b = (float)(100 * (l + 1));                 // col l -> 100(l+1)
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 0, 0);
  for (int j = 0; j < 4; ++j) out[j * 64 + l] = c[j];
}
int main() {
  float* d; hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int j = 0; j < 4; ++j) for (int l = 0; l < 64; ++l) {
    float want = (float)(j + 1) * 100.f * (l + 1);
    if (h[j * 64 + l] != want) { if (bad < 8) printf("vgpr %d lane %d: got %g want %g\n", j, l, h[j*64+l], want); ++bad; }
  }
  printf("probe_mfma4: %s (%d mismatches)\n", bad ? "LAYOUT DIFFERS" : "OK: D[vgpr j][lane l] = A_row_j * B_col_l", bad);
  return 0;
}
