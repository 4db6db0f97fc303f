// General fp32 MFMA GEMM for the small dense layers of the LSTUR path (gfx950):
//   C[M,N] = alpha * op(A) . op(B) + bias      (Dense(U) paper.py:159, GRU input projection and the
//   weight/input gradients of both).  64x64 tile, BK = 32, 4 waves x (2x2) v_mfma_f32_16x16x4_f32,
//   register-staged prefetch of the next tile under the MFMAs of the current one.
// Weight-gradient shapes (small M*N, long K) are split along K into partial slabs that a second
// kernel reduces in fixed order (bitwise reproducible, unlike float atomics).
#include "lstur_device.h"
#include "lstur_internal.h"

namespace lstur {

constexpr int GT = 64, GK = 32, GLD = 80;  // tile, k-step, LDS row stride (80 % 32 == 16: conflict-free b32 reads)

// four consecutive elements base[i0 .. i0+3] with element-level bounds (count valid from i0)
__device__ __forceinline__ float4 load4_guard(const float* __restrict__ p, int valid, bool vec) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (vec && valid >= 4) return *(const float4*)p;
  if (valid > 0) v.x = p[0];
  if (valid > 1) v.y = p[1];
  if (valid > 2) v.z = p[2];
  if (valid > 3) v.w = p[3];
  return v;
}

// One 64 x 32 operand tile = 512 float4, two per thread.  KMAJOR: the operand is stored with k contiguous
// (A[m][k] or B[n][k]) and is transposed on its way into the k-major LDS image; otherwise (A[k][m] / B[k][n])
// the image is a straight copy.
template <bool KMAJOR>
__device__ __forceinline__ void tile_load(const float* __restrict__ P, int ld, int x0, int xmax, int k0, int kend,
                                          bool vec, int tid, float4 (&r)[2]) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int idx = tid + q * 256;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (KMAJOR) {
      const int x = idx >> 3, kg = (idx & 7) * 4;
      if (x0 + x < xmax) v = load4_guard(P + (size_t)(x0 + x) * ld + k0 + kg, kend - (k0 + kg), vec);
    } else {
      const int k = idx >> 4, xg = (idx & 15) * 4;
      if (k0 + k < kend) v = load4_guard(P + (size_t)(k0 + k) * ld + x0 + xg, xmax - (x0 + xg), vec);
    }
    r[q] = v;
  }
}
template <bool KMAJOR>
__device__ __forceinline__ void tile_store(float (*S)[GLD], int tid, const float4 (&r)[2]) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int idx = tid + q * 256;
    if (KMAJOR) {
      const int x = idx >> 3, kg = (idx & 7) * 4;
      S[kg + 0][x] = r[q].x;
      S[kg + 1][x] = r[q].y;
      S[kg + 2][x] = r[q].z;
      S[kg + 3][x] = r[q].w;
    } else {
      const int k = idx >> 4, xg = (idx & 15) * 4;
      *(float4*)&S[k][xg] = r[q];
    }
  }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_kernel(const float* __restrict__ A, int lda,
                                                   const float* __restrict__ B, int ldb, float* __restrict__ C,
                                                   int ldc, const float* __restrict__ bias, int M, int N, int K,
                                                   float alpha, int k_per_split, bool vecA, bool vecB,
                                                   bool to_slab) {
  __shared__ __attribute__((aligned(16))) float As[GK][GLD];
  __shared__ __attribute__((aligned(16))) float Bs[GK][GLD];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r16 = lane & 15, kq = lane >> 4;
  const int wr = wv >> 1, wc = wv & 1;
  const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
  const int kbeg = blockIdx.z * k_per_split, kend = min(K, kbeg + k_per_split);

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // register-staged pipeline: the next tile's global loads are in flight while this tile's MFMAs run
  float4 ra[2], rb[2];
  tile_load<!TA>(A, lda, m0, M, kbeg, kend, vecA, tid, ra);
  tile_load<TB>(B, ldb, n0, N, kbeg, kend, vecB, tid, rb);
  for (int k0 = kbeg; k0 < kend; k0 += GK) {
    tile_store<!TA>(As, tid, ra);
    tile_store<TB>(Bs, tid, rb);
    __syncthreads();
    if (k0 + GK < kend) {
      tile_load<!TA>(A, lda, m0, M, k0 + GK, kend, vecA, tid, ra);
      tile_load<TB>(B, ldb, n0, N, k0 + GK, kend, vecB, tid, rb);
    }
#pragma unroll
    for (int s = 0; s < GK / 4; ++s) {
      float a0 = As[s * 4 + kq][wr * 32 + r16];
      float a1 = As[s * 4 + kq][wr * 32 + 16 + r16];
      float b0 = Bs[s * 4 + kq][wc * 32 + r16];
      float b1 = Bs[s * 4 + kq][wc * 32 + 16 + r16];
      acc[0][0] = mfma16(a0, b0, acc[0][0]);
      acc[0][1] = mfma16(a0, b1, acc[0][1]);
      acc[1][0] = mfma16(a1, b0, acc[1][0]);
      acc[1][1] = mfma16(a1, b1, acc[1][1]);
    }
    __syncthreads();
  }

  float* dst = C;
  int ld = ldc;
  if (to_slab) {
    dst = C + (size_t)blockIdx.z * M * N;
    ld = N;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int n = n0 + wc * 32 + j * 16 + r16;
      if (n >= N) continue;
      float bv = (!to_slab && bias) ? bias[n] : 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int m = m0 + wr * 32 + i * 16 + kq * 4 + q;
        if (m < M) dst[(size_t)m * ld + n] = to_slab ? acc[i][j][q] : alpha * acc[i][j][q] + bv;
      }
    }
}

__global__ void gemm_reduce_kernel(const float* __restrict__ slabs, int nsplit, int M, int N, float* __restrict__ C,
                                   int ldc, const float* __restrict__ bias, float alpha) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t count = (size_t)M * N;
  if (i >= count) return;
  float acc = 0.f;
  for (int s = 0; s < nsplit; ++s) acc += slabs[(size_t)s * count + i];
  int m = (int)(i / N), n = (int)(i - (size_t)m * N);
  C[(size_t)m * ldc + n] = alpha * acc + (bias ? bias[n] : 0.f);
}

static int gemm_splits(int M, int N, int K) {
  int tiles = cdiv(M, GT) * cdiv(N, GT);
  if (tiles >= 128 || K < 512) return 1;
  int want = cdiv(256, tiles);
  int maxs = std::max(1, K / 128);
  return std::max(1, std::min(want, maxs));
}

size_t gemm_scratch_bytes(int M, int N, int K) {
  int ns = gemm_splits(M, N, K);
  return ns > 1 ? (size_t)ns * M * N * sizeof(float) : 0;
}

void launch_gemm(const float* A, int lda, int ta, const float* B, int ldb, int tb, float* C, int ldc,
                 const float* bias, int M, int N, int K, float alpha, void* scratch, hipStream_t s, ReduceJob* defer) {
  if (defer) defer->nslices = 0;
  if (M == 0 || N == 0) return;
  int ns = scratch ? gemm_splits(M, N, K) : 1;
  int kps = cdiv(cdiv(K, ns), GK) * GK;
  ns = std::max(1, cdiv(K, kps));
  bool slab = ns > 1;
  bool vecA = (lda % 4 == 0) && (((uintptr_t)A & 15) == 0);
  bool vecB = (ldb % 4 == 0) && (((uintptr_t)B & 15) == 0);
  dim3 grid(cdiv(N, GT), cdiv(M, GT), ns), block(256);
  float* dst = slab ? (float*)scratch : C;
#define LSTUR_GEMM(TA, TB)                                                                                       \
  hipLaunchKernelGGL((gemm_kernel<TA, TB>), grid, block, 0, s, A, lda, B, ldb, dst, ldc, bias, M, N, K, alpha, kps, \
                     vecA, vecB, slab)
  if (!ta && !tb)
    LSTUR_GEMM(false, false);
  else if (!ta && tb)
    LSTUR_GEMM(false, true);
  else if (ta && !tb)
    LSTUR_GEMM(true, false);
  else
    LSTUR_GEMM(true, true);
#undef LSTUR_GEMM
  if (slab && defer && !bias) {
    *defer = ReduceJob{(const float*)scratch, C, M, N, ldc, ns, (size_t)M * N, alpha};
    return;
  }
  if (slab) {
    size_t count = (size_t)M * N;
    hipLaunchKernelGGL(gemm_reduce_kernel, dim3((unsigned)cdivz(count, 256)), dim3(256), 0, s, (const float*)scratch,
                       ns, M, N, C, ldc, bias, alpha);
  }
}

}  // namespace lstur
