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
constexpr int GQ = GT * GK / 4 / 256;       // float4 per thread and operand tile
constexpr int GKF = GK / 4;                 // float4 per k-row of a k-contiguous operand

// One 64 x GK operand tile, GQ float4 per thread (GK = 64 was tried: the transposing LDS stores of the k-contiguous
// operand double per step and the forward shapes lose 40 %).  KMAJOR: the operand is stored with k contiguous
// (A[m][k] or B[n][k]) and is transposed on its way into the k-major LDS image; otherwise (A[k][m] / B[k][n])
// the image is a straight copy.
// Loads go through a buffer descriptor covering exactly the operand (reads past its end return 0) and carry NO
// guards: a guarded load compiles to a branch plus a s_waitcnt per load, which turned the prefetch of a tile into four
// serialised round trips.  The k-tail (k >= kend: the next slab's rows, or the next row's head) is zeroed by selects
// when the tile is written to LDS; rows / columns past M / N hold in-range garbage that only reaches outputs which
// are never stored.
template <bool KMAJOR>
__device__ __forceinline__ void tile_load(__amdgpu_buffer_rsrc_t rs, int ld, int x0, int k0, int tid, float4 (&r)[GQ]) {
#pragma unroll
  for (int q = 0; q < GQ; ++q) {
    const int idx = tid + q * 256;
    uint32_t off;
    if (KMAJOR) {
      const int x = idx / GKF, kg = (idx % GKF) * 4;
      off = ((uint32_t)(x0 + x) * (uint32_t)ld + (uint32_t)(k0 + kg)) * 4u;
    } else {
      const int k = idx >> 4, xg = (idx & 15) * 4;
      off = ((uint32_t)(k0 + k) * (uint32_t)ld + (uint32_t)(x0 + xg)) * 4u;
    }
    r[q] = buf_load4(rs, off, 0);
  }
}
template <bool KMAJOR>
__device__ __forceinline__ void tile_store(float (*S)[GLD], int tid, const float4 (&r)[GQ], int k0, int kend) {
#pragma unroll
  for (int q = 0; q < GQ; ++q) {
    const int idx = tid + q * 256;
    if (KMAJOR) {
      const int x = idx / GKF, kg = (idx % GKF) * 4, kv = kend - (k0 + kg);   // valid k from kg on
      S[kg + 0][x] = kv > 0 ? r[q].x : 0.f;
      S[kg + 1][x] = kv > 1 ? r[q].y : 0.f;
      S[kg + 2][x] = kv > 2 ? r[q].z : 0.f;
      S[kg + 3][x] = kv > 3 ? r[q].w : 0.f;
    } else {
      const int k = idx >> 4, xg = (idx & 15) * 4;
      const bool ok = k0 + k < kend;
      *(float4*)&S[k][xg] = ok ? r[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_kernel(const float* __restrict__ A, int lda,
                                                   const float* __restrict__ B, int ldb, float* __restrict__ C,
                                                   int ldc, const float* __restrict__ bias, int M, int N, int K,
                                                   float alpha, int k_per_split, uint32_t bytesA, uint32_t bytesB,
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
  float4 ra[GQ], rb[GQ];
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(A, bytesA), rsB = make_rsrc(B, bytesB);
  tile_load<!TA>(rsA, lda, m0, kbeg, tid, ra);
  tile_load<TB>(rsB, ldb, n0, kbeg, tid, rb);
  for (int k0 = kbeg; k0 < kend; k0 += GK) {
    tile_store<!TA>(As, tid, ra, k0, kend);
    tile_store<TB>(Bs, tid, rb, k0, kend);
    __syncthreads();
    if (k0 + GK < kend) {
      tile_load<!TA>(rsA, lda, m0, k0 + GK, tid, ra);
      tile_load<TB>(rsB, ldb, n0, k0 + GK, tid, rb);
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

  // branch-free epilogue: rows >= M fall outside the descriptor (it ends behind row M - 1), columns >= N get the
  // out-of-range offset
  float* dst = C;
  int ld = ldc;
  if (to_slab) {
    dst = C + (size_t)blockIdx.z * M * N;
    ld = N;
  }
  const __amdgpu_buffer_rsrc_t rsC = make_rsrc(dst, (uint32_t)(((size_t)(M - 1) * ld + N) * 4));
  const __amdgpu_buffer_rsrc_t rsBias = make_rsrc(bias, (!to_slab && bias) ? (uint32_t)N * 4u : 0u);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wc * 32 + j * 16 + r16;
      const float bv = buf_load1(rsBias, (uint32_t)n * 4u, 0);     // no bias / n >= N: 0
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = m0 + wr * 32 + i * 16 + kq * 4 + q;
        const float v = to_slab ? acc[i][j][q] : alpha * acc[i][j][q] + bv;
        buf_store1(v, rsC, n < N ? ((uint32_t)m * (uint32_t)ld + (uint32_t)n) * 4u : kBufOob, 0);
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
  // operand extents in bytes (descriptor ranges): op(A) is M x K, op(B) is K x N
  const uint32_t bytesA = (uint32_t)((ta ? (size_t)(K - 1) * lda + M : (size_t)(M - 1) * lda + K) * 4);
  const uint32_t bytesB = (uint32_t)((tb ? (size_t)(N - 1) * ldb + K : (size_t)(K - 1) * ldb + N) * 4);
  dim3 grid(cdiv(N, GT), cdiv(M, GT), ns), block(256);
  float* dst = slab ? (float*)scratch : C;
#define LSTUR_GEMM(TA, TB)                                                                                       \
  hipLaunchKernelGGL((gemm_kernel<TA, TB>), grid, block, 0, s, A, lda, B, ldb, dst, ldc, bias, M, N, K, alpha, kps, \
                     bytesA, bytesB, slab)
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
