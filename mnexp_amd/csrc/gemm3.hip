// fp32-grade GEMM on the bf16 matrix cores for the small dense layers ON the step's critical path (gfx950):
//   C[M,N] = A[M,K] . op(B) + bias      A row-major fp32; B fp32, [K][N] (B_NK = false) or [N][K] (B_NK = true: C = A . B^T)
// i.e. Dense(U) forward, the GRU input projection, and the input gradients of both (reference task/paper.py:159, 592-613):
// 0.6 - 0.8 GFLOP each, four of them per step between the big kernels.  Both operands are split into three bf16 parts
// on their way into LDS (bf16_mfma.h: split3) and six products per k-step reproduce the fp32 product to ~2^-23 -- the
// scheme of the conv kernels (conv3f.hip); the exact-fp32 MFMA kernel of gemm.hip stays for everything else.
//
// Why a second kernel: at these sizes the fp32 MFMA kernel is a sum of small latencies -- 32 v_mfma_f32_16x16x4_f32 per
// wave and k-step behind scalar LDS reads, 64-byte store segments -- and took 16 - 30 us per GEMM.  Here a k-step of a
// 64 x 64 tile is 12 v_mfma_f32_32x32x16_bf16 per wave behind 16-byte fragment reads, the operand loads of EIGHT k-steps
// are in flight from the start (a register ring with compile-time slots; K = 200 is seven steps: the whole operand panel
// is requested before the first MFMA), LDS is double-buffered with one barrier per step, and an output row segment is 128
// contiguous bytes per store.
#include "bf16_mfma.h"
#include "lstur_internal.h"

namespace lstur {

constexpr int G3_T = 64, G3_K = 32;                    // tile, k-step
#ifndef G3_PIPE
#define G3_PIPE 1
#endif
constexpr int G3_SA = G3_K + 8;                        // [x][k] image row: 80 bytes (conflict-free 16-byte fragment reads)
constexpr int G3_SB = tr_stride(G3_T);                 // [k][n] image row (transposed fragment reads)
constexpr int G3_AIMG = G3_T * G3_SA;                  // elements of one part of an [x][k] tile
constexpr int G3_BIMG = G3_K * G3_SB > G3_AIMG ? G3_K * G3_SB : G3_AIMG;
constexpr int G3_BUF = 3 * G3_AIMG + 3 * G3_BIMG;     // elements of one LDS buffer

// one operand tile = 2 float4 per thread.  XK: the operand is stored [x][k] (k contiguous); else [k][x].
template <bool XK>
__device__ __forceinline__ void g3_load(__amdgpu_buffer_rsrc_t rs, int ld, int x0, int k0, int tid, float4 (&r)[2]) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int idx = tid + q * 256;
    uint32_t off;
    if (XK) off = ((uint32_t)(x0 + (idx >> 3)) * (uint32_t)ld + (uint32_t)(k0 + (idx & 7) * 4)) * 4u;
    else off = ((uint32_t)(k0 + (idx >> 4)) * (uint32_t)ld + (uint32_t)(x0 + (idx & 15) * 4)) * 4u;
    r[q] = buf_load4(rs, off, 0);
  }
}
// split into the three bf16 images of the tile; k >= K is stored as zeros (an [x][k] row runs on into the next row there;
// past the end of a [k][x] operand the descriptor already returned zeros)
template <bool XK>
__device__ __forceinline__ void g3_store(unsigned short* img, int img_elems, int tid, const float4 (&r)[2], int k0, int K) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int idx = tid + q * 256;
    float4 v = r[q];
    int pos;
    if (XK) {
      if (k0 + (idx & 7) * 4 >= K) v = make_float4(0.f, 0.f, 0.f, 0.f);
      pos = (idx >> 3) * G3_SA + (idx & 7) * 4;
    } else {
      if (k0 + (idx >> 4) >= K) v = make_float4(0.f, 0.f, 0.f, 0.f);   // (a slab's tail: the next slab's rows)
      pos = (idx >> 4) * G3_SB + (idx & 15) * 4;
    }
    u16x4 h, m, l;
    split3(v, h, m, l);
    *(u16x4*)(img + pos) = h;
    *(u16x4*)(img + img_elems + pos) = m;
    *(u16x4*)(img + 2 * img_elems + pos) = l;
  }
}

// A_KM: A is stored [K][M] (C = A^T . B: the parameter gradients, contraction over the rows of both operands); such
// launches are split along K into gridDim.z slabs of k_per_split rows, written as C + z * M * N (ld N, no bias) and summed
// in fixed order by the caller's reduction (bitwise reproducible).
// NS: k-steps of operand loads in flight (even).  8 for the critical-path products (alone on the chip between the big
// kernels: latency is all that matters); 2 for the parameter-gradient form, which has to squeeze in BESIDE the attention
// backward on the side stream -- 96 registers instead of 188.
template <bool A_KM, bool B_NK, int NS>
__global__ __launch_bounds__(256) void gemm3_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                    float* __restrict__ C, int ldc, const float* __restrict__ bias, int M,
                                                    int N, int Kfull, uint32_t bytesA, uint32_t bytesB, int k_per_split) {
  extern __shared__ __attribute__((aligned(16))) unsigned short sm3[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1;
  // C = A . op(B) launches come as a 1-D grid in XCD-aware order: consecutive workgroup ids go round-robin over the 8 XCDs
  // (each with its own L2), so the column tiles (workgroups) that read the SAME rows of A get ids 8 apart --
  // they run on one XCD at about the same time and all but the first find the rows in its L2
  int by = blockIdx.y, bx = blockIdx.x;
  if (!A_KM) {
    const int tn = k_per_split, seq = blockIdx.x >> 3;   // (k_per_split carries the column-tile count of these launches)
    by = (seq / tn) * 8 + (blockIdx.x & 7);
    bx = seq - (seq / tn) * tn;
    if (by * G3_T >= M) return;
  }
  const int m0 = by * G3_T, n0 = bx * G3_T;
  const int kbeg = A_KM ? (int)blockIdx.z * k_per_split : 0, K = A_KM ? min(Kfull, kbeg + k_per_split) : Kfull;
  constexpr int AIMG = A_KM ? G3_BIMG : G3_AIMG;      // elements of one part of the A tile image
  constexpr int BUF = 3 * AIMG + 3 * G3_BIMG;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(A, bytesA), rsB = make_rsrc(B, bytesB);
  float4 ra[NS][2], rb[NS][2];
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    g3_load<!A_KM>(rsA, lda, m0, kbeg + u * G3_K, tid, ra[u]);
    g3_load<B_NK>(rsB, ldb, n0, kbeg + u * G3_K, tid, rb[u]);
  }
  g3_store<!A_KM>(sm3, AIMG, tid, ra[0], kbeg, K);
  g3_store<B_NK>(sm3 + 3 * AIMG, G3_BIMG, tid, rb[0], kbeg, K);
  g3_load<!A_KM>(rsA, lda, m0, kbeg + NS * G3_K, tid, ra[0]);
  g3_load<B_NK>(rsB, ldb, n0, kbeg + NS * G3_K, tid, rb[0]);
  __syncthreads();
  for (int k0 = kbeg; k0 < K;) {
#pragma unroll
    for (int u = 0; u < NS; ++u) {      // step k0 / G3_K == u (mod NS); its tile is in LDS buffer u & 1
      const int kn = k0 + G3_K;
      // (1) ALL fragments of this step out of LDS first: the stores of the next step's tile below go to the other buffer, but
      // the compiler orders every LDS read behind every earlier LDS write -- reads first leaves it free to mix what follows
      const unsigned short* as = sm3 + (u & 1) * BUF;
      const unsigned short* bs = as + 3 * AIMG;
      u16x8 fa[G3_K / 16][3], fb[G3_K / 16][3];
#pragma unroll
      for (int s = 0; s < G3_K / 16; ++s)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          if (A_KM) fa[s][p] = tr_frag(as + p * AIMG, G3_SB, 16 * s, wm * 32, lane);
          else fa[s][p] = *(const u16x8*)(as + p * AIMG + (wm * 32 + (lane & 31)) * G3_SA + 16 * s + 8 * (lane >> 5));
          if (B_NK) fb[s][p] = *(const u16x8*)(bs + p * G3_BIMG + (wn * 32 + (lane & 31)) * G3_SA + 16 * s + 8 * (lane >> 5));
          else fb[s][p] = tr_frag(bs + p * G3_BIMG, G3_SB, 16 * s, wn * 32, lane);
        }
      {   // (2) stage the next step (loaded NS - 1 steps ago), unconditionally: past K it stores zeros nobody reads -- a
          // branch here makes the compiler wait for ALL outstanding loads
        const int nx = (u + 1) % NS;
        unsigned short* nb = sm3 + ((u + 1) & 1) * BUF;
        g3_store<!A_KM>(nb, AIMG, tid, ra[nx], kn, K);
        g3_store<B_NK>(nb + 3 * AIMG, G3_BIMG, tid, rb[nx], kn, K);
        g3_load<!A_KM>(rsA, lda, m0, kn + NS * G3_K, tid, ra[nx]);
        g3_load<B_NK>(rsB, ldb, n0, kn + NS * G3_K, tid, rb[nx]);
      }
      // (3) the step's 12 MFMAs are ONE dependent chain (a single 32 x 32 accumulator) -- a wave issues in order, and left
      // alone the compiler puts the ~140 VALU instructions of (2) in front of the chain, which then runs with nothing beside
      // it.  The scheduling directives below interleave them: one MFMA, a dozen VALU, one LDS store.
#pragma unroll
      for (int s = 0; s < G3_K / 16; ++s) {
        acc = mfma32b(fa[s][2], fb[s][0], acc);   // l h   (small terms first)
        acc = mfma32b(fa[s][0], fb[s][2], acc);   // h l
        acc = mfma32b(fa[s][1], fb[s][1], acc);   // m m
        acc = mfma32b(fa[s][1], fb[s][0], acc);   // m h
        acc = mfma32b(fa[s][0], fb[s][1], acc);   // h m
        acc = mfma32b(fa[s][0], fb[s][0], acc);   // h h
      }
      if (G3_PIPE) {
#pragma unroll
        for (int i = 0; i < 6 * (G3_K / 16); ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);   // VALU
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);    // DS write
        }
      }
      __syncthreads();
      k0 = kn;
      if (k0 >= K) break;
    }
  }
  // rows >= M fall outside the descriptor, columns >= N get the out-of-range offset: no branches
  if (A_KM) {
    C += (size_t)blockIdx.z * M * N;
    ldc = N;
  }
  const __amdgpu_buffer_rsrc_t rsC = make_rsrc(C, (uint32_t)(((size_t)(M - 1) * ldc + N) * 4));
  const __amdgpu_buffer_rsrc_t rsBias = make_rsrc(bias, (bias && !A_KM) ? (uint32_t)N * 4u : 0u);
  const int col = n0 + wn * 32 + (lane & 31);
  const float bv = buf_load1(rsBias, (uint32_t)col * 4u, 0);     // no bias / col >= N: 0
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    buf_store1(acc[r] + bv, rsC, col < N ? ((uint32_t)row * (uint32_t)ldc + (uint32_t)col) * 4u : kBufOob, 0);
  }
}

bool gemm3_supported(int lda, int ldb, int M, int N, int K) {
  static const bool off = getenv("LSTUR_GEMM_F32") != nullptr;
  return !off && M > 0 && N > 0 && K > 0 && K % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 &&
         (size_t)M * lda * 4 < ((size_t)1 << 32) && (size_t)std::max(N, K) * ldb * 4 < ((size_t)1 << 32);
}

static void g3_attrs() {
  static bool attr_set = false;
  if (attr_set) return;
  constexpr int lds = 2 * (6 * G3_BIMG) * (int)sizeof(unsigned short);   // the largest variant
  (void)hipFuncSetAttribute((const void*)gemm3_kernel<false, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  (void)hipFuncSetAttribute((const void*)gemm3_kernel<false, false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  (void)hipFuncSetAttribute((const void*)gemm3_kernel<true, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  attr_set = true;
}
// b_nk: B is [N][K] (C = A . B^T); else [K][N].  The caller checks gemm3_supported.
void launch_gemm3(const float* A, int lda, const float* B, int ldb, bool b_nk, float* C, int ldc, const float* bias, int M,
                  int N, int K, hipStream_t s) {
  constexpr size_t lds = 2 * G3_BUF * sizeof(unsigned short);
  g3_attrs();
  const uint32_t bytesA = (uint32_t)(((size_t)(M - 1) * lda + K) * 4);
  const uint32_t bytesB = (uint32_t)((b_nk ? (size_t)(N - 1) * ldb + K : (size_t)(K - 1) * ldb + N) * 4);
  const int tn = cdiv(N, G3_T), tm = cdiv(M, G3_T);
  const dim3 grid(cdiv(tm, 8) * 8 * tn);   // XCD-aware linear id (see the kernel)
  if (b_nk)
    launch_k(gemm3_kernel<false, true, 8>, grid, dim3(256), lds, s, A, lda, B, ldb, C, ldc, bias, M, N, K, bytesA, bytesB, tn);
  else
    launch_k(gemm3_kernel<false, false, 8>, grid, dim3(256), lds, s, A, lda, B, ldb, C, ldc, bias, M, N, K, bytesA, bytesB, tn);
}

// C[M,N] = A^T . B with A [K][M], B [K][N] (parameter gradients: long K, small M x N), split along K into slabs in
// `scratch` (gemm3_tn_scratch_bytes) that *job sums in slab order.
static int g3_tn_splits(int M, int N, int K) {
  const int tiles = cdiv(M, G3_T) * cdiv(N, G3_T);
  return std::max(1, std::min(cdiv(512, tiles), K / (4 * G3_K)));
}
bool gemm3_tn_supported(int lda, int ldb, int M, int N, int K) {
  static const bool off = getenv("LSTUR_GEMM_F32") != nullptr;
  return !off && M > 0 && N > 0 && K > 0 && lda % 4 == 0 && ldb % 4 == 0 && (size_t)K * std::max(lda, ldb) * 4 < ((size_t)1 << 32);
}
size_t gemm3_tn_scratch_bytes(int M, int N, int K) { return (size_t)g3_tn_splits(M, N, K) * M * N * sizeof(float); }
void launch_gemm3_tn(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K, float alpha,
                     void* scratch, hipStream_t s, ReduceJob* job) {
  constexpr size_t lds = 2 * (6 * G3_BIMG) * sizeof(unsigned short);
  g3_attrs();
  int ns = g3_tn_splits(M, N, K);
  const int kps = cdiv(cdiv(K, ns), G3_K) * G3_K;
  ns = std::max(1, cdiv(K, kps));
  const uint32_t bytesA = (uint32_t)(((size_t)(K - 1) * lda + M) * 4), bytesB = (uint32_t)(((size_t)(K - 1) * ldb + N) * 4);
  hipLaunchKernelGGL((gemm3_kernel<true, false, 2>), dim3(cdiv(N, G3_T), cdiv(M, G3_T), ns), dim3(256), lds, s, A, lda, B, ldb,
                     (float*)scratch, N, (const float*)nullptr, M, N, K, bytesA, bytesB, kps);
  *job = ReduceJob{(const float*)scratch, C, M, N, ldc, ns, (size_t)M * N, alpha};
}

}  // namespace lstur
