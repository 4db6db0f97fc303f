// bf16-operand / fp32-accumulate MFMA GEMMs of the `mhsa` news encoder's projections (gfx950):
//   forward : QKV[N_tok, 3F] (bf16) = E[N_tok, D] . qkv_w[D, 3F] + qkv_b          (launch_proj_fwd_bf16)
//   backward: [dqkv_w ; dqkv_b] [D+1, 3F] (fp32) = [E | 1]^T . dQKV[N_tok, 3F]     (launch_proj_wgrad_bf16)
// v_mfma_f32_32x32x16_bf16; operands are rounded to bf16 (round-to-nearest-even, the oracle's bf16_round) on their way
// into LDS, which keeps the tiles exactly as they lie in HBM:
//   * an operand stored with k contiguous ([m][k], the forward's E) is read back with ds_read_b128 (80-byte rows:
//     conflict-free);
//   * an operand stored with k as its ROW index ([k][n] weights, and BOTH operands of the weight gradient, whose
//     contraction index is the token) is read with ds_read_b64_tr_b16 -- the hardware transposes 4 k-rows x 16 columns
//     per 16-lane group, so no operand is ever re-packed or transposed in HBM.
// Two LDS buffers, register-staged prefetch of the next k-tile under the MFMAs of the current one, one barrier per
// k-tile.  The weight gradient is split along the tokens into slabs reduced in fixed order by the step's single
// multi_reduce launch (bitwise reproducible); its extra row of ones yields the bias gradient in the same pass.
#include "bf16_mfma.h"
#include "lstur_internal.h"

namespace lstur {

// A_KM = false: A is [M][K] (k contiguous), fp32.   A_KM = true: A is [K][M] fp32, and ONES_ROW appends the row
// m == M of ones (bias gradient).  B is [K][N], fp32 (TB = float) or bf16 (TB = unsigned short).
// OUT_BF16: C bf16 [M][ldc] = acc + bias;  else C = fp32 slabs [gridDim.z][Mo][N] (Mo = M + ONES_ROW).
// TA: A's element type in HBM -- float (rounded to bf16 on the way into LDS) or unsigned short (bf16 already: 16-byte copies).
template <int TM, int TN, int WM, int WN, int BK, bool A_KM, bool ONES_ROW, typename TB, bool OUT_BF16, typename TA = float>
__global__ __launch_bounds__(WM* WN * 64) void gemm_bf16_kernel(const TA* __restrict__ A, int lda,
                                                                const TB* __restrict__ B, int ldb, void* __restrict__ C,
                                                                int ldc, const float* __restrict__ bias, int M, int N, int K,
                                                                int k_per_split, int tiles_n, int groups) {
  constexpr int NT = WM * WN * 64, BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int SA = A_KM ? tr_stride(BM) : BK + 8, SB = tr_stride(BN);   // [m][k] rows: 80 / 144 bytes, conflict-free b128
  constexpr int A_ELEMS = A_KM ? BK * SA : BM * SA, B_ELEMS = BK * SB;
  constexpr int A_VEC = sizeof(TA) == 4 ? 4 : 8;                               // elements per 16-byte load of A
  constexpr int KF4 = BK / A_VEC;                                              // 16-byte pieces per k-row of an [m][k] tile
  constexpr int A_F4 = BM * BK / A_VEC, A_PER = (A_F4 + NT - 1) / NT;        // 16-byte pieces per thread for the A tile
  constexpr int B_VEC = sizeof(TB) == 4 ? 4 : 8;                               // elements per 16-byte load of B
  constexpr int B_CH = BK * BN / B_VEC, B_PER = (B_CH + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  unsigned short* As = smem;                  // 2 buffers
  unsigned short* Bs = smem + 2 * A_ELEMS;    // 2 buffers
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv / WN, wn = wv - wm * WN;
  // XCD-aware block order (1-D grid): consecutive workgroup ids go round-robin over the 8 XCDs, each with its own L2,
  // so the `tiles_n` column tiles that re-read the same A rows (forward) / the same token slab of E (weight gradient)
  // are given ids 8 apart: they run on ONE XCD at about the same time and all but the first hit its L2.
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int grp = (seq / tiles_n) * 8 + xcd, bn = seq - (seq / tiles_n) * tiles_n;
  if (grp >= groups) return;
  const int bm = A_KM ? 0 : grp, bz = A_KM ? grp : 0;
  const int m0 = bm * BM, n0 = bn * BN;
  const int kbeg = bz * k_per_split, kend = min(K, kbeg + k_per_split);
  const int Mo = M + (ONES_ROW ? 1 : 0);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Global loads go through buffer descriptors that end at this block's last k-row: reads past it return 0, so the
  // k-tail of a slab needs no branch (a guarded load compiles to a branch + s_waitcnt per load and serialises the
  // whole prefetch).  Rows / columns outside M / N read in-range garbage that only reaches outputs never stored.
  float4 ra[A_PER];
  uint4 rb[B_PER];
  const __amdgpu_buffer_rsrc_t rsA =
      make_rsrc(A, (uint32_t)(A_KM ? (size_t)kend * lda * sizeof(TA) : std::min((size_t)M * lda * sizeof(TA), (size_t)0xFFFFFFFFu)));
  const __amdgpu_buffer_rsrc_t rsB = make_rsrc(B, (uint32_t)((size_t)kend * ldb * sizeof(TB)));
  auto load = [&](int k0) {
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int idx = tid + i * NT;
      uint32_t off;
      if (!A_KM) {
        const int row = idx / KF4, seg = idx - row * KF4;        // k >= K meets zero rows of B
        off = ((uint32_t)(m0 + row) * (uint32_t)lda + (uint32_t)(k0 + seg * A_VEC)) * (uint32_t)sizeof(TA);
      } else {
        const int krow = idx / (BM / A_VEC), seg = idx - krow * (BM / A_VEC);
        off = ((uint32_t)(k0 + krow) * (uint32_t)lda + (uint32_t)(m0 + seg * A_VEC)) * (uint32_t)sizeof(TA);
      }
      ra[i] = buf_load4(rsA, (A_F4 % NT == 0 || idx < A_F4) ? off : kBufOob, 0);
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int idx = tid + i * NT;
      const int krow = idx / (BN / B_VEC), seg = idx - krow * (BN / B_VEC);
      const uint32_t off = ((uint32_t)(k0 + krow) * (uint32_t)ldb + (uint32_t)(n0 + seg * B_VEC)) * (uint32_t)sizeof(TB);
      const float4 v = buf_load4(rsB, (B_CH % NT == 0 || idx < B_CH) ? off : kBufOob, 0);
      rb[i] = *(const uint4*)&v;
    }
  };
  auto store = [&](int buf, int kcur) {
    unsigned short* as = As + buf * A_ELEMS;
    unsigned short* bs = Bs + buf * B_ELEMS;
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int idx = tid + i * NT;
      if (A_F4 % NT == 0 || idx < A_F4) {
        if (!A_KM) {
          const int row = idx / KF4, seg = idx - row * KF4;
          if (sizeof(TA) == 4) *(u16x4*)(as + row * SA + seg * 4) = cvt4(ra[i]);
          else *(float4*)(as + row * SA + seg * 8) = ra[i];
        } else if (sizeof(TA) == 4) {
          const int krow = idx / (BM / 4), seg = idx - krow * (BM / 4);
          u16x4 v = cvt4(ra[i]);
          if (ONES_ROW && m0 + seg * 4 == M) {     // M % 4 == 0: the row of ones (bias gradient) starts a float4
            v[0] = (kcur + krow < kend) ? (unsigned short)0x3F80 : (unsigned short)0;
            v[1] = v[2] = v[3] = 0;
          }
          *(u16x4*)(as + krow * SA + seg * 4) = v;
        } else {
          const int krow = idx / (BM / 8), seg = idx - krow * (BM / 8);
          u16x8 v = __builtin_bit_cast(u16x8, ra[i]);
          const int mrel = M - (m0 + seg * 8);     // M % 4 == 0: the row of ones sits at element 0 or 4 of a piece
          if (ONES_ROW && mrel >= 0 && mrel < 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (e >= mrel) v[e] = (e == mrel && kcur + krow < kend) ? (unsigned short)0x3F80 : (unsigned short)0;
          }
          *(u16x8*)(as + krow * SA + seg * 8) = v;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int idx = tid + i * NT;
      if (B_CH % NT == 0 || idx < B_CH) {
        const int krow = idx / (BN / B_VEC), seg = idx - krow * (BN / B_VEC);
        if (sizeof(TB) == 4) {
          *(u16x4*)(bs + krow * SB + seg * 4) = cvt4(*(const float4*)&rb[i]);
        } else {
          *(uint4*)(bs + krow * SB + seg * 8) = rb[i];
        }
      }
    }
  };

  const int nit = (kend - kbeg + BK - 1) / BK;
  if (nit > 0) load(kbeg);
  for (int it = 0; it < nit; ++it) {
    const int buf = it & 1;
    store(buf, kbeg + it * BK);
    __syncthreads();
    if (it + 1 < nit) load(kbeg + (it + 1) * BK);
    const unsigned short* as = As + buf * A_ELEMS;
    const unsigned short* bs = Bs + buf * B_ELEMS;
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      u16x8 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int mt = (wm * TM + i) * 32;
        if (!A_KM) fa[i] = *(const u16x8*)(as + (mt + (lane & 31)) * SA + 16 * s + 8 * (lane >> 5));
        else fa[i] = tr_frag(as, SA, 16 * s, mt, lane);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = tr_frag(bs, SB, 16 * s, (wn * TN + j) * 32, lane);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = OUT_BF16 ? mfma32b(fb[j], fa[i], acc[i][j]) : mfma32b(fa[i], fb[j], acc[i][j]);
    }
  }

  // Epilogue through buffer descriptors, branch-free (a guarded store costs a branch and a s_waitcnt each: 128
  // serialised round trips per wave): rows past the end fall outside the descriptor, columns past N get the OOB offset.
  const int hh = lane >> 5, c32 = lane & 31;
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  if (OUT_BF16) {
    // operands were swapped in the MFMA: the lane holds output row m, registers 4g..4g+3 four consecutive columns.
    // A 32-row x (TN*32)-column bf16 tile goes through a wave-private LDS strip so that the global stores are whole
    // 16-byte pieces of contiguous row segments (direct stores would be 8 bytes per lane at 64 different rows).
    constexpr int CW = TN * 32, SS = CW + 8, CPR = CW / 8, RPI = 64 / CPR;   // strip: 32 rows of SS elements
    static_assert(2 * A_ELEMS + 2 * B_ELEMS >= WM * WN * 32 * SS, "operand buffers must hold the store strips");
    const __amdgpu_buffer_rsrc_t rsC = make_rsrc(C, (uint32_t)std::min((size_t)M * ldc * 2, (size_t)0xFFFFFFFFu));
    const __amdgpu_buffer_rsrc_t rsBias = make_rsrc(bias, bias ? (uint32_t)N * 4u : 0u);
    __syncthreads();                                   // every wave is done with the operand tiles
    unsigned short* strip = smem + wv * (32 * SS);
    const int srow = lane / CPR, sch = lane - srow * CPR;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nl = j * 32 + 8 * g + 4 * hh;
          const float4 bv = buf_load4(rsBias, (uint32_t)(n0 + wn * CW + nl) * 4u, 0);   // NULL bias / n >= N: zeros
          const f32x4v f = {acc[i][j][4 * g] + bv.x, acc[i][j][4 * g + 1] + bv.y, acc[i][j][4 * g + 2] + bv.z,
                            acc[i][j][4 * g + 3] + bv.w};
          *(u16x4*)(strip + c32 * SS + nl) = __builtin_bit_cast(u16x4, __builtin_convertvector(f, bf16x4));
        }
#pragma unroll
      for (int r0 = 0; r0 < 32; r0 += RPI) {
        const int row = r0 + srow, m = m0 + (wm * TM + i) * 32 + row, n = n0 + wn * CW + sch * 8;
        const uint4 v = *(const uint4*)(strip + row * SS + sch * 8);
        const uint32_t off = n < N ? ((uint32_t)m * (uint32_t)ldc + (uint32_t)n) * 2u : kBufOob;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsC, (int)off, 0, 0);
      }
    }
  } else {
    const __amdgpu_buffer_rsrc_t rsC = make_rsrc((float*)C + (size_t)bz * Mo * N, (uint32_t)((size_t)Mo * N * 4));
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + (wn * TN + j) * 32 + c32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          buf_store1(acc[i][j][r], rsC, col < N ? ((uint32_t)row * (uint32_t)N + (uint32_t)col) * 4u : kBufOob, 0);
        }
      }
  }
}

// ---- forward projection: 256 x 256 workgroup tile, 4 x 2 waves of 64 x 128 ------------------------------------------
bool proj_bf16_supported(int D, int N) { return D % 4 == 0 && N % 8 == 0 && D <= 320; }

void launch_proj_fwd_bf16(const float* E, const float* W, const float* bias, unsigned short* out, int M, int N, int K,
                          hipStream_t s) {
  if (M == 0) return;
  constexpr int TM = 2, TN = 4, WM = 4, WN = 2, BK = 32, BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr size_t lds = (2 * (BM * (BK + 8)) + 2 * (BK * tr_stride(BN))) * sizeof(unsigned short);
  auto kern = gemm_bf16_kernel<TM, TN, WM, WN, BK, false, false, float, true>;
  static bool attr_set = false;
  if (!attr_set) {   // 76 KiB of dynamic LDS
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  const int tiles_n = cdiv(N, BN), groups = cdiv(M, BM);
  hipLaunchKernelGGL(kern, dim3(cdiv(groups, 8) * 8 * tiles_n), dim3(WM * WN * 64), lds, s, E, K, W, N, (void*)out, N, bias, M,
                     N, K, cdiv(K, BK) * BK, tiles_n, groups);
}

// The same with both operands already bf16 in HBM (the step's path: the embedding gather writes bf16 rows for this encoder,
// the weight is rounded once per step): half the operand bytes per k-tile.
void launch_proj_fwd_bf16e(const unsigned short* E16, const unsigned short* W16, const float* bias, unsigned short* out, int M,
                           int N, int K, hipStream_t s) {
  if (M == 0) return;
  constexpr int TM = 2, TN = 4, WM = 4, WN = 2, BK = 32, BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr size_t lds = (2 * (BM * (BK + 8)) + 2 * (BK * tr_stride(BN))) * sizeof(unsigned short);
  auto kern = gemm_bf16_kernel<TM, TN, WM, WN, BK, false, false, unsigned short, true, unsigned short>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  const int tiles_n = cdiv(N, BN), groups = cdiv(M, BM);
  hipLaunchKernelGGL(kern, dim3(cdiv(groups, 8) * 8 * tiles_n), dim3(WM * WN * 64), lds, s, E16, K, W16, N, (void*)out, N, bias,
                     M, N, K, cdiv(K, BK) * BK, tiles_n, groups);
}
__global__ void round_bf16_kernel(const float4* __restrict__ src, u16x4* __restrict__ dst, size_t n4) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n4) dst[i] = cvt4(src[i]);
}
void launch_round_bf16(const float* src, unsigned short* dst, size_t n, hipStream_t s) {   // n % 4 == 0
  if (n) hipLaunchKernelGGL(round_bf16_kernel, dim3((unsigned)cdivz(n / 4, 256)), dim3(256), 0, s, (const float4*)src, (u16x4*)dst, n / 4);
}

// ---- weight (+ bias) gradient: 320 x 128 workgroup tile (all of D), 2 x 4 waves of 160 x 32, token slabs ----------
static int wgrad_splits(int N, int K) {
  const int tiles = cdiv(N, 128);
  // one workgroup per CU (128 KiB of LDS each) and NOT ONE MORE: 43 slabs x 6 column tiles = 258 workgroups ran as two
  // rounds on 256 CUs -- the last two alone in the second -- and the launch took twice as long as 42 x 6 = 252 do
  return std::max(1, std::min(256 / tiles, K / 512));
}
size_t proj_wgrad_scratch_bytes(int D, int N, int K) {
  const int most = std::max(wgrad_splits(N, K), std::max(1, std::min(256 / cdiv(N, 256), K / 512)));   // (either tile width)
  return (size_t)most * (D + 1) * N * sizeof(float);
}

// dW [D+1, N] (row D = bias gradient; the caller's dW and db must be contiguous) through *defer (slabs in `scratch`)
void launch_proj_wgrad_bf16(const float* E, const unsigned short* dqkv, float* dW, void* scratch, int D, int N, int K,
                            hipStream_t s, ReduceJob* defer) {
  constexpr int TM = 5, TN = 1, WM = 2, WN = 4, BK = 64, BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr size_t lds = (2 * (BK * tr_stride(BM)) + 2 * (BK * tr_stride(BN))) * sizeof(unsigned short);
  int ns = wgrad_splits(N, K);
  const int kps = cdiv(cdiv(K, ns), BK) * BK;
  ns = std::max(1, cdiv(K, kps));
  auto kern = gemm_bf16_kernel<TM, TN, WM, WN, BK, true, true, unsigned short, false>;
  static bool attr_set = false;
  if (!attr_set) {   // 128 KiB of dynamic LDS
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  const int tiles_n = cdiv(N, BN);
  hipLaunchKernelGGL(kern, dim3(cdiv(ns, 8) * 8 * tiles_n), dim3(WM * WN * 64), lds, s, E, D, dqkv, N, scratch, N,
                     (const float*)nullptr, D, N, K, kps, tiles_n, ns);
  *defer = ReduceJob{(const float*)scratch, dW, D + 1, N, N, ns, (size_t)(D + 1) * N, 1.0f};
}

template <int TN, int BK>
static void proj_wgrad16_launch(const unsigned short* E16, const unsigned short* dqkv, float* dW, void* scratch, int D, int N,
                                int K, hipStream_t s, ReduceJob* defer) {
  constexpr int TM = 5, WM = 2, WN = 4, BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr size_t lds = (2 * (BK * tr_stride(BM)) + 2 * (BK * tr_stride(BN))) * sizeof(unsigned short);
  const int tiles_n = cdiv(N, BN);
  int ns = std::max(1, std::min(256 / tiles_n, K / 512));   // one workgroup per CU
  const int kps = cdiv(cdiv(K, ns), BK) * BK;
  ns = std::max(1, cdiv(K, kps));
  auto kern = gemm_bf16_kernel<TM, TN, WM, WN, BK, true, true, unsigned short, false, unsigned short>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3(cdiv(ns, 8) * 8 * tiles_n), dim3(WM * WN * 64), lds, s, E16, D, dqkv, N, scratch, N,
                     (const float*)nullptr, D, N, K, kps, tiles_n, ns);
  *defer = ReduceJob{(const float*)scratch, dW, D + 1, N, N, ns, (size_t)(D + 1) * N, 1.0f};
}
void launch_proj_wgrad_bf16e(const unsigned short* E16, const unsigned short* dqkv, float* dW, void* scratch, int D, int N, int K,
                             hipStream_t s, ReduceJob* defer) {
  // 320 x 256 workgroup tile (2 x 4 waves of 160 x 64), k-tiles of 32 tokens: every column tile re-reads the slab's
  // embedding rows, and the launch is bound by what goes through L2 (3 TB/s at this access pattern) -- three column tiles
  // instead of six took it from 177 to 124 us.
  proj_wgrad16_launch<2, 32>(E16, dqkv, dW, scratch, D, N, K, s, defer);
}

}  // namespace lstur
