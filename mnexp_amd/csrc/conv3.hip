// Conv1D('same', relu) x (id != 0) of the cnnatt news encoder (task/paper.py:146-155) as an fp32-GRADE contraction on
// the bf16 matrix cores of gfx950 ("bf16x3"): every fp32 operand is split into three bf16 parts h + m + l (split3,
// bf16_mfma.h: exact to ~2^-25) and the six products hh + hm + mh + hl + lh + mm are accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16.  The dropped terms (ml, lm, ll) are below 2^-24 of |a||b|, i.e. the result carries the
// same error as an fp32 FMA chain in another summation order -- it is NOT a reduced-precision kernel (tests hold it
// to the same tolerances as the exact-fp32 MFMA kernels of conv.hip, and to 2e-6 against them directly).
// The fp32 MFMA (v_mfma_f32_16x16x4_f32) runs at 1/16 of the bf16 rate, so six bf16 products cost 6/16 of the time.
//
// Forward: implicit im2col -- the row of token (n, t) is the contiguous window E_flat[(n*Tp + t)*D .. + k*D) of the
// zero-padded embedding layout -- over the compacted list of non-pad tokens.  One workgroup = 128 token rows x ALL F
// columns (<= 448), 8 waves as 4 (rows) x 2 (7 + up to 7 column tiles of 32); BK = 16: per k-step a wave reads one A
// fragment per split part (ds_read_b128) and the B fragments by transposed LDS reads (ds_read_b64_tr_b16) from the
// row-major [k][n] weight tile, 42 MFMAs per barrier.  The weight's three parts are split once per step
// (conv3_pack_w); the embedding rows are split on their way into LDS.
#include "bf16_mfma.h"
#include "lstur_internal.h"

namespace lstur {

constexpr int C3_BK = 16, C3_SA = 24;   // k-step; A image row stride (48-byte rows: conflict-free ds_read_b128)

// One kernel, two shapes.  <4, 2, 7> (512 threads): 128 token rows x all column tiles (7 + up to 7) -- the bulk.
// <1, 2, 1> (128 threads): 32 rows x 64 columns -- the REMAINDER.  Row tiles of 128 are coarse: 514 of them on 256
// CUs (one resident workgroup per CU, 127 KB of LDS) take three rounds for 2.008 rounds of work.  So the bulk launch
// only takes whole rounds of 256 tiles (plus a last partial round of >= 64 tiles), and the rows behind them go to the
// small shape, whose many light workgroups (7 column groups per 32 rows) spread over all CUs.
__host__ __device__ constexpr int c3_rows(int RW) { return 32 * RW; }
__host__ __device__ constexpr int c3_sb(int CW, int NTW) { return tr_stride(CW * NTW * 32); }
__host__ __device__ constexpr int c3_stage(int RW, int CW, int NTW) {
  return 3 * c3_rows(RW) * C3_SA + 3 * C3_BK * c3_sb(CW, NTW);
}
__host__ __device__ constexpr size_t c3_lds(int RW, int CW, int NTW) {
  return 2 * c3_stage(RW, CW, NTW) * sizeof(unsigned short) + 2 * c3_rows(RW) * sizeof(int32_t);
}
// rows [0, bulk_rows) belong to the bulk shape, [bulk_rows, count) to the remainder shape
__host__ __device__ inline int c3_bulk_rows(int count) {
  const int tiles = (count + 127) / 128, part = tiles % 256;
  const int bulk_tiles = part < 64 ? tiles - part : tiles;
  return min(count, bulk_tiles * 128);
}

bool conv3_supported(int D, int ksize, int F) { return D % 4 == 0 && F >= 1 && F <= 448; }
static int conv3_kp(int K) { return cdiv(K, C3_BK) * C3_BK; }
static int conv3_np(int F) { return cdiv(F, 32) * 32; }
size_t conv3_w3_elems(int D, int ksize, int F) { return (size_t)3 * conv3_kp(ksize * D) * conv3_np(F); }

// W [K, F] fp32 -> three bf16 parts [3][Kp][Np], zero padded
__global__ void conv3_pack_w_kernel(const float* __restrict__ W, int K, int F, int Kp, int Np, unsigned short* __restrict__ W3) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)Kp * Np) return;
  const int n = (int)(i % Np), k = (int)(i / Np);
  const float v = (k < K && n < F) ? W[(size_t)k * F + n] : 0.f;
  u16x4 h, m, l;
  split3(make_float4(v, 0.f, 0.f, 0.f), h, m, l);
  W3[i] = h[0];
  W3[(size_t)Kp * Np + i] = m[0];
  W3[(size_t)2 * Kp * Np + i] = l[0];
}
void launch_conv3_pack_w(const float* W, int K, int F, unsigned short* W3, hipStream_t s) {
  const int Kp = conv3_kp(K), Np = conv3_np(F);
  const size_t total = (size_t)Kp * Np;
  hipLaunchKernelGGL(conv3_pack_w_kernel, dim3((unsigned)cdivz(total, 256)), dim3(256), 0, s, W, K, F, Kp, Np, W3);
}

template <int RW, int CW, int NTW, bool REMAINDER>
__global__ __launch_bounds__(RW* CW * 64) void conv3_fwd_kernel(const float* __restrict__ E, uint32_t e_bytes,
                                                               const unsigned short* __restrict__ W3,
                                                               const float* __restrict__ bias,
                                                               const int32_t* __restrict__ ids, float* __restrict__ C,
                                                               const int32_t* __restrict__ vidx,
                                                               const int32_t* __restrict__ vcount, int M, int T, int Tp, int D,
                                                               int F, int Kp, int Np) {
  constexpr int BM = c3_rows(RW), NTH = RW * CW * 64, SB = c3_sb(CW, NTW), BCOLS = CW * NTW * 32;
  constexpr int A_IMG = BM * C3_SA, B_IMG = C3_BK * SB, STAGE = c3_stage(RW, CW, NTW);
  constexpr int A_PER = (BM * 4 + NTH - 1) / NTH;                    // float4 per thread of the BM x 16 A tile
  constexpr int B_CH = 3 * C3_BK * (BCOLS / 8), B_PER = (B_CH + NTH - 1) / NTH;
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  int32_t* rowm = (int32_t*)(smem + 2 * STAGE);           // output row of each tile row (-1: none)
  float* rowkeep = (float*)(rowm + BM);                   // 1 / 0: id != 0
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int count = vcount ? vcount[0] : M;
  const int bulk = c3_bulk_rows(count);
  const int m0 = REMAINDER ? bulk + blockIdx.x * BM : blockIdx.x * BM;
  const int m_end = REMAINDER ? count : bulk;
  if (m0 >= m_end) return;
  const int NT = Np / 32;                                  // column tiles of 32
  const int col0 = blockIdx.y * BCOLS;                     // first column of this workgroup
  const int wm = wv / CW, wn = wv - wm * CW;
  const int nt_lo = col0 / 32 + wn * NTW, my_nt = max(0, min(NTW, NT - nt_lo));

  // A loader role: piece -> (row, 4 k) of the BM x 16 tile; the row's window offset in E is fixed for the kernel
  uint32_t a_off[A_PER];
  int a_lds[A_PER];
#pragma unroll
  for (int p = 0; p < A_PER; ++p) {
    const int idx = tid + p * NTH, arow = idx >> 2, aseg = idx & 3;
    const bool use = (BM * 4) % NTH == 0 || idx < BM * 4;
    const int i = min(m0 + min(arow, BM - 1), count - 1);
    const int m = vidx ? vidx[i] : i;
    const int title = m / T, t = m - title * T;
    a_off[p] = use ? ((uint32_t)(title * Tp + t) * (uint32_t)D + (uint32_t)aseg * 4u) * 4u : kBufOob;
    a_lds[p] = use ? arow * C3_SA + aseg * 4 : -1;
    if (use && aseg == 0) {
      rowm[arow] = (m0 + arow < m_end) ? m : -1;
      rowkeep[arow] = ids[m] != 0 ? 1.f : 0.f;
    }
  }
  // B loader role: pieces of 16 bytes (8 columns) of the 3 x [16][BCOLS] window of the weight parts
  uint32_t b_goff[B_PER];
  int b_loff[B_PER];
#pragma unroll
  for (int i = 0; i < B_PER; ++i) {
    const int idx = tid + i * NTH;
    constexpr int C8 = BCOLS / 8;
    const int img = idx / (C3_BK * C8), rem = idx - img * (C3_BK * C8), krow = rem / C8, ch = rem - krow * C8;
    const bool use = idx < B_CH && col0 + ch * 8 < Np;
    b_goff[i] = use ? (((uint32_t)img * (uint32_t)Kp + (uint32_t)krow) * (uint32_t)Np + (uint32_t)(col0 + ch * 8)) * 2u : kBufOob;
    b_loff[i] = use ? 3 * A_IMG + img * B_IMG + krow * SB + ch * 8 : -1;
  }
  const __amdgpu_buffer_rsrc_t rsE = make_rsrc(E, e_bytes);
  const __amdgpu_buffer_rsrc_t rsW = make_rsrc(W3, (uint32_t)((size_t)3 * Kp * Np * 2));

  f32x16 acc[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  float4 ra[A_PER];
  float4 rb[B_PER];
  auto load = [&](int k0) {
#pragma unroll
    for (int p = 0; p < A_PER; ++p) ra[p] = buf_load4(rsE, a_off[p], (uint32_t)k0 * 4u);
#pragma unroll
    for (int i = 0; i < B_PER; ++i) rb[i] = buf_load4(rsW, b_goff[i], (uint32_t)k0 * (uint32_t)Np * 2u);
  };
  auto store = [&](int buf) {
    unsigned short* st = smem + buf * STAGE;
#pragma unroll
    for (int p = 0; p < A_PER; ++p) {
      if (a_lds[p] >= 0) {
        u16x4 h, m, l;
        split3(ra[p], h, m, l);
        *(u16x4*)(st + a_lds[p]) = h;
        *(u16x4*)(st + A_IMG + a_lds[p]) = m;
        *(u16x4*)(st + 2 * A_IMG + a_lds[p]) = l;
      }
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i)
      if (b_loff[i] >= 0) *(float4*)(st + b_loff[i]) = rb[i];
  };

  // Software pipeline: global loads two k-steps ahead; the split + LDS store of step it + 1 is issued in step it before
  // that step's MFMAs with no barrier in between (its VALU work overlaps the matrix pipe); one barrier per step.
  const int nit = Kp / C3_BK;
  load(0);
  store(0);
  if (nit > 1) load(C3_BK);
  __syncthreads();
  // The two waves that share a SIMD (w and w + NW/2) run the two halves of a step in opposite order (see the kernel
  // gradient below): one stores the next tile while the other issues MFMAs.
  auto prefetch = [&](int it) {
    if (it + 1 < nit) {
      store((it & 1) ^ 1);
      if (it + 2 < nit) load((it + 2) * C3_BK);
    }
  };
  auto contract = [&](int it) {
    const unsigned short* st = smem + (it & 1) * STAGE;
    const unsigned short* ap = st + (32 * wm + (lane & 31)) * C3_SA + 8 * (lane >> 5);
    const u16x8 ah = *(const u16x8*)ap, am = *(const u16x8*)(ap + A_IMG), al = *(const u16x8*)(ap + 2 * A_IMG);
    const unsigned short* bp = st + 3 * A_IMG;
    // column tiles two at a time, their six-product chains interleaved (consecutive MFMAs never wait on each other's
    // accumulator); the `j < my_nt` guards are wave-uniform, so EXEC stays full for the transposed reads
    auto frags = [&](int j, u16x8 (&b)[3]) {
      const int x0 = (wn * NTW + j) * 32;                // column inside this workgroup's window
      b[0] = tr_frag(bp, SB, 0, x0, lane);
      b[1] = tr_frag(bp + B_IMG, SB, 0, x0, lane);
      b[2] = tr_frag(bp + 2 * B_IMG, SB, 0, x0, lane);
    };
#pragma unroll
    for (int j = 0; j < NTW; j += 2) {
      if (j + 1 < NTW && j + 1 < my_nt) {
        u16x8 b0[3], b1[3];
        frags(j, b0);
        frags(j + 1, b1);
        f32x16 c0 = acc[j], c1 = acc[j + 1];
        c0 = mfma32b(al, b0[0], c0);  c1 = mfma32b(al, b1[0], c1);     // small terms first
        c0 = mfma32b(ah, b0[2], c0);  c1 = mfma32b(ah, b1[2], c1);
        c0 = mfma32b(am, b0[1], c0);  c1 = mfma32b(am, b1[1], c1);
        c0 = mfma32b(am, b0[0], c0);  c1 = mfma32b(am, b1[0], c1);
        c0 = mfma32b(ah, b0[1], c0);  c1 = mfma32b(ah, b1[1], c1);
        c0 = mfma32b(ah, b0[0], c0);  c1 = mfma32b(ah, b1[0], c1);
        acc[j] = c0;
        acc[j + 1] = c1;
      } else if (j < my_nt) {
        u16x8 b0[3];
        frags(j, b0);
        f32x16 c = acc[j];
        c = mfma32b(al, b0[0], c);
        c = mfma32b(ah, b0[2], c);
        c = mfma32b(am, b0[1], c);
        c = mfma32b(am, b0[0], c);
        c = mfma32b(ah, b0[1], c);
        c = mfma32b(ah, b0[0], c);
        acc[j] = c;
      }
    }
  };
  const bool store_first = true;   // (opposite orders on the two waves of a SIMD pay off in the kernel gradient, whose
                                   // store phase is 4x heavier; here they measured the same)
  for (int it = 0; it < nit; ++it) {
    if (store_first) {
      prefetch(it);
      contract(it);
    } else {
      contract(it);
      prefetch(it);
    }
    __syncthreads();
  }

  // epilogue: + bias, ReLU, x (id != 0); rows scattered to the token's place, branch-free through a buffer descriptor
  const int hh = lane >> 5, c32 = lane & 31;
  const __amdgpu_buffer_rsrc_t rsC = make_rsrc(C, (uint32_t)std::min((size_t)M * F * 4, (size_t)0xFFFFFFFFu));
  const __amdgpu_buffer_rsrc_t rsBias = make_rsrc(bias, (uint32_t)F * 4u);
  int orow[16];
  float okeep[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int lr = 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * hh;
    orow[r] = rowm[lr];
    okeep[r] = rowkeep[lr];
  }
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int col = (nt_lo + j) * 32 + c32;
    const bool cok = j < my_nt && col < F;
    const float bv = buf_load1(rsBias, cok ? (uint32_t)col * 4u : kBufOob, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r)
      buf_store1(fmaxf(acc[j][r] + bv, 0.f) * okeep[r], rsC,
                 (cok && orow[r] >= 0) ? ((uint32_t)orow[r] * (uint32_t)F + (uint32_t)col) * 4u : kBufOob, 0);
  }
}

void launch_conv3_fwd(const float* emb, const unsigned short* W3, const float* bias, const int32_t* ids, float* out,
                      int n_titles, int T, int D, int ksize, int F, const int32_t* vidx, const int32_t* vcount,
                      hipStream_t s) {
  const int M = n_titles * T;
  if (M == 0) return;
  const int Tp = T + ksize - 1, K = ksize * D, Np = conv3_np(F);
  auto bulk = conv3_fwd_kernel<4, 2, 7, false>;
  auto rest = conv3_fwd_kernel<1, 2, 1, true>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)bulk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c3_lds(4, 2, 7));
    attr_set = true;
  }
  // the padded embedding layout + its zero slack (kEmbSlack floats): the k-tail of the last window reads into it
  const uint32_t e_bytes = (uint32_t)std::min(((size_t)n_titles * Tp * D + kEmbSlack) * 4, (size_t)0xFFFFFFFFu);
  hipLaunchKernelGGL(bulk, dim3(cdiv(M, 128)), dim3(512), c3_lds(4, 2, 7), s, emb, e_bytes, W3, bias, ids, out, vidx, vcount,
                     M, T, Tp, D, F, conv3_kp(K), Np);
  // remainder: fewer than 64 tiles of 128 rows (the device decides from the token count), 64 columns per workgroup
  const int rest_rows = std::min(M, 64 * 128);
  hipLaunchKernelGGL(rest, dim3(cdiv(rest_rows, 32), cdiv(Np, 64)), dim3(128), c3_lds(1, 2, 1), s, emb, e_bytes, W3, bias, ids,
                     out, vidx, vcount, M, T, Tp, D, F, conv3_kp(K), Np);
}


// ---------------------------------------------------------------------------------------------
// Kernel gradient dW[k*D, F] = im2col(E)^T . dC over the (compacted) tokens, same bf16x3 scheme.
// The contraction index is the token, and BOTH operands lie token-major in HBM (the window of a token is 900
// contiguous floats of the padded layout, its conv-output gradient a row of 400): both LDS tiles are row-major
// [token][x] images read with transposed LDS reads.  Workgroup = 8 waves, tile 256 (k*D) x 224 (F): wave w owns
// row tile w and 7 column tiles (112 accumulator registers); 16 tokens per step, 42 MFMAs per wave and barrier.
// Tokens are cut into slabs (one per workgroup row); slabs are reduced in fixed order by the step's multi_reduce.
// ---------------------------------------------------------------------------------------------
constexpr int W3_TM = 8, W3_TN = 7, W3_BK = 16, W3_THREADS = W3_TM * 64;
constexpr int W3_BMR = W3_TM * 32, W3_BNC = W3_TN * 32;                       // 256 x 224
constexpr int W3_SA = tr_stride(W3_BMR), W3_SB = tr_stride(W3_BNC);          // 288 / 224 elements
constexpr int W3_A_IMG = W3_BK * W3_SA, W3_B_IMG = W3_BK * W3_SB;
constexpr int W3_STAGE = 3 * W3_A_IMG + 3 * W3_B_IMG;
constexpr size_t W3_LDS = 2 * W3_STAGE * sizeof(unsigned short);
constexpr int W3_APC = W3_BK * (W3_BMR / 4) / W3_THREADS;                     // 2 float4 per thread and step
constexpr int W3_BPC = (W3_BK * (W3_BNC / 4) + W3_THREADS - 1) / W3_THREADS;  // 3
static_assert(W3_BK * (W3_BMR / 4) % W3_THREADS == 0, "A pieces divide evenly");

static int conv3_wgrad_mblocks(int K) { return cdiv(cdiv(K, 32), W3_TM); }
static int conv3_wgrad_nblocks(int F) { return cdiv(cdiv(F, 32), W3_TN); }
// slabs: about one workgroup per CU (or whole multiples), at least 512 tokens each
static int conv3_wgrad_slabs(int M, int K, int F) {
  const int per = conv3_wgrad_mblocks(K) * conv3_wgrad_nblocks(F);
  const int rounds = std::max(1, cdiv(M, 120000));                 // ~105 600 tokens at B = 64 -> one round
  // whole multiples of 8 slabs: each XCD (32 CUs) then gets the same number of workgroups (<= 32 per round)
  const int want = rounds * 256 / per / 8 * 8;
  return std::max(1, std::min(std::max(8, want), std::max(1, M / 512)));
}
size_t conv3_wgrad_scratch_bytes(int n_titles, int T, int D, int ksize, int F) {
  return (size_t)conv3_wgrad_slabs(n_titles * T, ksize * D, F) * ksize * D * F * sizeof(float);
}

template <bool HAS_OFF>
__global__ __launch_bounds__(W3_THREADS) void conv3_wgrad_kernel(const float* __restrict__ E, uint32_t e_bytes,
                                                                 const float* __restrict__ G, uint32_t g_bytes,
                                                                 float* __restrict__ slabs,
                                                                 const int32_t* __restrict__ vidx,
                                                                 const int32_t* __restrict__ vcount, int M, int T, int Tp,
                                                                 int D, int F, int K, int mblocks, int nblocks, int nslabs,
                                                                 const int32_t* __restrict__ offA,
                                                                 const int32_t* __restrict__ offB) {
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int count = vcount ? vcount[0] : M;
  // block -> (slab, m-block, n-block); the blocks of one slab are adjacent (they re-read the same tokens from L2)
  // XCD-aware order: workgroup ids go round-robin over the 8 XCDs (each with its own L2), so the blocks of one slab get
  // ids 8 apart: they run on ONE XCD at about the same time and share the slab's tokens through its L2
  const int per = mblocks * nblocks;
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int slab = (seq / per) * 8 + xcd, rem = seq - (seq / per) * per, mb = rem / nblocks, nb = rem - mb * nblocks;
  if (slab >= nslabs) return;
  const int per_slab = (count + nslabs - 1) / nslabs, tpad = (per_slab + W3_BK - 1) / W3_BK * W3_BK;
  const int t_begin = slab * tpad, t_end = min(count, t_begin + tpad);
  const int kd0 = mb * W3_BMR, f0 = nb * W3_BNC;
  const __amdgpu_buffer_rsrc_t rsE = make_rsrc(E, e_bytes), rsG = make_rsrc(G, g_bytes);

  f32x16 acc[W3_TN];
#pragma unroll
  for (int j = 0; j < W3_TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // loader roles: A piece p -> (token row, 4 kd); B piece p -> (token row, 4 f)
  int a_kr[W3_APC], a_sg[W3_APC], b_kr[W3_BPC], b_sg[W3_BPC];
#pragma unroll
  for (int p = 0; p < W3_APC; ++p) {
    const int idx = tid + p * W3_THREADS;
    a_kr[p] = idx / (W3_BMR / 4);
    a_sg[p] = idx - a_kr[p] * (W3_BMR / 4);
  }
#pragma unroll
  for (int p = 0; p < W3_BPC; ++p) {
    const int idx = tid + p * W3_THREADS;
    b_kr[p] = idx / (W3_BNC / 4);
    b_sg[p] = idx - b_kr[p] * (W3_BNC / 4);      // b_kr >= 16: not a piece of the tile
  }
  // Row offsets of a step's pieces come in two halves so that no wait sits between the prefetch loads: `fetch_rows`
  // only issues the index loads (clamped, branch-free), `make_offsets` turns them into byte offsets one step later.
  // A token past the slab's end contributes 0 through the out-of-range offset of G.
  uint32_t a_off[W3_APC], b_off[W3_BPC];
  int a_raw[W3_APC], b_raw[W3_BPC];
  auto fetch_rows = [&](int t0) {
#pragma unroll
    for (int p = 0; p < W3_APC; ++p) {
      const int tok = min(t0 + a_kr[p], count - 1);
      a_raw[p] = HAS_OFF ? offA[tok] : (vidx ? vidx[tok] : tok);
    }
#pragma unroll
    for (int p = 0; p < W3_BPC; ++p) {
      const int tok = min(t0 + b_kr[p], count - 1);
      b_raw[p] = HAS_OFF ? offB[tok] : (vidx ? vidx[tok] : tok);
    }
  };
  auto make_offsets = [&](int t0) {
#pragma unroll
    for (int p = 0; p < W3_APC; ++p) {
      uint32_t row = (uint32_t)a_raw[p];              // HAS_OFF: float offset of the token's window (misc.hip)
      if (!HAS_OFF) {
        const int title = a_raw[p] / T, t = a_raw[p] - title * T;
        row = (uint32_t)(title * Tp + t) * (uint32_t)D;
      }
      a_off[p] = (row + (uint32_t)(kd0 + a_sg[p] * 4)) * 4u;
    }
#pragma unroll
    for (int p = 0; p < W3_BPC; ++p) {
      const bool ok = b_kr[p] < W3_BK && t0 + b_kr[p] < t_end;
      const uint32_t row = HAS_OFF ? (uint32_t)b_raw[p] : (uint32_t)b_raw[p] * (uint32_t)F;
      b_off[p] = ok ? (row + (uint32_t)(f0 + b_sg[p] * 4)) * 4u : kBufOob;
    }
  };
  float4 ra[W3_APC], rb[W3_BPC];
  auto load = [&]() {
#pragma unroll
    for (int p = 0; p < W3_APC; ++p) ra[p] = buf_load4(rsE, a_off[p], 0);
#pragma unroll
    for (int p = 0; p < W3_BPC; ++p) rb[p] = buf_load4(rsG, b_off[p], 0);
  };
  auto store = [&](int buf) {
    unsigned short* st = smem + buf * W3_STAGE;
#pragma unroll
    for (int p = 0; p < W3_APC; ++p) {
      u16x4 h, m, l;
      split3(ra[p], h, m, l);
      unsigned short* d = st + a_kr[p] * W3_SA + a_sg[p] * 4;
      *(u16x4*)d = h;
      *(u16x4*)(d + W3_A_IMG) = m;
      *(u16x4*)(d + 2 * W3_A_IMG) = l;
    }
#pragma unroll
    for (int p = 0; p < W3_BPC; ++p) {
      if (b_kr[p] < W3_BK) {
        u16x4 h, m, l;
        split3(rb[p], h, m, l);
        unsigned short* d = st + 3 * W3_A_IMG + b_kr[p] * W3_SB + b_sg[p] * 4;
        *(u16x4*)d = h;
        *(u16x4*)(d + W3_B_IMG) = m;
        *(u16x4*)(d + 2 * W3_B_IMG) = l;
      }
    }
  };

  // Software pipeline: global loads run two steps ahead (their row offsets three), the split + LDS store of step
  // it + 1 is issued in step it BEFORE that step's MFMAs and with no barrier between them, so its VALU work overlaps the
  // matrix pipe; one barrier per step.
  const int nit = t_end > t_begin ? (t_end - t_begin + W3_BK - 1) / W3_BK : 0;
  if (nit > 0) {
    fetch_rows(t_begin);
    make_offsets(t_begin);
    load();                                          // step 0
    fetch_rows(t_begin + W3_BK);
    store(0);
    make_offsets(t_begin + W3_BK);
    load();                                          // step 1
    fetch_rows(t_begin + 2 * W3_BK);
    __syncthreads();
  }
  // The two waves that share a SIMD (w and w + 4) run the two halves of a step in opposite order: one splits + stores
  // the next tile while the other issues its MFMAs, then they swap -- both orders are legal inside a step (one barrier
  // at its end), and the matrix pipe no longer idles while both waves of a SIMD convert and store.
  auto prefetch = [&](int it) {
    if (it + 1 < nit) {
      store((it & 1) ^ 1);                           // step it + 1 (its loads were issued a step ago)
      make_offsets(t_begin + (it + 2) * W3_BK);      // from the rows fetched a step ago
      load();                                        // step it + 2
      fetch_rows(t_begin + (it + 3) * W3_BK);        // consumed in the next step
    }
  };
  auto contract = [&](int it) {
    const unsigned short* st = smem + (it & 1) * W3_STAGE;
    const unsigned short* bp = st + 3 * W3_A_IMG;
    const u16x8 ah = tr_frag(st, W3_SA, 0, wv * 32, lane);
    const u16x8 am = tr_frag(st + W3_A_IMG, W3_SA, 0, wv * 32, lane);
    const u16x8 al = tr_frag(st + 2 * W3_A_IMG, W3_SA, 0, wv * 32, lane);
    // two column tiles at a time, their six-product chains interleaved: consecutive MFMAs never wait on each other's
    // accumulator
    auto six = [&](f32x16& c0, f32x16& c1, const u16x8 (&b0)[3], const u16x8 (&b1)[3]) {
      c0 = mfma32b(al, b0[0], c0);  c1 = mfma32b(al, b1[0], c1);
      c0 = mfma32b(ah, b0[2], c0);  c1 = mfma32b(ah, b1[2], c1);
      c0 = mfma32b(am, b0[1], c0);  c1 = mfma32b(am, b1[1], c1);
      c0 = mfma32b(am, b0[0], c0);  c1 = mfma32b(am, b1[0], c1);
      c0 = mfma32b(ah, b0[1], c0);  c1 = mfma32b(ah, b1[1], c1);
      c0 = mfma32b(ah, b0[0], c0);  c1 = mfma32b(ah, b1[0], c1);
    };
    auto frags = [&](int j, u16x8 (&b)[3]) {
      b[0] = tr_frag(bp, W3_SB, 0, j * 32, lane);
      b[1] = tr_frag(bp + W3_B_IMG, W3_SB, 0, j * 32, lane);
      b[2] = tr_frag(bp + 2 * W3_B_IMG, W3_SB, 0, j * 32, lane);
    };
#pragma unroll
    for (int j = 0; j + 1 < W3_TN; j += 2) {
      u16x8 b0[3], b1[3];
      frags(j, b0);
      frags(j + 1, b1);
      six(acc[j], acc[j + 1], b0, b1);
    }
    if (W3_TN & 1) {
      u16x8 b0[3];
      frags(W3_TN - 1, b0);
      f32x16 c = acc[W3_TN - 1];
      c = mfma32b(al, b0[0], c);
      c = mfma32b(ah, b0[2], c);
      c = mfma32b(am, b0[1], c);
      c = mfma32b(am, b0[0], c);
      c = mfma32b(ah, b0[1], c);
      c = mfma32b(ah, b0[0], c);
      acc[W3_TN - 1] = c;
    }
  };
  const bool store_first = wv < W3_TM / 2;           // wave-uniform
  for (int it = 0; it < nit; ++it) {
    if (store_first) {
      prefetch(it);
      contract(it);
    } else {
      contract(it);
      prefetch(it);
    }
    __syncthreads();
  }

  // slab [K, F] of this token range (zeros when the range is empty: the reduction reads every slab)
  const int hh = lane >> 5, c32 = lane & 31;
  const __amdgpu_buffer_rsrc_t rsC = make_rsrc(slabs + (size_t)slab * K * F, (uint32_t)((size_t)K * F * 4));
#pragma unroll
  for (int j = 0; j < W3_TN; ++j) {
    const int f = f0 + j * 32 + c32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kd = kd0 + wv * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
      buf_store1(acc[j][r], rsC, (f < F && kd < K) ? ((uint32_t)kd * (uint32_t)F + (uint32_t)f) * 4u : kBufOob, 0);
    }
  }
}

// dw [k*D, F] through *defer (slabs in `scratch`, conv3_wgrad_scratch_bytes); dout rows of pad tokens are not read
void launch_conv3_wgrad(const float* emb, const float* dout, float* dw, void* scratch, int n_titles, int T, int D,
                        int ksize, int F, const int32_t* vidx, const int32_t* vcount, hipStream_t s, ReduceJob* defer,
                        const int32_t* offA, const int32_t* offB) {
  const int M = n_titles * T, Tp = T + ksize - 1, K = ksize * D;
  if (M == 0) {
    defer->nslices = 0;
    return;
  }
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv3_wgrad_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)W3_LDS);
    (void)hipFuncSetAttribute((const void*)conv3_wgrad_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)W3_LDS);
    attr_set = true;
  }
  const int mblocks = conv3_wgrad_mblocks(K), nblocks = conv3_wgrad_nblocks(F), ns = conv3_wgrad_slabs(M, K, F);
  const uint32_t e_bytes = (uint32_t)std::min(((size_t)n_titles * Tp * D + kEmbSlack) * 4, (size_t)0xFFFFFFFFu);
  const uint32_t g_bytes = (uint32_t)std::min((size_t)M * F * 4, (size_t)0xFFFFFFFFu);
  const dim3 grid(cdiv(ns, 8) * 8 * mblocks * nblocks);
  if (offA && offB)
    hipLaunchKernelGGL(conv3_wgrad_kernel<true>, grid, dim3(W3_THREADS), W3_LDS, s, emb, e_bytes, dout, g_bytes,
                       (float*)scratch, vidx, vcount, M, T, Tp, D, F, K, mblocks, nblocks, ns, offA, offB);
  else
    hipLaunchKernelGGL(conv3_wgrad_kernel<false>, grid, dim3(W3_THREADS), W3_LDS, s, emb, e_bytes, dout, g_bytes,
                       (float*)scratch, vidx, vcount, M, T, Tp, D, F, K, mblocks, nblocks, ns, offA, offB);
  *defer = ReduceJob{(const float*)scratch, dw, K, F, F, ns, (size_t)K * F, 1.0f};
}

}  // namespace lstur
