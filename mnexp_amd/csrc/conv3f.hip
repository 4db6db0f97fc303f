// Conv1D('same', relu) x (id != 0) of the cnnatt news encoder (task/paper.py:146-155), forward, second generation of the
// fp32-grade "bf16x3" contraction of conv3.hip: BOTH operands arrive PRE-SPLIT.
//
//   * the word-embedding gather (misc.hip: gather3_kernel) writes the three bf16 parts h | m | l of every embedding
//     element straight into the padded per-title layout "E3": [title][T + k - 1][Dp / 8][part][8] bf16 (Dp = D rounded up to
//     8; the pad elements d >= D are zeros), i.e. the three 16-byte pieces of an 8-element k-chunk lie side by side, the
//     window of token (n, t) is the contiguous run of k * Dp / 8 chunks that starts at padded row n * Tp + t, and the
//     contraction index is k' = tap * Dp + d;
//   * the kernel weight is split once per step into W3 [part][Kp][Np] (conv3_pack_w2: rows in k' order, zero rows at d >= D).
//   So the k-loop carries no conversion arithmetic at all: 16-byte copies HBM/L2 -> registers -> LDS, LDS fragment reads,
//   MFMAs.
//
// Schedule (what the first generation lost half its time on, seen in its ISA: an `s_waitcnt lgkmcnt(0)` in front of
// every MFMA pair and exec-mask branches around every column tile): the B fragments of column tile j + 1 are read while
// the six products of tile j run (two register sets, static indices), the column-tile count is a scalar
// (readfirstlane) so its tests are s_cbranch, and the tile loop is fully unrolled.
//
// Geometry: 8 waves, 128 token rows x all F (<= 448) columns, one workgroup per CU (117 KB of LDS); wave = 32 rows x up to
// 7 column tiles; the two waves of a SIMD take the two column groups of the same rows (7 + 6 tiles at F = 400: every SIMD
// carries 13).  Measured alternatives (tools/bench_conv3f.py, round 2): 4-wave workgroups on column halves, two per CU,
// +17 % time (the embedding rows are staged twice); 4 waves of 64 x 224 with 512 registers, one per SIMD, +15 %; opposite
// half-step order on the two waves of a SIMD +5 %; s_setprio around the MFMAs +2 %.  Timing-only ablation (OPT bits, wrong
// results): no global loads -7 %, no LDS stores -6 % more, no barrier -0.5 %, no B fragment reads -7 %: 80 % of the time is
// MFMA issue at the clock the chip holds under this load.
//
// Row quantisation: tiles of 128 rows on 256 CUs come in rounds; a last partial round of few tiles (514 tiles = 2.008
// rounds at config 2) would cost a whole round.  Tiles beyond the last whole round are therefore cut along k into S
// slices (S = min(8, CUs / tiles left)) that run side by side and leave raw partial sums in slabs; conv3_fwd2_fix sums
// the slabs of a tile in slice order (bitwise reproducible) and applies the epilogue.
#include <type_traits>

#include "bf16_mfma.h"
#include "lstur_internal.h"

namespace lstur {

constexpr int F2_BK = 16;

int conv3_dp(int D) { return (D + 7) / 8 * 8; }
static int f2_kp(int D, int ksize) { return cdiv(ksize * conv3_dp(D), F2_BK) * F2_BK; }
static int f2_np(int F) { return cdiv(F, 32) * 32; }
// bf16 elements of E3 for n_titles titles (+ slack: the k tail of the last window and clamped loads stay inside)
size_t conv3_e3_elems(int n_titles, int T, int D, int ksize) {
  return (size_t)n_titles * (T + ksize - 1) * conv3_dp(D) * 3 + 3 * 256;
}
size_t conv3_w3_elems2(int D, int ksize, int F) { return (size_t)3 * f2_kp(D, ksize) * f2_np(F); }
void conv3_w2_dims(int D, int ksize, int F, int* Kp, int* Np) {
  *Kp = f2_kp(D, ksize);
  *Np = f2_np(F);
}

// W [ksize * D, F] fp32 -> three bf16 parts [3][Kp][Np] in k' = tap * Dp + d order, zero padded
__global__ void conv3_pack_w2_kernel(const float* __restrict__ W, int D, int Dp, int ksize, int F, int Kp, int Np,
                                     unsigned short* __restrict__ W3) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)Kp * Np) return;
  const int n = (int)(i % Np), kp = (int)(i / Np);
  const int tap = kp / Dp, d = kp - tap * Dp;
  const float v = (tap < ksize && d < D && n < F) ? W[((size_t)tap * D + d) * F + n] : 0.f;
  u16x4 h, m, l;
  split3(make_float4(v, 0.f, 0.f, 0.f), h, m, l);
  W3[i] = h[0];
  W3[(size_t)Kp * Np + i] = m[0];
  W3[(size_t)2 * Kp * Np + i] = l[0];
}
void launch_conv3_pack_w2(const float* W, int D, int ksize, int F, unsigned short* W3, hipStream_t s) {
  const int Kp = f2_kp(D, ksize), Np = f2_np(F);
  const size_t total = (size_t)Kp * Np;
  hipLaunchKernelGGL(conv3_pack_w2_kernel, dim3((unsigned)cdivz(total, 256)), dim3(256), 0, s, W, D, conv3_dp(D), ksize, F,
                     Kp, Np, W3);
}

__host__ __device__ constexpr int f2_rows(int RWM) { return 32 * RWM; }   // RWM = row waves x row tiles per wave
__host__ __device__ constexpr int f2_sb(int CW, int NTW) { return tr_stride(CW * NTW * 32); }
// one stage: three A images [BM][16] (32-byte rows, 16-byte halves swizzled by row bit 3) + three B images [16][SB]
__host__ __device__ constexpr int f2_stage(int RW, int CW, int NTW) {
  return 3 * f2_rows(RW) * F2_BK + 3 * F2_BK * f2_sb(CW, NTW) + 64 * 8;   // + a 16-byte trash slot per lane
}
__host__ __device__ constexpr size_t f2_lds(int RW, int CW, int NTW) {
  return 2 * f2_stage(RW, CW, NTW) * sizeof(unsigned short) + 2 * f2_rows(RW) * sizeof(int32_t);
}

// how the row tiles of a launch are scheduled (same arithmetic in the kernel, the fix-up kernel and the launcher)
struct TailPlan {
  int full;   // tiles [0, full): one workgroup each, whole rounds of ncu
  int S;      // tiles [full, ntiles): S k-slices each (1: plain tiles)
};
__host__ __device__ inline TailPlan tail_plan(int count, int BM, int ncu) {
  const int ntiles = (count + BM - 1) / BM;
  if (ncu <= 0) return TailPlan{ntiles, 1};            // no slicing (no scratch)
  const int full = ntiles / ncu * ncu, rem = ntiles - full;
  TailPlan p{full, 1};
  if (rem > 0 && rem * 2 <= ncu) p.S = min(8, ncu / rem);
  if (p.S <= 1) p.full = ntiles;
  return p;
}

template <int RW, int CW, int NTW, int MT, int OPT = 0>
__global__ __launch_bounds__(RW* CW * 64, MT == 1 ? 2 : 1) void conv3_fwd2_kernel(
    const unsigned short* __restrict__ E3, uint32_t e_bytes, const unsigned short* __restrict__ W3,
    const float* __restrict__ bias, const int32_t* __restrict__ ids, float* __restrict__ C,
    const int32_t* __restrict__ vidx, const int32_t* __restrict__ vcount, int M, int T, int Tp, int Dp8, int F, int Kp,
    int Np, int ncu, float* __restrict__ slabs) {
  constexpr int BM = f2_rows(RW * MT), NTH = RW * CW * 64, SB = f2_sb(CW, NTW), BCOLS = CW * NTW * 32;
  constexpr int A_IMG = BM * F2_BK, B_IMG = F2_BK * SB, STAGE = f2_stage(RW * MT, CW, NTW);
  constexpr int A_PCS = BM * 6, A_PER = (A_PCS + NTH - 1) / NTH;     // 16-byte pieces (row, k-half, part) of a step
  constexpr int B_CH = 3 * F2_BK * (BCOLS / 8), B_PER = (B_CH + NTH - 1) / NTH;
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  int32_t* rowm = (int32_t*)(smem + 2 * STAGE);           // output row of each tile row (-1: none)
  float* rowkeep = (float*)(rowm + BM);                   // 1 / 0: id != 0
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // pieces past the end of a tile image go to a per-lane trash slot: the store phase has no exec-mask branch
  const int trash = STAGE - 64 * 8 + lane * 8;
  const int count = vcount ? vcount[0] : M;
  // block -> (row tile, k slice): whole rounds of `ncu` tiles first, then the slices of the tiles left over
  const TailPlan tp = tail_plan(count, BM, ncu);
  int tile = blockIdx.x, ks = 0;
  if (tile >= tp.full) {
    const int r = tile - tp.full;
    tile = tp.full + r / tp.S;
    ks = r - (r / tp.S) * tp.S;
  }
  const int m0 = tile * BM;
  if (m0 >= count) return;
  const bool sliced = tile >= tp.full && tp.S > 1;
  constexpr int cb = 0;
  const int NT = Np / 32;                                  // column tiles of 32
  const int col0 = cb * BCOLS;                             // first column of this workgroup
  // waves w and w + NW/2 share a SIMD: give them the SAME rows and the two column groups, so that with an odd number of
  // column tiles (F = 400: 7 + 6) every SIMD carries the same 13 tiles per step instead of 14 on two SIMDs and 12 on two
  const int wm = wv % RW, wn = wv / RW;
  const int nt_lo = col0 / 32 + wn * NTW;
  const int my_nt = max(0, min(NTW, NT - nt_lo));          // scalar (wave-uniform by construction)

  // A loader role: piece idx = row * 6 + q, q = k-half * 3 + part: the six pieces of a row and step are 96 contiguous bytes
  uint32_t a_off[A_PER];
  int a_lds[A_PER];
#pragma unroll
  for (int p = 0; p < A_PER; ++p) {
    const int idx = tid + p * NTH, arow = idx / 6, q = idx - arow * 6, seg = q / 3, part = q - seg * 3;
    const bool use = A_PCS % NTH == 0 || idx < A_PCS;
    const int i = min(m0 + min(arow, BM - 1), count - 1);
    const int m = vidx ? vidx[i] : i;
    const int title = m / T, t = m - title * T;
    a_off[p] = use ? ((uint32_t)(title * Tp + t) * (uint32_t)Dp8 * 3u + (uint32_t)q) * 16u : kBufOob;
    a_lds[p] = use ? part * A_IMG + arow * F2_BK + ((seg ^ ((arow >> 3) & 1)) << 3) : trash;
    if (use && q == 0) {
      rowm[arow] = (m0 + arow < count) ? m : -1;
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
    const int img = idx / (F2_BK * C8), rem = idx - img * (F2_BK * C8), krow = rem / C8, ch = rem - krow * C8;
    const bool use = idx < B_CH && col0 + ch * 8 < Np;
    b_goff[i] = use ? (((uint32_t)img * (uint32_t)Kp + (uint32_t)krow) * (uint32_t)Np + (uint32_t)(col0 + ch * 8)) * 2u : kBufOob;
    b_loff[i] = use ? 3 * A_IMG + img * B_IMG + krow * SB + ch * 8 : trash;
  }
  const __amdgpu_buffer_rsrc_t rsE = make_rsrc(E3, e_bytes);
  const __amdgpu_buffer_rsrc_t rsW = make_rsrc(W3, (uint32_t)((size_t)3 * Kp * Np * 2));

  f32x16 acc[MT][NTW];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[A_PER];
  float4 rb[B_PER];
  int it0_ = 0;
  auto load = [&](int k0) {
    if ((OPT & 4) && k0 >= (it0_ + 2) * F2_BK) return;     // timing-only ablation: no global loads inside the loop
#pragma unroll
    for (int p = 0; p < A_PER; ++p) ra[p] = buf_load4(rsE, a_off[p], (uint32_t)k0 * 6u);
#pragma unroll
    for (int i = 0; i < B_PER; ++i) rb[i] = buf_load4(rsW, b_goff[i], (uint32_t)k0 * (uint32_t)Np * 2u);
  };
  auto store = [&](int buf) {
    unsigned short* st = smem + buf * STAGE;
#pragma unroll
    for (int p = 0; p < A_PER; ++p) *(float4*)(st + a_lds[p]) = ra[p];
#pragma unroll
    for (int i = 0; i < B_PER; ++i) *(float4*)(st + b_loff[i]) = rb[i];
  };

  const int nit_all = Kp / F2_BK;
  const int it0 = sliced ? ks * nit_all / tp.S : 0, it1 = sliced ? (ks + 1) * nit_all / tp.S : nit_all;
  it0_ = it0;
  const int nit = it1 - it0;                       // >= 1 for nit_all >= S (S <= 8; guarded by the launcher)
  load(it0 * F2_BK);
  store(0);
  if (nit > 1) load((it0 + 1) * F2_BK);
  __syncthreads();

  // fragment addresses (elements): A = row-major [row][16] with the k-halves swizzled; B by transposed reads
  const int arow = 32 * MT * wm + (lane & 31);            // (+ 32 i for row tile i: the swizzle bit is unchanged)
  const int a_frag = arow * F2_BK + (((lane >> 5) ^ ((arow >> 3) & 1)) << 3);
  // The k-loop exists once per number NA of live column tiles of the wave (a compile-time constant inside): straight-line
  // code with no test in it, so the compiler counts its LDS reads exactly (lgkmcnt(N), not (0)) and keeps the reads of
  // tile j + 1 in flight under the six products of tile j.
  auto kloop = [&](auto na_tag) {
    constexpr int NA = decltype(na_tag)::value;
    for (int it = 0; it < nit; ++it) {
      if (it + 1 < nit) {
        if (!(OPT & 8) || it == 0) store((it & 1) ^ 1);
        if (it + 2 < nit) load((it0 + it + 2) * F2_BK);
      }
      if constexpr (NA > 0) {
        const unsigned short* st = smem + (it & 1) * STAGE;
        u16x8 ah[MT], am[MT], al[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          ah[i] = *(const u16x8*)(st + a_frag + i * 32 * F2_BK);
          am[i] = *(const u16x8*)(st + A_IMG + a_frag + i * 32 * F2_BK);
          al[i] = *(const u16x8*)(st + 2 * A_IMG + a_frag + i * 32 * F2_BK);
        }
        const unsigned short* bp = st + 3 * A_IMG;
        auto frags = [&](int j, u16x8 (&b)[3]) {
          const int x0 = (wn * NTW + j) * 32;                // column inside this workgroup's window
          b[0] = tr_frag(bp, SB, 0, x0, lane);
          b[1] = tr_frag(bp + B_IMG, SB, 0, x0, lane);
          b[2] = tr_frag(bp + 2 * B_IMG, SB, 0, x0, lane);
        };
        auto six = [&](int j, const u16x8 (&b)[3]) {
          if (OPT & 64) {   // timing-only probe (WRONG results): every 32x32x16 product as two 16x16x32 ones -- the same
                            // FLOPs, cycles, operand registers and LDS reads; does the chip hold a higher clock on this shape?
#pragma unroll
            for (int i = 0; i < MT; ++i) {
              f32x4v q[4];
#pragma unroll
              for (int x = 0; x < 4; ++x) q[x] = {acc[i][j][4 * x], acc[i][j][4 * x + 1], acc[i][j][4 * x + 2], acc[i][j][4 * x + 3]};
              q[0] = mfma16b(al[i], b[0], q[0]); q[1] = mfma16b(al[i], b[0], q[1]);
              q[2] = mfma16b(ah[i], b[2], q[2]); q[3] = mfma16b(ah[i], b[2], q[3]);
              q[0] = mfma16b(am[i], b[1], q[0]); q[1] = mfma16b(am[i], b[1], q[1]);
              q[2] = mfma16b(am[i], b[0], q[2]); q[3] = mfma16b(am[i], b[0], q[3]);
              q[0] = mfma16b(ah[i], b[1], q[0]); q[1] = mfma16b(ah[i], b[1], q[1]);
              q[2] = mfma16b(ah[i], b[0], q[2]); q[3] = mfma16b(ah[i], b[0], q[3]);
#pragma unroll
              for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[i][j][4 * x + y] = q[x][y];
            }
            return;
          }
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            f32x16 c = acc[i][j];
            c = mfma32b(al[i], b[0], c);       // small terms first
            c = mfma32b(ah[i], b[2], c);
            c = mfma32b(am[i], b[1], c);
            c = mfma32b(am[i], b[0], c);
            c = mfma32b(ah[i], b[1], c);
            c = mfma32b(ah[i], b[0], c);
            acc[i][j] = c;
          }
        };
        u16x8 bA[3], bB[3];
        if (OPT & 32) {                      // timing-only ablation: no B fragment reads (operands = whatever A holds)
#pragma unroll
          for (int q = 0; q < 3; ++q) bA[q] = bB[q] = ah[0];
#pragma unroll
          for (int j = 0; j < NA; ++j) {
            asm volatile("" : "+v"(bA[0]), "+v"(bA[1]), "+v"(bA[2]));
            six(j, bA);
          }
        } else {
        frags(0, bA);
#pragma unroll
        for (int j = 0; j < NA; j += 2) {
          if (j + 1 < NA) frags(j + 1, bB);
          six(j, bA);
          if (j + 2 < NA) frags(j + 2, bA);
          if (j + 1 < NA) six(j + 1, bB);
        }
        }
      }
      if (!(OPT & 16)) __syncthreads();
    }
  };
  static_assert(NTW <= 7, "one k-loop instance per live tile count");
  switch (my_nt) {
    case 7: if constexpr (NTW >= 7) kloop(std::integral_constant<int, 7>()); break;
    case 6: if constexpr (NTW >= 6) kloop(std::integral_constant<int, 6>()); break;
    case 5: if constexpr (NTW >= 5) kloop(std::integral_constant<int, 5>()); break;
    case 4: if constexpr (NTW >= 4) kloop(std::integral_constant<int, 4>()); break;
    case 3: if constexpr (NTW >= 3) kloop(std::integral_constant<int, 3>()); break;
    case 2: if constexpr (NTW >= 2) kloop(std::integral_constant<int, 2>()); break;
    case 1: kloop(std::integral_constant<int, 1>()); break;
    default: kloop(std::integral_constant<int, 0>()); break;
  }

  const int hh = lane >> 5, c32 = lane & 31;
  if (sliced) {
    // raw partial sums of this k slice: slab [(tile - full) * S + ks][BM][BCOLS]
    float* slab = slabs + ((size_t)(tile - tp.full) * tp.S + ks) * (size_t)(BM * BCOLS);
    const __amdgpu_buffer_rsrc_t rsS = make_rsrc(slab, (uint32_t)(BM * BCOLS * 4));
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const int col = (wn * NTW + j) * 32 + c32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int lr = 32 * (MT * wm + i) + (r & 3) + 8 * (r >> 2) + 4 * hh;
          buf_store1(acc[i][j][r], rsS, j < my_nt ? (uint32_t)(lr * BCOLS + col) * 4u : kBufOob, 0);
        }
      }
    return;
  }
  // epilogue: + bias, ReLU, x (id != 0); rows scattered to the token's place, branch-free through a buffer descriptor
  const __amdgpu_buffer_rsrc_t rsC = make_rsrc(C, (uint32_t)std::min((size_t)M * F * 4, (size_t)0xFFFFFFFFu));
  const __amdgpu_buffer_rsrc_t rsBias = make_rsrc(bias, (uint32_t)F * 4u);
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    int orow[16];
    float okeep[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int lr = 32 * (MT * wm + i) + (r & 3) + 8 * (r >> 2) + 4 * hh;
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
        buf_store1(fmaxf(acc[i][j][r] + bv, 0.f) * okeep[r], rsC,
                   (cok && orow[r] >= 0) ? ((uint32_t)orow[r] * (uint32_t)F + (uint32_t)col) * 4u : kBufOob, 0);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// A third take on the forward, for ksize == 3 and Dp % 16 == 0 (the reference's Conv1D(400, 3) on 300-d embeddings) -- OPT-IN
// (variant 2 / LSTUR_CONV_FWD3=1: measured slower, see the end of this comment): the TAPS SHARE the embedding rows on chip.  A token's window is the three padded rows g, g + 1, g + 2 of its title;
// conv3_fwd2 fetches the window of every token on its own, so a row that three neighbouring tokens cover is fetched three
// times, ~19 k-steps apart -- from L2 / the Infinity Cache at best (PMC: 407 MB fetched for 133 MB of embeddings).  Here
// the contraction runs chunk-major: for each 16-wide slice c of the embedding dimension the DISTINCT rows of the tile
// (128 tokens in 5-8 titles: ~140 rows; 384 when every token stands alone) are staged ONCE as an LDS image
// [part][row][16], and the three taps read it at rows lrow, lrow + 1, lrow + 2 (lrow: rank of the token's row g among the
// tile's rows -- the rows a token needs are consecutive in E3 and therefore consecutive in the image).  The weight slice
// of step (c, tap) is rows tap * Dp + 16 c .. + 16 of the same W3 image as before.  A fetch: 1/3 .. 1/2.7 of before, and
// as many fewer 16-byte LDS stores.  The image is single-buffered (384 rows x 3 parts = 36 KB; the weight stages stay
// double-buffered: 92 KB): the next slice's rows wait in registers and are stored behind a barrier of their own when
// the last tap of a slice is through -- 19 extra barriers per tile.
// Everything else (tile plan, k-slices of the tail tiles, column groups, epilogue) is conv3_fwd2's.
// Measured (tools/bench_conv3f.py, same box): 249 us against 221 us for conv3_fwd2 (a first form that kept a whole slice's
// pieces in registers and stored them behind an extra barrier: 223-226 us).  Both forms sit at the 256-register limit and
// the compiler gives up the counted LDS waits of the B-fragment double buffer (lgkmcnt(0) in front of most MFMA groups):
// the kernel is bound by MFMA issue, not by the bytes the sharing saves.
// ---------------------------------------------------------------------------------------------
constexpr int F3_AROWS = 384;                                 // rows of the shared A image: 128 tokens x 3 taps at most
constexpr int F3_AIMG = F3_AROWS * F2_BK;
// one weight stage: three images [16][SB] + a 16-byte trash slot per lane (pieces past a tile image: no exec-mask branches)
__host__ __device__ constexpr int f3_stage_b(int CW, int NTW) { return 3 * F2_BK * f2_sb(CW, NTW) + 64 * 8; }
__host__ __device__ constexpr size_t f3_lds(int CW, int NTW) {
  return (3 * F3_AIMG + 2 * f3_stage_b(CW, NTW)) * sizeof(unsigned short) + (3 * 128 + F3_AROWS) * sizeof(int32_t);
}

template <int CW, int NTW>
__global__ __launch_bounds__(4 * CW * 64, 2) void conv3_fwd3_kernel(
    const unsigned short* __restrict__ E3, uint32_t e_bytes, const unsigned short* __restrict__ W3,
    const float* __restrict__ bias, const int32_t* __restrict__ ids, float* __restrict__ C,
    const int32_t* __restrict__ vidx, const int32_t* __restrict__ vcount, int M, int T, int Tp, int Dp8, int F, int Kp,
    int Np, int ncu, float* __restrict__ slabs) {
  constexpr int RW = 4, BM = 128, NTH = RW * CW * 64, SB = f2_sb(CW, NTW), BCOLS = CW * NTW * 32;
  constexpr int B_IMG = F2_BK * SB, STAGE_B = f3_stage_b(CW, NTW);
  constexpr int A_SUB = (F3_AROWS * 6 + 3 * NTH - 1) / (3 * NTH);     // 16-byte pieces (row, k-half, part) per thread and STEP:
                                                                      // a slice's pieces go out in three sub-batches
  constexpr int B_CH = 3 * F2_BK * (BCOLS / 8), B_PER = (B_CH + NTH - 1) / NTH;
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  unsigned short* aimg = smem;                                        // [3][F3_AROWS][16]
  unsigned short* bst = smem + 3 * F3_AIMG;                           // two weight stages
  int32_t* rowm = (int32_t*)(bst + 2 * STAGE_B);                      // output row of each tile row (-1: none)
  float* rowkeep = (float*)(rowm + BM);                               // 1 / 0: id != 0
  int32_t* lrow = (int32_t*)(rowkeep + BM);                           // image row of the token's first tap
  int32_t* grow = lrow + BM;                                          // E3 row of every image row
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int trash_b = STAGE_B - 64 * 8 + lane * 8;                    // inside a weight stage
  const int trash = 3 * F3_AIMG + trash_b;                            // (the A loader borrows stage 0's)
  const int count = vcount ? vcount[0] : M;
  const TailPlan tp = tail_plan(count, BM, ncu);
  int tile = blockIdx.x, ks = 0;
  if (tile >= tp.full) {
    const int r = tile - tp.full;
    tile = tp.full + r / tp.S;
    ks = r - (r / tp.S) * tp.S;
  }
  const int m0 = tile * BM;
  if (m0 >= count) return;
  const bool sliced = tile >= tp.full && tp.S > 1;
  const int NT = Np / 32;
  const int wm = wv % RW, wn = wv / RW;
  const int nt_lo = wn * NTW;
  const int my_nt = max(0, min(NTW, NT - nt_lo));                     // scalar (wave-uniform by construction)
  const int Dp = Dp8 * 8;

  // ---- the tile's tokens, their E3 rows and the image rows they map to
  if (tid < BM) {
    const int i = min(m0 + tid, count - 1);
    const int m = vidx ? vidx[i] : i;
    const int title = m / T, t = m - title * T;
    rowm[tid] = (m0 + tid < count) ? m : -1;
    rowkeep[tid] = ids[m] != 0 ? 1.f : 0.f;
    grow[tid] = title * Tp + t;                                        // (scratch: the tokens' rows g, overwritten below)
  }
  __syncthreads();
  int my_l = 0, my_g = 0;
  if (tid < BM) {
    my_g = grow[tid];
    int dlt = tid == 0 ? 0 : min(3, my_g - grow[tid - 1]);            // new image rows this token brings (0 past the count)
    // inclusive scan over the 128 tokens: inside each wave by shuffles, wave 1 adds wave 0's total
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(dlt, o, 64);
      if (lane >= o) dlt += v;
    }
    my_l = dlt;
    if (tid == 63) lrow[0] = dlt;                                      // (scratch: wave 0's total)
  }
  __syncthreads();
  if (tid >= 64 && tid < BM) my_l += lrow[0];
  __syncthreads();
  if (tid < BM) {
    lrow[tid] = my_l;
    grow[my_l] = my_g;                                                 // consecutive E3 rows: every image row is written
    grow[my_l + 1] = my_g + 1;                                         // with the same value by every token that covers it
    grow[my_l + 2] = my_g + 2;
  }
  __syncthreads();
  const int R = lrow[BM - 1] + 3;                                      // image rows of this tile (<= 384)
  // A loader role: piece idx = row * 6 + q, q = k-half * 3 + part: the six pieces of a row and slice are 96 contiguous bytes.
  // Sub-batch b (one per step of the slice before) carries pieces idx = tid + (b * A_SUB + u) * NTH.
  uint32_t a_off[3][A_SUB];
#pragma unroll
  for (int b = 0; b < 3; ++b)
#pragma unroll
    for (int u = 0; u < A_SUB; ++u) {
      const int idx = tid + (b * A_SUB + u) * NTH, arow = idx / 6, q = idx - arow * 6;
      a_off[b][u] = arow < R ? ((uint32_t)grow[min(arow, F3_AROWS - 1)] * (uint32_t)Dp8 * 3u + (uint32_t)q) * 16u : kBufOob;
    }
  // A tile with at most half the image's rows (the usual case: ~140) keeps TWO slices in it, rows [0, 192) and [192, 384):
  // the next slice is stored beside the one in use, a sub-batch per step, with no barrier of its own.  A tile with more rows
  // (tokens scattered over many tiny titles) loads each slice in the open behind its last tap: correct, slow, rare.
  const bool dbl = R <= F3_AROWS / 2;
  constexpr int A_HALF = F3_AROWS / 2 * F2_BK;
  auto a_dst = [&](int b, int u, int base) {     // LDS element of piece (b, u); past the tile's rows: the trash slot
    const int idx = tid + (b * A_SUB + u) * NTH, arow = idx / 6, q = idx - arow * 6, seg = q / 3, part = q - seg * 3;
    return arow < R ? base + part * F3_AIMG + arow * F2_BK + ((seg ^ ((arow >> 3) & 1)) << 3) : trash;
  };
  // B loader role: pieces of 16 bytes (8 columns) of the 3 x [16][BCOLS] window of the weight parts
  uint32_t b_goff[B_PER];
  int b_loff[B_PER];
#pragma unroll
  for (int i = 0; i < B_PER; ++i) {
    const int idx = tid + i * NTH;
    constexpr int C8 = BCOLS / 8;
    const int img = idx / (F2_BK * C8), rem = idx - img * (F2_BK * C8), krow = rem / C8, ch = rem - krow * C8;
    const bool use = idx < B_CH && ch * 8 < Np;
    b_goff[i] = use ? (((uint32_t)img * (uint32_t)Kp + (uint32_t)krow) * (uint32_t)Np + (uint32_t)(ch * 8)) * 2u : kBufOob;
    b_loff[i] = use ? img * B_IMG + krow * SB + ch * 8 : trash_b;
  }
  const __amdgpu_buffer_rsrc_t rsE = make_rsrc(E3, e_bytes);
  const __amdgpu_buffer_rsrc_t rsW = make_rsrc(W3, (uint32_t)((size_t)3 * Kp * Np * 2));

  f32x16 acc[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  float4 ra[A_SUB];
  float4 rb[B_PER];
  // step s = 3 c + tap: slice c of the embedding dimension (16 wide), weight rows tap * Dp + 16 c
  auto load_a = [&](int c, auto b_tag) {
    constexpr int b = decltype(b_tag)::value;
#pragma unroll
    for (int u = 0; u < A_SUB; ++u) ra[u] = buf_load4(rsE, a_off[b][u], (uint32_t)c * 96u);
  };
  auto store_a = [&](int c, auto b_tag) {
    constexpr int b = decltype(b_tag)::value;
    const int base = (dbl && (c & 1)) ? A_HALF : 0;
#pragma unroll
    for (int u = 0; u < A_SUB; ++u) *(float4*)(smem + a_dst(b, u, base)) = ra[u];
  };
  typedef std::integral_constant<int, 0> B0;
  typedef std::integral_constant<int, 1> B1;
  typedef std::integral_constant<int, 2> B2;
  auto fill_a = [&](int c) {            // a whole slice in the open (the first one; every one of a tile with > 192 rows)
    load_a(c, B0()); store_a(c, B0());
    load_a(c, B1()); store_a(c, B1());
    load_a(c, B2()); store_a(c, B2());
  };
  auto load_b = [&](int s) {
    const int c = s / 3, tap = s - c * 3;
#pragma unroll
    for (int i = 0; i < B_PER; ++i) rb[i] = buf_load4(rsW, b_goff[i], (uint32_t)(tap * Dp + c * F2_BK) * (uint32_t)Np * 2u);
  };
  auto store_b = [&](int buf) {
    unsigned short* st = bst + buf * STAGE_B;
#pragma unroll
    for (int i = 0; i < B_PER; ++i) *(float4*)(st + b_loff[i]) = rb[i];
  };

  // k-slices of the tail tiles are cut at slice boundaries (whole triples of steps)
  const int nch = Kp / (3 * F2_BK);                                    // Dp / 16 slices
  const int c0 = sliced ? ks * nch / tp.S : 0, c1 = sliced ? (ks + 1) * nch / tp.S : nch;
  const int s0 = 3 * c0, s1 = 3 * c1;
  if (c0 < c1) {
    fill_a(c0);
    load_b(s0);
    store_b(s0 & 1);
    load_b(s0 + 1);
    if (dbl && c0 + 1 < c1) load_a(c0 + 1, B0());
  }
  __syncthreads();

  const int tok_row = 32 * wm + (lane & 31), hh = lane >> 5;
  const int lr = lrow[tok_row];
  auto kloop = [&](auto na_tag) {
    constexpr int NA = decltype(na_tag)::value;
    auto step = [&](int c, auto tap_tag) {
      constexpr int tap = decltype(tap_tag)::value;
      const int s = 3 * c + tap;
      const bool more = c + 1 < c1;                // another slice follows: its sub-batch `tap` is in ra (dbl)
      if (s + 1 < s1) {
        store_b((s + 1) & 1);
        if (s + 2 < s1) load_b(s + 2);
      }
      if (dbl && more) {
        store_a(c + 1, tap_tag);
        if (tap == 0) load_a(c + 1, B1());
        if (tap == 1) load_a(c + 1, B2());
        if (tap == 2 && c + 2 < c1) load_a(c + 2, B0());
      }
      if constexpr (NA > 0) {
        const int rho = lr + tap;
        const unsigned short* ap = aimg + ((dbl && (c & 1)) ? A_HALF : 0) + rho * F2_BK + ((hh ^ ((rho >> 3) & 1)) << 3);
        const u16x8 ah = *(const u16x8*)ap, am = *(const u16x8*)(ap + F3_AIMG), al = *(const u16x8*)(ap + 2 * F3_AIMG);
        const unsigned short* bp = bst + (s & 1) * STAGE_B;
        auto frags = [&](int j, u16x8 (&b)[3]) {
          const int x0 = (wn * NTW + j) * 32;
          b[0] = tr_frag(bp, SB, 0, x0, lane);
          b[1] = tr_frag(bp + B_IMG, SB, 0, x0, lane);
          b[2] = tr_frag(bp + 2 * B_IMG, SB, 0, x0, lane);
        };
        auto six = [&](int j, const u16x8 (&b)[3]) {
          f32x16 cc = acc[j];
          cc = mfma32b(al, b[0], cc);       // small terms first
          cc = mfma32b(ah, b[2], cc);
          cc = mfma32b(am, b[1], cc);
          cc = mfma32b(am, b[0], cc);
          cc = mfma32b(ah, b[1], cc);
          cc = mfma32b(ah, b[0], cc);
          acc[j] = cc;
        };
        u16x8 bA[3], bB[3];
        frags(0, bA);
#pragma unroll
        for (int j = 0; j < NA; j += 2) {
          if (j + 1 < NA) frags(j + 1, bB);
          six(j, bA);
          if (j + 2 < NA) frags(j + 2, bA);
          if (j + 1 < NA) six(j + 1, bB);
        }
      }
      __syncthreads();
    };
    for (int c = c0; c < c1; ++c) {
      step(c, B0());
      step(c, B1());
      step(c, B2());
      if (!dbl && c + 1 < c1) {          // (block-uniform) the image makes room for the next slice
        fill_a(c + 1);
        __syncthreads();
      }
    }
  };
  static_assert(NTW <= 7, "one k-loop instance per live tile count");
  switch (my_nt) {
    case 7: if constexpr (NTW >= 7) kloop(std::integral_constant<int, 7>()); break;
    case 6: if constexpr (NTW >= 6) kloop(std::integral_constant<int, 6>()); break;
    case 5: if constexpr (NTW >= 5) kloop(std::integral_constant<int, 5>()); break;
    case 4: if constexpr (NTW >= 4) kloop(std::integral_constant<int, 4>()); break;
    case 3: if constexpr (NTW >= 3) kloop(std::integral_constant<int, 3>()); break;
    case 2: if constexpr (NTW >= 2) kloop(std::integral_constant<int, 2>()); break;
    case 1: kloop(std::integral_constant<int, 1>()); break;
    default: kloop(std::integral_constant<int, 0>()); break;
  }

  const int c32 = lane & 31;
  if (sliced) {
    float* slab = slabs + ((size_t)(tile - tp.full) * tp.S + ks) * (size_t)(BM * BCOLS);
    const __amdgpu_buffer_rsrc_t rsS = make_rsrc(slab, (uint32_t)(BM * BCOLS * 4));
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const int col = (wn * NTW + j) * 32 + c32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lr2 = 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * hh;
        buf_store1(acc[j][r], rsS, j < my_nt ? (uint32_t)(lr2 * BCOLS + col) * 4u : kBufOob, 0);
      }
    }
    return;
  }
  const __amdgpu_buffer_rsrc_t rsC = make_rsrc(C, (uint32_t)std::min((size_t)M * F * 4, (size_t)0xFFFFFFFFu));
  const __amdgpu_buffer_rsrc_t rsBias = make_rsrc(bias, (uint32_t)F * 4u);
  int orow[16];
  float okeep[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int lr2 = 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * hh;
    orow[r] = rowm[lr2];
    okeep[r] = rowkeep[lr2];
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

// Epilogue of the k-sliced tiles: out[token] = relu(bias + sum over slices, in slice order) x (id != 0).
// grid (ncu / 2 tiles at most, 64 row pairs); one thread = one column quad of one row, all slice loads in flight at once
__global__ __launch_bounds__(256) void conv3_fwd2_fix_kernel(const float* __restrict__ slabs, const float* __restrict__ bias,
                                                             const int32_t* __restrict__ ids, float* __restrict__ C,
                                                             const int32_t* __restrict__ vidx,
                                                             const int32_t* __restrict__ vcount, int M, int F, int ncu) {
  constexpr int BM = 128, BCOLS = 448;
  const int count = vcount ? vcount[0] : M;
  const TailPlan tp = tail_plan(count, BM, ncu);
  if (tp.S <= 1) return;
  // (the number of sliced tiles is a device-side value: a few workgroup columns walk over them -- a grid sized for the most
  //  there can be, ncu / 2, spent ~5 us dispatching 8 000 workgroups that returned at once)
  for (int bt = blockIdx.x; (tp.full + bt) * BM < count; bt += gridDim.x) {
  const int tile = tp.full + bt;
  const int c4 = threadIdx.x & 127, lr = blockIdx.y * 2 + (threadIdx.x >> 7);   // 112 column quads (448 columns), 2 rows
  const int i = tile * BM + lr;
  if (c4 >= BCOLS / 4 || i >= count) continue;
  const int m = vidx ? vidx[i] : i;
  const float* p = slabs + (size_t)bt * tp.S * (BM * BCOLS) + (size_t)lr * BCOLS + c4 * 4;
  float4 v[8];
#pragma unroll
  for (int sl = 0; sl < 8; ++sl) v[sl] = *(const float4*)(p + (size_t)min(sl, tp.S - 1) * (BM * BCOLS));
  float4 a = v[0];
#pragma unroll
  for (int sl = 1; sl < 8; ++sl)
    if (sl < tp.S) {
      a.x += v[sl].x; a.y += v[sl].y; a.z += v[sl].z; a.w += v[sl].w;
    }
  const float keep = ids[m] != 0 ? 1.f : 0.f;
  const int col = c4 * 4;
  const float o[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (col + e < F) C[(size_t)m * F + col + e] = fmaxf(o[e] + bias[col + e], 0.f) * keep;
  }
}

bool conv3f_supported(int D, int ksize, int F) { return D % 4 == 0 && F >= 1 && F <= 448; }

size_t conv3_fwd2_scratch_bytes(int ncu) { return (size_t)ncu * 128 * 448 * sizeof(float); }
static int conv3_ncu() {
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t prop;
    ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
              ? prop.multiProcessorCount : 256;
  }
  return ncu;
}

// variant 0: the product kernel; 10 .. 15: timing-only ablations of it (WRONG results, tools/bench_conv3f.py): 10 no global
// loads in the loop, 11 + no LDS stores, 12 + no barrier, 13 + no B fragment reads, 14 only "no barrier", 15 only "no B reads".
// scratch: conv3_fwd2_scratch_bytes(CUs) for the k-sliced tail tiles (NULL: no tail slicing).
void launch_conv3_fwd2(const unsigned short* e3, const unsigned short* W3, const float* bias, const int32_t* ids, float* out,
                       int n_titles, int T, int D, int ksize, int F, const int32_t* vidx, const int32_t* vcount, int variant,
                       void* scratch, hipStream_t s, int cus) {
  const int M = n_titles * T;
  if (M == 0) return;
  const int Tp = T + ksize - 1, Dp = conv3_dp(D), Np = f2_np(F), Kp = f2_kp(D, ksize);
  const uint32_t e_bytes = (uint32_t)std::min(conv3_e3_elems(n_titles, T, D, ksize) * 2, (size_t)0xFFFFFFFFu);
  const int tiles = cdiv(M, 128);
  // tail slicing needs at least one k-step per slice; without scratch every tile is a plain tile (ncu = "infinite")
  // `cus` (0: all): the CUs this launch may count on -- the two-lane step leaves some to the other lane's recurrence, so
  // whole rounds are rounds of `cus` tiles and what is left over is sliced
  const int ncu = (scratch && Kp / F2_BK >= 8) ? std::min(cus > 0 ? std::min(cus, conv3_ncu()) : conv3_ncu(), 256) : 0;   // scratch holds <= 256 slabs; 0: no slicing
  auto kern = variant == 10 ? conv3_fwd2_kernel<4, 2, 7, 1, 4> : variant == 11 ? conv3_fwd2_kernel<4, 2, 7, 1, 12> :
              variant == 12 ? conv3_fwd2_kernel<4, 2, 7, 1, 28> : variant == 13 ? conv3_fwd2_kernel<4, 2, 7, 1, 60> :
              variant == 14 ? conv3_fwd2_kernel<4, 2, 7, 1, 16> : variant == 15 ? conv3_fwd2_kernel<4, 2, 7, 1, 32> :
              variant == 16 ? conv3_fwd2_kernel<4, 2, 7, 1, 64> :
                              conv3_fwd2_kernel<4, 2, 7, 1, 0>;
  // grid: every tile as a plain tile, or whole rounds + at most ncu slices (the device decides from the token count)
  const int grid = tiles + ncu;
  if (ncu <= 0) last_launch();
  // variant 2 (or LSTUR_CONV_FWD3=1 for variant 0): the tap-sharing kernel where it applies -- opt-in: it fetches a third
  // of the embedding bytes and takes 13 % longer
  static const bool want3 = getenv("LSTUR_CONV_FWD3") != nullptr;
  if ((variant == 2 || (variant == 0 && want3)) && ksize == 3 && Dp % 16 == 0) {
    auto k3 = conv3_fwd3_kernel<2, 7>;
    (void)hipFuncSetAttribute((const void*)k3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)f3_lds(2, 7));
    launch_k(k3, dim3(grid), dim3(512), f3_lds(2, 7), s, e3, e_bytes, W3, bias, ids, out, vidx, vcount, M, T, Tp, Dp / 8, F, Kp, Np,
             ncu, (float*)scratch);
  } else {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)f2_lds(4, 2, 7));
    launch_k(kern, dim3(grid), dim3(512), f2_lds(4, 2, 7), s, e3, e_bytes, W3, bias, ids, out, vidx, vcount, M, T, Tp, Dp / 8, F,
             Kp, Np, ncu, (float*)scratch);
  }
  if (ncu > 0) {
    last_launch();
    launch_k(conv3_fwd2_fix_kernel, dim3(8, 64), dim3(256), 0, s, (const float*)scratch, bias, ids, out, vidx, vcount, M, F,
             ncu);
  }
}


// ---------------------------------------------------------------------------------------------
// Kernel gradient dW[k', F] = im2col(E)^T . dC over the (compacted) tokens, bf16x3, second generation: the A operand (the
// tokens' windows) comes PRE-SPLIT from E3 -- 16-byte copies, no conversion arithmetic; the conv-output gradient dC
// (fp32, written by the attention backward) is still split on its way into LDS.  The contraction index is the token and
// both operands lie token-major in HBM, so both LDS tiles are row-major [token][x] images read with transposed LDS reads.
// Workgroup = 8 waves, tile 256 (k') x 224 (F): wave w owns row tile w and up to 7 column tiles; 16 tokens per step.
// Waves whose row tile lies past k' = ksize * Dp and column tiles past F issue NO MFMAs (the first generation computed a
// 1024 x 448 grid for 900 x 400: 19 % more products than needed; here 29 x 13 tiles of 32 x 32 instead of 32 x 14).
// The k-loop exists once per live column-tile count (compile-time): straight-line code, counted LDS waits.
// Tokens are cut into slabs (one per workgroup row) reduced in fixed order by the step's multi_reduce: one job per tap
// (rows d < D of tap j land on conv_w rows j * D + d).
// ---------------------------------------------------------------------------------------------
constexpr int G2_TM = 8, G2_TN = 7, G2_BK = 16, G2_THREADS = G2_TM * 64;
constexpr int G2_BMR = G2_TM * 32, G2_BNC = G2_TN * 32;                       // 256 x 224
constexpr int G2_SA = tr_stride(G2_BMR), G2_SB = tr_stride(G2_BNC);          // 288 / 224 elements
// The three part images of A lie 48 bytes (mod 256) apart: the loader's 16-byte stores go out in w = chunk * 3 + part order, so
// eight consecutive lanes then cover eight different 16-byte bank groups (with images a multiple of 256 bytes apart the three
// parts of a chunk hit the same banks: a 3-way conflict on every A store)
constexpr int G2_A_IMG = G2_BK * G2_SA + 24, G2_B_IMG = G2_BK * G2_SB;
static_assert((G2_BK * G2_SA * 2) % 256 == 0, "part skew assumes image sizes of whole bank rows");
constexpr int G2_STAGE = 3 * G2_A_IMG + 3 * G2_B_IMG + 64 * 4;          // + an 8-byte trash slot per lane
constexpr size_t G2_LDS = 2 * G2_STAGE * sizeof(unsigned short);
constexpr int G2_APC = G2_BK * (G2_BMR / 8) * 3 / G2_THREADS;                 // 3 sixteen-byte pieces per thread and step
constexpr int G2_BPC = (G2_BK * (G2_BNC / 4) + G2_THREADS - 1) / G2_THREADS;  // 2 float4 per thread and step
static_assert(G2_BK * (G2_BMR / 8) * 3 % G2_THREADS == 0, "A pieces divide evenly");

static int g2_mblocks(int Kq) { return cdiv(cdiv(Kq, 32), G2_TM); }
static int g2_nblocks(int F) { return cdiv(cdiv(F, 32), G2_TN); }
// slabs: about one workgroup per CU (or whole multiples), at least 512 tokens each
static int g2_slabs(int M, int Kq, int F, int cus = 256) {
  const int per = g2_mblocks(Kq) * g2_nblocks(F);
  const int rounds = std::max(1, cdiv(M, 120000));                 // ~105 600 tokens at B = 64 -> one round
  // all CUs: whole multiples of 8 slabs (the same share for every XCD); fewer CUs (the other lane's recurrence holds some):
  // as many slabs as fit, a few XCDs then carry one slab less
  const int want = cus >= 256 ? rounds * 256 / per / 8 * 8 : std::max(1, rounds * cus / per);
  return std::max(1, std::min(std::max(8, want), std::max(1, M / 512)));
}
// (three times the slab count of one launch + one: a step may give the candidate titles' and the late history slots' tokens
//  launches of their own and reduce their slabs into one ahead of time)
size_t conv3_wgrad2_scratch_bytes(int n_titles, int T, int D, int ksize, int F) {
  return (size_t)(3 * g2_slabs(n_titles * T, ksize * conv3_dp(D), F) + 1) * ksize * conv3_dp(D) * F * sizeof(float);
}
int conv3_wgrad2_slabs(int plan_titles, int T, int D, int ksize, int F, int cus) {
  return g2_slabs(plan_titles * T, ksize * conv3_dp(D), F, cus > 0 ? std::min(cus, 256) : 256);
}
size_t conv3_wgrad2_slab_floats(int D, int ksize, int F) { return (size_t)ksize * conv3_dp(D) * F; }

// OPT != 0: timing-only ablations (WRONG results; LSTUR_WGRAD_ABLATE, tools/bench_conv3f.py).  SIDE: the same kernel under
// another name -- the launches over a title sub-range that a step runs beside the recurrence show up as their own line in
// rocprof summaries and PMC passes
template <int OPT = 0, bool SIDE = false>
__global__ __launch_bounds__(G2_THREADS, 2) void conv3_wgrad2_kernel(
    const unsigned short* __restrict__ E3, uint32_t e_bytes, const float* __restrict__ G, uint32_t g_bytes,
    float* __restrict__ slabs, const int32_t* __restrict__ vcount, int M, int F, int Kq, int mblocks, int nblocks, int nslabs,
    const int32_t* __restrict__ offA, const int32_t* __restrict__ offB, int range_sel, int slab_base) {
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // tokens [lo, count) of the compacted list: all of it, or a title range between the boundaries the compaction counted
  // (vcount[0] total, [1] history | candidates, [2] early | late history slots of a slot-major step).  range_sel: 0 all,
  // 1 history, 2 candidates, 3 early slots, 4 late slots -- what the recurrence's backward has not reached yet runs later
  const int count = vcount ? vcount[range_sel == 1 || range_sel == 4 ? 1 : range_sel == 3 ? 2 : 0] : M;
  const int lo = !vcount ? 0 : range_sel == 2 ? vcount[1] : range_sel == 4 ? vcount[2] : 0;
  // block -> (slab, m-block, n-block); XCD-aware: workgroup ids go round-robin over the 8 XCDs, so the blocks of one slab
  // get ids 8 apart: they run on ONE XCD at about the same time and share the slab's tokens through its L2
  const int per = mblocks * nblocks;
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int slab = (seq / per) * 8 + xcd, rem = seq - (seq / per) * per, mb = rem / nblocks, nb = rem - mb * nblocks;
  if (slab >= nslabs) return;
  const int per_slab = (count - lo + nslabs - 1) / nslabs, tpad = (per_slab + G2_BK - 1) / G2_BK * G2_BK;
  const int t_begin = lo + slab * tpad, t_end = min(count, t_begin + tpad);
  const int kd0 = mb * G2_BMR, f0 = nb * G2_BNC;
  const __amdgpu_buffer_rsrc_t rsE = make_rsrc(E3, e_bytes), rsG = make_rsrc(G, g_bytes);
  // live work of this wave: its row tile must start inside k' < Kq; column tiles inside F
  const bool row_live = kd0 + wv * 32 < Kq;
  const int na = row_live ? max(0, min(G2_TN, (F - f0 + 31) / 32)) : 0;

  f32x16 acc[G2_TN];
#pragma unroll
  for (int j = 0; j < G2_TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // loader roles: a thread serves ONE token row of the step (row = tid / 32), so a step costs it one index load per operand
  // (offA / offB of its token) -- with pieces dealt out in idx order a thread's pieces lay in up to five different rows,
  // five index loads per step: 4 % of the kernel's time (LSTUR_WGRAD_ABLATE=1).  A: pieces w = c + 32 p (p < 3) of the 96 of
  // its row (w = chunk * 3 + part: 1536 contiguous bytes of the token's window in E3); B: float4 slots c + 32 p (p < 2) --
  // 64 slots for the 56 of a 224-column tile; the other 8 load through the out-of-range offset and store to a trash slot
  static_assert(G2_APC == 3 && G2_BPC == 2 && G2_THREADS == 32 * G2_BK, "row-per-32-threads loader grid");
  const int l_row = tid >> 5, l_c = tid & 31;
  int a_lds[G2_APC];
#pragma unroll
  for (int p = 0; p < G2_APC; ++p) {
    const int w = l_c + 32 * p, chunk = w / 3, part = w - chunk * 3;
    a_lds[p] = part * G2_A_IMG + l_row * G2_SA + chunk * 8;
  }
  const uint32_t a_w16 = (uint32_t)l_c * 16u + (uint32_t)kd0 * 6u;   // byte offset inside the window (kd0 / 8 chunks of 48 bytes)
  int b_lds[G2_BPC];
#pragma unroll
  for (int p = 0; p < G2_BPC; ++p)
    b_lds[p] = (l_c + 32 * p) * 4 < G2_BNC ? 3 * G2_A_IMG + l_row * G2_SB + (l_c + 32 * p) * 4 : G2_STAGE - 64 * 4 + (tid & 63) * 4;
  // Row offsets of a step's pieces come in two halves so that no wait sits between the prefetch loads: `fetch_rows`
  // only issues the index loads (clamped, branch-free), `make_offsets` turns them into byte offsets one step later.
  // A token past the slab's end contributes 0 through the out-of-range offset of G.
  uint32_t a_off, b_off[G2_BPC];
  int a_raw, b_raw;
  auto fetch_rows = [&](int t0) {
    if ((OPT & 1) && t0 > t_begin) return;          // ablation: the row offsets of the first step for every step
    a_raw = offA[min(t0 + l_row, count - 1)];
    b_raw = offB[min(t0 + l_row, count - 1)];
  };
  auto make_offsets = [&](int t0) {
    a_off = (uint32_t)a_raw * 16u + a_w16;                                               // offA: 16-byte pieces
#pragma unroll
    for (int p = 0; p < G2_BPC; ++p) {
      const int col = f0 + (l_c + 32 * p) * 4;
      b_off[p] = (t0 + l_row < t_end && col < F && col < f0 + G2_BNC) ? ((uint32_t)b_raw + (uint32_t)col) * 4u : kBufOob;   // offB: floats
    }
  };
  float4 ra[G2_APC], rb[G2_BPC];
  int loads_done = 0;
  auto load = [&]() {
    if ((OPT & 4) && loads_done >= 2) return;       // ablation: no global loads inside the loop
    ++loads_done;
#pragma unroll
    for (int p = 0; p < G2_APC; ++p) ra[p] = buf_load4(rsE, a_off, (uint32_t)(p * 512));
#pragma unroll
    for (int p = 0; p < G2_BPC; ++p) rb[p] = buf_load4(rsG, b_off[p], 0);
  };
  auto store = [&](int buf) {
    unsigned short* st = smem + buf * G2_STAGE;
#pragma unroll
    for (int p = 0; p < G2_APC; ++p) *(float4*)(st + a_lds[p]) = ra[p];
#pragma unroll
    for (int p = 0; p < G2_BPC; ++p) {
      u16x4 h, m, l;
      if (OPT & 2) {                                // ablation: no split arithmetic
        h = __builtin_bit_cast(u16x4, make_float2(rb[p].x, rb[p].y));
        m = __builtin_bit_cast(u16x4, make_float2(rb[p].z, rb[p].w));
        l = __builtin_bit_cast(u16x4, make_float2(rb[p].y, rb[p].z));
      } else {
        split3(rb[p], h, m, l);
      }
      const bool in_tile = (l_c + 32 * p) * 4 < G2_BNC;
      unsigned short* d = st + b_lds[p];
      *(u16x4*)d = h;
      *(u16x4*)(d + (in_tile ? G2_B_IMG : 0)) = m;
      *(u16x4*)(d + (in_tile ? 2 * G2_B_IMG : 0)) = l;
    }
  };

  const int nit = t_end > t_begin ? (t_end - t_begin + G2_BK - 1) / G2_BK : 0;
  if (nit > 0) {
    fetch_rows(t_begin);
    make_offsets(t_begin);
    load();                                          // step 0
    fetch_rows(t_begin + G2_BK);
    store(0);
    make_offsets(t_begin + G2_BK);
    load();                                          // step 1
    fetch_rows(t_begin + 2 * G2_BK);
    __syncthreads();
  }
  auto kloop = [&](auto na_tag) {
    constexpr int NA = decltype(na_tag)::value;
    for (int it = 0; it < nit; ++it) {
      if (it + 1 < nit) {
        if (!(OPT & 8) || it == 0) store((it & 1) ^ 1);   // step it + 1 (its loads were issued a step ago)
        make_offsets(t_begin + (it + 2) * G2_BK);      // from the rows fetched a step ago
        load();                                        // step it + 2
        fetch_rows(t_begin + (it + 3) * G2_BK);        // consumed in the next step
      }
      if constexpr (NA > 0) {
        const unsigned short* st = smem + (it & 1) * G2_STAGE;
        const unsigned short* bp = st + 3 * G2_A_IMG;
        const u16x8 ah = tr_frag(st, G2_SA, 0, wv * 32, lane);
        const u16x8 am = tr_frag(st + G2_A_IMG, G2_SA, 0, wv * 32, lane);
        const u16x8 al = tr_frag(st + 2 * G2_A_IMG, G2_SA, 0, wv * 32, lane);
        auto frags = [&](int j, u16x8 (&b)[3]) {
          b[0] = tr_frag(bp, G2_SB, 0, j * 32, lane);
          b[1] = tr_frag(bp + G2_B_IMG, G2_SB, 0, j * 32, lane);
          b[2] = tr_frag(bp + 2 * G2_B_IMG, G2_SB, 0, j * 32, lane);
        };
        auto six = [&](f32x16& c, const u16x8 (&b)[3]) {
          c = mfma32b(al, b[0], c);
          c = mfma32b(ah, b[2], c);
          c = mfma32b(am, b[1], c);
          c = mfma32b(am, b[0], c);
          c = mfma32b(ah, b[1], c);
          c = mfma32b(ah, b[0], c);
        };
        u16x8 bA[3], bB[3];
        frags(0, bA);
#pragma unroll
        for (int j = 0; j < NA; j += 2) {
          if (j + 1 < NA) frags(j + 1, bB);
          six(acc[j], bA);
          if (j + 2 < NA) frags(j + 2, bA);
          if (j + 1 < NA) six(acc[j + 1], bB);
        }
      }
      __syncthreads();
    }
  };
  switch (na) {
    case 7: kloop(std::integral_constant<int, 7>()); break;
    case 6: kloop(std::integral_constant<int, 6>()); break;
    case 5: kloop(std::integral_constant<int, 5>()); break;
    case 4: kloop(std::integral_constant<int, 4>()); break;
    case 3: kloop(std::integral_constant<int, 3>()); break;
    case 2: kloop(std::integral_constant<int, 2>()); break;
    case 1: kloop(std::integral_constant<int, 1>()); break;
    default: kloop(std::integral_constant<int, 0>()); break;
  }

  // slab [Kq, F] of this token range (zeros when the range is empty: the reduction reads every slab)
  const int hh = lane >> 5, c32 = lane & 31;
  const __amdgpu_buffer_rsrc_t rsC = make_rsrc(slabs + (size_t)(slab_base + slab) * Kq * F, (uint32_t)((size_t)Kq * F * 4));
#pragma unroll
  for (int j = 0; j < G2_TN; ++j) {
    const int f = f0 + j * 32 + c32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kd = kd0 + wv * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
      buf_store1(acc[j][r], rsC, (f < F && kd < Kq) ? ((uint32_t)kd * (uint32_t)F + (uint32_t)f) * 4u : kBufOob, 0);
    }
  }
}

// dw [ksize * D, F] through jobs[0 .. ksize) (one per tap; slabs in `scratch`, conv3_wgrad2_scratch_bytes); dout rows of pad
// tokens are not read.  offA: per-token window offset in 16-byte pieces of E3 ((title * Tp + t) * Dp / 8 * 3), offB: per-token
// row offset in floats of dout (token * F) -- launch_compact_write_offsets with rowA = Dp / 8 * 3.  Returns the job count.
// A step may cut the token list at a title boundary (range_sel 2: the tokens behind vcount[1], 1: those in front) and give
// each part its own launch: the parts' slabs lie one after the other (slab_base = slabs of the launches before, plan_titles
// = the titles of this part for the slab count) and the LAST launch writes the jobs over all of them (jobs == NULL: none).
int launch_conv3_wgrad2(const unsigned short* e3, const float* dout, float* dw, void* scratch, int n_titles, int T, int D,
                        int ksize, int F, const int32_t* vcount, hipStream_t s, ReduceJob* jobs, const int32_t* offA,
                        const int32_t* offB, float alpha, int cus, int range_sel, int slab_base, int plan_titles, int* slabs_out) {
  const int M = n_titles * T, Dp = conv3_dp(D), Kq = ksize * Dp;
  if (slabs_out) *slabs_out = 0;
  if (M == 0) {
    for (int j = 0; jobs && j < ksize; ++j) jobs[j].nslices = 0;
    return jobs ? ksize : 0;
  }
  static const int ablate = getenv("LSTUR_WGRAD_ABLATE") ? atoi(getenv("LSTUR_WGRAD_ABLATE")) : 0;
  auto kern = ablate == 1 ? conv3_wgrad2_kernel<1> : ablate == 2 ? conv3_wgrad2_kernel<2> : ablate == 3 ? conv3_wgrad2_kernel<3> :
              ablate == 4 ? conv3_wgrad2_kernel<4> : ablate == 7 ? conv3_wgrad2_kernel<7> : ablate == 15 ? conv3_wgrad2_kernel<15> :
              (vcount && (range_sel == 2 || range_sel == 4)) ? conv3_wgrad2_kernel<0, true> : conv3_wgrad2_kernel<0, false>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G2_LDS);
  // (the scratch is sized for the full-chip slab count; fewer CUs -> fewer slabs)
  const int Mp = (plan_titles > 0 ? plan_titles : n_titles) * T;
  const int mblocks = g2_mblocks(Kq), nblocks = g2_nblocks(F), ns = g2_slabs(Mp, Kq, F, cus > 0 ? std::min(cus, 256) : 256);
  const uint32_t e_bytes = (uint32_t)std::min(conv3_e3_elems(n_titles, T, D, ksize) * 2, (size_t)0xFFFFFFFFu);
  const uint32_t g_bytes = (uint32_t)std::min((size_t)M * F * 4, (size_t)0xFFFFFFFFu);
  const dim3 grid(cdiv(ns, 8) * 8 * mblocks * nblocks);
  last_launch();
  launch_k(kern, grid, dim3(G2_THREADS), G2_LDS, s, e3, e_bytes, dout, g_bytes, (float*)scratch,
                     vcount, M, F, Kq, mblocks, nblocks, ns, offA, offB, vcount ? range_sel : 0, slab_base);
  if (slabs_out) *slabs_out = ns;
  if (!jobs) return 0;
  for (int j = 0; j < ksize; ++j)
    jobs[j] = ReduceJob{(const float*)scratch + (size_t)j * Dp * F, dw + (size_t)j * D * F, D, F, F, slab_base + ns, (size_t)Kq * F, alpha, 0.f};
  return ksize;
}

}  // namespace lstur
