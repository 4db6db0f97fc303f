// Conv1D over title tokens as an implicit-im2col fp32 MFMA contraction (gfx950).
//
// Reference op: keras.layers.Conv1D(F, k, padding='same', activation='relu') followed by the
// pad-id mask Lambda (task/paper.py:146-155).  The embedding tensor is materialised ONCE per step in a
// zero-padded per-title layout  E[n][pad_l + T + pad_r][D]  (misc.hip: gather_kernel), so the im2col
// row of output token (n,t) is the CONTIGUOUS window  E_flat[(n*Tp + t)*D .. +k*D)  -- im2col is never
// built, and both the forward (M = tokens, K = k*D, N = F) and the kernel gradient (M = k*D, N = F,
// K = tokens) read plain row-major tiles.
//
// MFMA: v_mfma_f32_16x16x4_f32 (exact fp32, 64 FLOP/clk/SIMD).  Operand lane map: A lane l holds
// A[row l&15][k l>>4], B lane l holds B[k l>>4][col l&15]; C/D lane l, reg j: row (l>>4)*4+j, col l&15.
#include <stdlib.h>

#include "lstur_device.h"
#include "lstur_internal.h"

namespace lstur {

// ---------------------------------------------------------------------------------------------
// Forward.  Workgroup = WAVES waves; wave w owns 16 token rows x NTW column tiles (all of F when
// NTW*16 >= F).  Per 16-deep K chunk: one float4 of A per lane straight from global (each A element is
// reused by the wave's NTW tiles from registers), the packed-W chunk [NTW*16][16] through LDS
// (double buffered, shared by all waves), NTW*4 MFMAs.  The k index inside a chunk is permuted
// identically on both operands (lane kq supplies k = 4kq+s to MFMA s), which is legal for a sum.
// ---------------------------------------------------------------------------------------------
// One pass over K for NT column tiles [first, first+NT) of 16 output rows: the wave's share of a row block.
// All waves of the workgroup run the chunk loop together (W staging + barriers); `active` gates the MFMAs and
// the epilogue only.
// EPI 0: the Conv1D epilogue (+ bias, ReLU, pad-id mask).  EPI 1: the same contraction used as the conv's INPUT
// gradient for a trainable word table (--textual-embedding-trainable): E = zero-padded conv-output gradients, the
// weight image is the flipped kernel, and the epilogue applies the embedding dropout mask and scatter-adds row
// (title, t) into word_grad[id(title, t)] (float atomics, like TF's unsorted_segment_sum on a GPU).
struct EpiDrop { uint32_t thresh, key; float scale; };
template <int NTW, int WAVES, int NT, int EPI>
__device__ __forceinline__ void conv_rows_pass(const float* __restrict__ E, const float* __restrict__ wsrc,
                                               const float* __restrict__ bias, const int32_t* __restrict__ ids,
                                               float* __restrict__ C, const int32_t* __restrict__ vidx, float* smem,
                                               int T, int Tp, int D, int F, int Fp, int nchunk, int valid4, int tile0,
                                               int m0, int r_end, bool active, int first, int tid, int lane,
                                               EpiDrop ed, __amdgpu_buffer_rsrc_t rsC) {
  constexpr int CH = NTW * 256;
  constexpr int NV4 = CH / 4;
  constexpr int PER = (NV4 + WAVES * 64 - 1) / (WAVES * 64);
  const int c = lane & 15, kq = lane >> 4;
  float4 stage[PER];
  auto load_w = [&](int ch) {
    const float4* src = (const float4*)(wsrc + (size_t)ch * Fp * 16);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      int idx = tid + i * WAVES * 64;
      stage[i] = (idx < valid4) ? src[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_w = [&](int buf) {
    float4* dst = (float4*)(smem + buf * CH);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      int idx = tid + i * WAVES * 64;
      if (idx < NV4) dst[idx] = stage[i];
    }
  };
  // A row of this lane (clamped so tail rows read valid memory; their results are not stored)
  int r = min(m0 + c, r_end - 1);
  int mr = vidx ? vidx[r] : r;
  int n = mr / T, t = mr - n * T;
  const float* arow = E + ((size_t)n * Tp + t) * D + kq * 4;

  // the accumulators start from the bias (EPI 0), so the epilogue has no loads: a load between its stores makes the
  // compiler drain every outstanding store first (loads and stores share vmcnt on gfx9)
  f32x4 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    float bv = 0.f;
    if (EPI == 0) {
      const int col = (tile0 + first + i) * 16 + c;
      bv = bias[(first + i < NTW && col < F) ? col : 0];
    }
    acc[i] = (f32x4){bv, bv, bv, bv};
  }

  __syncthreads();  // previous row block's LDS reads are done
  load_w(0);
  store_w(0);
  float4 a_cur = *(const float4*)(arow);
  __syncthreads();

  for (int ch = 0; ch < nchunk; ++ch) {
    const int buf = ch & 1;
    float4 a_nxt = a_cur;
    if (ch + 1 < nchunk) {
      load_w(ch + 1);
      a_nxt = *(const float4*)(arow + (ch + 1) * 16);
    }
    if (active) {
      const float* bs = smem + buf * CH + (first * 16 + c) * 16 + pack_b_slot(c, kq) * 4;   // conflict-free order
      // groups of G tiles: consecutive MFMAs hit different accumulators (dependent-issue latency 40
      // cycles against a 32-cycle issue interval would otherwise idle the matrix pipe)
      constexpr int G = (NT % 5 == 0) ? 5 : (NT % 4 == 0 ? 4 : (NT == 13 ? 13 : (NT == 7 ? 7 : 1)));
#pragma unroll
      for (int g0 = 0; g0 < NT; g0 += G) {
        float4 b[G];
#pragma unroll
        for (int j = 0; j < G; ++j) b[j] = *(const float4*)(bs + (g0 + j) * 256);
#pragma unroll
        for (int j = 0; j < G; ++j) acc[g0 + j] = mfma16(a_cur.x, b[j].x, acc[g0 + j]);
#pragma unroll
        for (int j = 0; j < G; ++j) acc[g0 + j] = mfma16(a_cur.y, b[j].y, acc[g0 + j]);
#pragma unroll
        for (int j = 0; j < G; ++j) acc[g0 + j] = mfma16(a_cur.z, b[j].z, acc[g0 + j]);
#pragma unroll
        for (int j = 0; j < G; ++j) acc[g0 + j] = mfma16(a_cur.w, b[j].w, acc[g0 + j]);
      }
    }
    if (ch + 1 < nchunk) store_w(buf ^ 1);
    a_cur = a_nxt;
    __syncthreads();
  }

  // epilogue: (+ bias: already in the accumulators), ReLU, * (id != 0)   (paper.py:146, 150-155)
  if (active) {
    const int rq = lane >> 4;
    float keep[4];
    int rows[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int rr = m0 + rq * 4 + j;
      bool ok = rr < r_end;
      int mm = vidx ? vidx[ok ? rr : r_end - 1] : (ok ? rr : r_end - 1);   // unconditional loads, predicate by select
      rows[j] = ok ? mm : -1;
      keep[j] = (ok && ids[mm] != 0) ? 1.f : 0.f;
    }
    if (EPI == 1) {
      int idr[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) idr[j] = rows[j] >= 0 ? ids[rows[j]] : 0;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int col = (tile0 + first + nt) * 16 + c;
        if (first + nt < NTW && col < F) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float v = acc[nt][j];
            if (rows[j] < 0 || v == 0.f) continue;   // positions without a non-pad neighbour contribute exactly 0
            if (ed.thresh) {
              if (!keep_elem(ed.key, (uint32_t)rows[j] * (uint32_t)F + (uint32_t)col, ed.thresh)) continue;
              v *= ed.scale;
            }
            atomicAdd(C + (size_t)idr[j] * F + col, v);
          }
        }
      }
      return;
    }
    // branch-free stores through a buffer descriptor: rows of the tail and columns past F get the out-of-range offset
    // (a guarded store is a branch + s_waitcnt each: 100 serialised round trips per wave and row block)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col = (tile0 + first + nt) * 16 + c;
      const bool cok = first + nt < NTW && col < F;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        buf_store1(fmaxf(acc[nt][j], 0.f) * keep[j], rsC,
                   (cok && rows[j] >= 0) ? ((uint32_t)rows[j] * (uint32_t)F + (uint32_t)col) * 4u : kBufOob, 0);
    }
  }
}

template <int NTW, int WAVES, int EPI = 0>
__global__ __launch_bounds__(WAVES * 64) void conv_fwd_kernel(const float* __restrict__ E,
                                                              const float* __restrict__ Wp,
                                                              const float* __restrict__ bias,
                                                              const int32_t* __restrict__ ids,
                                                              float* __restrict__ C, int M, int T, int Tp, int D,
                                                              int F, int Fp, int nchunk,
                                                              const int32_t* __restrict__ vidx,
                                                              const int32_t* __restrict__ vcount, EpiDrop ed) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int CH = NTW * 256;  // floats per packed chunk of this column group
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile0 = blockIdx.y * NTW;  // first column tile of this group

  // persistent row range of this workgroup over the (compacted) output rows
  const int count = vcount ? vcount[0] : M;
  const int per = ((count + (int)gridDim.x - 1) / (int)gridDim.x + 15) & ~15;
  const int r_begin = blockIdx.x * per;
  const int r_end = min(count, r_begin + per);
  const float* wsrc = Wp + (size_t)tile0 * 256;  // + ch * Fp * 16
  const int valid4 = min(CH / 4, (Fp / 16 - tile0) * 64);  // last column group may be partial
  const __amdgpu_buffer_rsrc_t rsC = make_rsrc(C, EPI == 0 ? (uint32_t)((size_t)M * F * 4) : 0u);   // EPI 0: out [M, F]

  int base = r_begin;
  // full rounds: one 16-row tile per wave, all NTW column tiles
  for (; base < r_end && (NTW != 25 || WAVES != 8 || r_end - base > 64); base += WAVES * 16) {
    const int m0 = base + wv * 16;
    conv_rows_pass<NTW, WAVES, NTW, EPI>(E, wsrc, bias, ids, C, vidx, smem, T, Tp, D, F, Fp, nchunk, valid4, tile0, m0,
                                         r_end, m0 < r_end, 0, tid, lane, ed, rsC);
  }
  // tail of <= 4 row tiles (25-tile / 8-wave configuration): split the column tiles of each row tile over the
  // idle waves so the last round costs a fraction of a full one instead of a whole wave-round
  if constexpr (NTW == 25 && WAVES == 8) {
    if (base < r_end) {
      const int rt = (r_end - base + 15) / 16;   // 1..4 row tiles left (block-uniform)
      if (rt == 1) {
        const int first = wv * 4;                // 8 waves x 4 tiles
        conv_rows_pass<NTW, WAVES, 4, EPI>(E, wsrc, bias, ids, C, vidx, smem, T, Tp, D, F, Fp, nchunk, valid4, tile0, base,
                                           r_end, first < NTW, first, tid, lane, ed, rsC);
      } else if (rt == 2) {
        const int tile = wv >> 2, first = (wv & 3) * 7;   // 4 waves x 7 tiles per row tile
        conv_rows_pass<NTW, WAVES, 7, EPI>(E, wsrc, bias, ids, C, vidx, smem, T, Tp, D, F, Fp, nchunk, valid4, tile0,
                                           base + tile * 16, r_end, first < NTW, first, tid, lane, ed, rsC);
      } else {
        const int tile = wv >> 1, first = (wv & 1) * 13;  // 2 waves x 13 tiles per row tile
        conv_rows_pass<NTW, WAVES, 13, EPI>(E, wsrc, bias, ids, C, vidx, smem, T, Tp, D, F, Fp, nchunk, valid4, tile0,
                                            base + tile * 16, r_end, tile < rt && first < NTW, first, tid, lane, ed, rsC);
      }
    }
  }
}

void launch_conv_fwd(const float* emb, const float* w_packed, const float* bias, const int32_t* ids, float* out,
                     int n_titles, int T, int D, int ksize, int F, const int32_t* vidx, const int32_t* vcount,
                     hipStream_t s) {
  int M = n_titles * T;
  if (M == 0) return;
  int Tp = T + ksize - 1, KD = ksize * D, Fp = ceil16(F), nchunk = cdiv(KD, 16), ntiles = Fp / 16;
  if (ntiles == 25) {
    constexpr int NTW = 25, WAVES = 8;
    int grid = std::min(cdiv(M, WAVES * 16), 256);  // one 8-wave workgroup per CU, persistent over its rows
    hipLaunchKernelGGL((conv_fwd_kernel<NTW, WAVES>), dim3(grid, 1), dim3(WAVES * 64),
                       2 * NTW * 256 * sizeof(float), s, emb, w_packed, bias, ids, out, M, T, Tp, D, F, Fp, nchunk,
                       vidx, vcount, EpiDrop{0u, 0u, 1.f});
  } else {
    constexpr int NTW = 4, WAVES = 4;
    int gy = cdiv(ntiles, NTW);
    int grid = std::min(cdiv(M, WAVES * 16), std::max(1, 1024 / gy));
    hipLaunchKernelGGL((conv_fwd_kernel<NTW, WAVES>), dim3(grid, gy), dim3(WAVES * 64),
                       2 * NTW * 256 * sizeof(float), s, emb, w_packed, bias, ids, out, M, T, Tp, D, F, Fp, nchunk,
                       vidx, vcount, EpiDrop{0u, 0u, 1.f});
  }
}

// ---------------------------------------------------------------------------------------------
// Input gradient of the Conv1D for a trainable word table, on the forward kernel:
//   dE[n,q,:] = sum_j dC[n, q - j + pl, :] . W[j]^T   =   window(dCpad)[n,q] . Wd ,   Wd[s*F + f][d] = W[k-1-s][d][f]
// dCpad [n][T + k - 1][F] = the conv-output gradient with zero rows for pad tokens and title borders.
// ---------------------------------------------------------------------------------------------
__global__ void pad_dconv_kernel(const float4* __restrict__ dc, const int32_t* __restrict__ ids, float4* __restrict__ out,
                                 uint32_t total4, int T, int Tp, int F4, int pad_l) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += gridDim.x * blockDim.x) {
    const uint32_t row = i / (uint32_t)F4, c4 = i - row * (uint32_t)F4;
    const uint32_t n = row / (uint32_t)Tp;
    const int t = (int)(row - n * (uint32_t)Tp) - pad_l;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t >= 0 && t < T) {
      const uint32_t tok = n * (uint32_t)T + (uint32_t)t;
      if (ids[tok] != 0) v = dc[(size_t)tok * F4 + c4];
    }
    out[i] = v;
  }
}
// packed operand image (pack_b layout) of Wd [k*F, D] straight from conv_w [k, D, F]
__global__ void pack_dgrad_w_kernel(const float* __restrict__ W, int k, int D, int F, int Np, float* __restrict__ P,
                                    size_t total) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int n = (int)(i % Np);
  const size_t r = i / Np;
  const int kk = (int)(r % 16), ch = (int)(r / 16), kr = ch * 16 + kk;   // kr = s*F + f
  float v = 0.f;
  if (kr < k * F && n < D) {
    const int sgm = kr / F, f = kr - sgm * F;
    v = W[((size_t)(k - 1 - sgm) * D + n) * F + f];
  }
  P[((size_t)ch * Np + n) * 16 + pack_b_slot(n, kk >> 2) * 4 + (kk & 3)] = v;
}
size_t conv_dgrad_scratch_floats(int n_titles, int T, int D, int ksize, int F) {
  return (size_t)n_titles * (T + ksize - 1) * F + pack_b_floats(ksize * F, D);
}
void launch_conv_dgrad_scatter(const float* dconv, const float* conv_w, const int32_t* ids, float* word_grad,
                               float* scratch, int n_titles, int T, int D, int ksize, int F, float drop, uint32_t key,
                               hipStream_t s) {
  const int M = n_titles * T;
  if (M == 0) return;
  const int Tp = T + ksize - 1, pl = (ksize - 1) / 2, pr = ksize - 1 - pl;
  float* dcp = scratch;
  float* wp = scratch + (size_t)n_titles * Tp * F;
  const size_t total4 = (size_t)n_titles * Tp * (F / 4);
  hipLaunchKernelGGL(pad_dconv_kernel, dim3((unsigned)std::min<size_t>(cdivz(total4, 256), 256 * 16)), dim3(256), 0, s,
                     (const float4*)dconv, ids, (float4*)dcp, (uint32_t)total4, T, Tp, F / 4, pr);
  const int Np = ceil16(D);
  const size_t wtotal = pack_b_floats(ksize * F, D);
  hipLaunchKernelGGL(pack_dgrad_w_kernel, dim3((unsigned)cdivz(wtotal, 256)), dim3(256), 0, s, conv_w, ksize, D, F, Np, wp,
                     wtotal);
  // the forward kernel with the roles of D and F exchanged: rows = all M token positions, K = k*F, N = D
  const int KF = ksize * F, nchunk = cdiv(KF, 16), ntiles = Np / 16;
  const EpiDrop ed{drop > 0.f ? dropout_thresh(drop) : 0u, key, drop > 0.f ? 1.0f / (1.0f - drop) : 1.0f};
  if (ntiles == 19) {
    constexpr int NTW = 19, WAVES = 8;
    const int grid = std::min(cdiv(M, WAVES * 16), 256);
    hipLaunchKernelGGL((conv_fwd_kernel<NTW, WAVES, 1>), dim3(grid, 1), dim3(WAVES * 64), 2 * NTW * 256 * sizeof(float), s,
                       dcp, wp, nullptr, ids, word_grad, M, T, Tp, F, D, Np, nchunk, nullptr, nullptr, ed);
  } else {
    constexpr int NTW = 4, WAVES = 4;
    const int gy = cdiv(ntiles, NTW);
    const int grid = std::min(cdiv(M, WAVES * 16), std::max(1, 1024 / gy));
    hipLaunchKernelGGL((conv_fwd_kernel<NTW, WAVES, 1>), dim3(grid, gy), dim3(WAVES * 64), 2 * NTW * 256 * sizeof(float), s,
                       dcp, wp, nullptr, ids, word_grad, M, T, Tp, F, D, Np, nchunk, nullptr, nullptr, ed);
  }
}

// ---------------------------------------------------------------------------------------------
// Kernel gradient  dW[kk, f] = sum_tokens im2col[tok, kk] * dC[tok, f]   (kk in [0, k*D)).
// Workgroup = WK x WN waves; wave tile = 64 kk x (CT*16) f.  The contraction index is the token, and
// both operands are token-major with the OUTPUT index contiguous, so LDS tiles are straight copies:
//   As[TK][64*WK] <- E_flat window rows,   Bs[TK][16*CT*WN] <- dC rows.
// A lane (i, kq) reads ONE float4 As[4s+kq][64wk + 4i ..+3]; element e feeds the MFMA whose output
// rows are kk = 64wk + 4i' + e  (an output-row permutation, undone in the store).  B lanes read b32.
// grid = (kk blocks, splits); split partial slabs are reduced by colsum-style kernel below.
// ---------------------------------------------------------------------------------------------
template <int WK, int WN, int CT, int TK>
__global__ __launch_bounds__(WK * WN * 64) void conv_wgrad_kernel(const float* __restrict__ E,
                                                                  const float* __restrict__ dC,
                                                                  float* __restrict__ part, int M, int T, int Tp,
                                                                  int D, int KD, int F,
                                                                  const int32_t* __restrict__ vidx,
                                                                  const int32_t* __restrict__ vcount) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NTH = WK * WN * 64;
  constexpr int AW = 64 * WK;        // A tile width (kk)
  constexpr int BW = 16 * CT * WN;   // B tile width (f)
  constexpr int STAGE = TK * (AW + BW);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wk = wv / WN, wn = wv - wk * WN;
  const int i16 = lane & 15, kq = lane >> 4;
  const int kk0 = blockIdx.x * AW;
  const int f0 = blockIdx.z * BW;
  // contraction range of this split over the (compacted) token list
  const int count = vcount ? vcount[0] : M;
  const int tok_per_split = (((count + (int)gridDim.y - 1) / (int)gridDim.y) + TK - 1) / TK * TK;
  const int tok_begin = blockIdx.y * tok_per_split;
  const int tok_end = min(count, tok_begin + tok_per_split);

  const bool fvec = (F & 3) == 0;
  constexpr int A4 = TK * AW / 4, B4 = TK * BW / 4;
  constexpr int PA = (A4 + NTH - 1) / NTH, PB = (B4 + NTH - 1) / NTH;
  float4 sa[PA], sb[PB];

  auto load_tiles = [&](int tok0) {
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      int idx = tid + p * NTH;
      if (idx < A4) {
        int r = idx / (AW / 4), x4 = idx - r * (AW / 4);
        int tk = min(tok0 + r, count - 1);  // tail rows: valid address; their dC row is zero
        int m = vidx ? vidx[max(tk, 0)] : max(tk, 0);
        int n = m / T, t = m - n * T;
        sa[p] = *(const float4*)(E + ((size_t)n * Tp + t) * D + kk0 + x4 * 4);
      }
    }
#pragma unroll
    for (int p = 0; p < PB; ++p) {
      int idx = tid + p * NTH;
      if (idx < B4) {
        int r = idx / (BW / 4), y4 = idx - r * (BW / 4);
        int tk = tok0 + r, f = f0 + y4 * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tk < tok_end) {
          int m = vidx ? vidx[tk] : tk;
          const float* src = dC + (size_t)m * F + f;
          if (fvec && f + 3 < F) {
            v = *(const float4*)src;
          } else {
            if (f < F) v.x = src[0];
            if (f + 1 < F) v.y = src[1];
            if (f + 2 < F) v.z = src[2];
            if (f + 3 < F) v.w = src[3];
          }
        }
        sb[p] = v;
      }
    }
  };
  auto store_tiles = [&](int buf) {
    float4* da = (float4*)(smem + buf * STAGE);
    float4* db = (float4*)(smem + buf * STAGE + TK * AW);
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      int idx = tid + p * NTH;
      if (idx < A4) da[idx] = sa[p];
    }
#pragma unroll
    for (int p = 0; p < PB; ++p) {
      int idx = tid + p * NTH;
      if (idx < B4) db[idx] = sb[p];
    }
  };

  f32x4 acc[4][CT];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[e][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nstage = (tok_end > tok_begin) ? cdiv(tok_end - tok_begin, TK) : 0;
  if (nstage > 0) {
    load_tiles(tok_begin);
    store_tiles(0);
  }
  __syncthreads();
  for (int st = 0; st < nstage; ++st) {
    const int buf = st & 1;
    if (st + 1 < nstage) load_tiles(tok_begin + (st + 1) * TK);
    const float* as = smem + buf * STAGE + wk * 64 + i16 * 4;
    const float* bs = smem + buf * STAGE + TK * AW + wn * CT * 16 + i16;
#pragma unroll
    for (int s4 = 0; s4 < TK / 4; ++s4) {
      float4 a = *(const float4*)(as + (s4 * 4 + kq) * AW);
      float b[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) b[ct] = bs[(s4 * 4 + kq) * BW + ct * 16];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        acc[0][ct] = mfma16(a.x, b[ct], acc[0][ct]);
        acc[1][ct] = mfma16(a.y, b[ct], acc[1][ct]);
        acc[2][ct] = mfma16(a.z, b[ct], acc[2][ct]);
        acc[3][ct] = mfma16(a.w, b[ct], acc[3][ct]);
      }
    }
    if (st + 1 < nstage) store_tiles(buf ^ 1);
    __syncthreads();
  }

  // partial slab [split][KD][F]; MFMA row (rq*4+j) of accumulator e is kk = kk0 + 64wk + 4(rq*4+j) + e
  float* dst = part + (size_t)blockIdx.y * KD * F;
  const int rq = lane >> 4;
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      int f = f0 + (wn * CT + ct) * 16 + i16;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int kk = kk0 + wk * 64 + 4 * (rq * 4 + j) + e;
        if (kk < KD && f < F) dst[(size_t)kk * F + f] = acc[e][ct][j];
      }
    }
}

// Same contraction with LDS-DMA staging (global_load_lds, 16 B per lane): the two operand tiles of the next
// stage go straight from L2 into LDS -- no staging VGPRs (the register version spilled them at the 128-VGPR
// limit of a 15-wave workgroup), no ds_write pass.  One wave instruction fills 64 consecutive float4 of the
// stage image [A 16x192 | B 16x400]; the source address is per lane (row gather through the precomputed
// per-token offsets), the instructions of a stage are dealt round-robin over the waves.  hipcc makes a wave
// wait vmcnt(0) before the first ds_read that follows its own LDS-DMA, so a wave's DMA does not overlap its own
// MFMAs -- the other 14 waves' MFMAs cover it (the row offsets of a stage are fetched one stage ahead, so the wave
// spends that wait on the transfer alone).  (A dedicated loader wave, which keeps every compute wave free of
// vmcnt waits, measured 2.5x slower: one wave cannot issue 37 gathers per stage fast enough.)
// F == 16*CT*WN (one column block).
// LDS-DMA issued through inline asm: invisible to hipcc's wait-count pass, which otherwise makes a wave wait
// vmcnt(0) before the first ds_read that follows its own DMA (so a wave's transfer never overlapped its own MFMAs).
// The wave waits for its transfers explicitly (wait_dma) before the barrier that publishes the stage.
// (M0 carries the LDS base of the transfer, exactly as in the code hipcc emits for the builtin; the kernel has no
//  other user of M0, hipcc only warns that it does not preserve a reserved register across the statement.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void glds16_asm(const float* src, uint32_t lds_byte_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_byte_base) : "m0");
}
#pragma clang diagnostic pop
__device__ __forceinline__ void wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int WK, int WN, int CT, int TK, bool ASMDMA>
__global__ __launch_bounds__(WK * WN * 64) void conv_wgrad_glds_kernel(
    const float* __restrict__ E, const float* __restrict__ dC, float* __restrict__ part, int KD, int F,
    const int32_t* __restrict__ offA, const int32_t* __restrict__ offB, const int32_t* __restrict__ vcount,
    const float* __restrict__ zeros) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NW = WK * WN;
  constexpr int AW = 64 * WK, BW = 16 * CT * WN;
  constexpr int STAGE = TK * (AW + BW);
  constexpr int A4 = TK * AW / 4, S4 = STAGE / 4;
  static_assert(A4 % 64 == 0 && S4 % 64 == 0, "stage image must be whole wave-instructions");
  constexpr int NINST = S4 / 64, PER = (NINST + NW - 1) / NW;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = wv / WN, wn = wv - wk * WN;
  const int i16 = lane & 15, kq = lane >> 4;
  const int kk0 = blockIdx.x * AW;
  const int count = vcount[0];
  const int tok_per_split = (((count + (int)gridDim.y - 1) / (int)gridDim.y) + TK - 1) / TK * TK;
  const int tok_begin = blockIdx.y * tok_per_split;
  const int tok_end = min(count, tok_begin + tok_per_split);
  const int nstage = (tok_end > tok_begin) ? cdiv(tok_end - tok_begin, TK) : 0;

  // source row of every transfer this wave issues for a stage (index loads batched, then the transfers)
  auto stage_sources = [&](int tok0, const float** src) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int q = wv + i * NW;        // wave-uniform
      src[i] = zeros;
      if (q < NINST) {
        const int idx = q * 64 + lane;
        if (q < A4 / 64) {                // A part: E rows (kk window of the token)
          const int r = idx / (AW / 4), x4 = idx - r * (AW / 4);
          const int tk = min(tok0 + r, count - 1);
          src[i] = E + (size_t)offA[tk] + kk0 + x4 * 4;
        } else {                          // B part: dC rows; rows past the split's range read zeros
          const int ib = idx - A4;
          const int r = ib / (BW / 4), y4 = ib - r * (BW / 4);
          const int tk = tok0 + r;
          if (tk < tok_end) src[i] = dC + (size_t)offB[tk] + y4 * 4;
        }
      }
    }
  };
  auto issue_stage = [&](const float* const* src, int buf) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int q = wv + i * NW;
      if (q < NINST) {
        if (ASMDMA)
          glds16_asm(src[i], (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)(smem + buf * STAGE + q * 256));
        else
          __builtin_amdgcn_global_load_lds(src[i], (__attribute__((address_space(3))) void*)(smem + buf * STAGE + q * 256),
                                           16, 0, 0);
      }
    }
  };

  f32x4 acc[4][CT];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[e][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float* src[PER];
  if (nstage > 0) {
    stage_sources(tok_begin, src);
    issue_stage(src, 0);
  }
  if (nstage > 1) stage_sources(tok_begin + TK, src);   // the row offsets of a stage are fetched one stage ahead
  if (ASMDMA) wait_dma();
  __syncthreads();
  for (int st = 0; st < nstage; ++st) {
    const int buf = st & 1;
    if (st + 1 < nstage) issue_stage(src, buf ^ 1);
    if (st + 2 < nstage) stage_sources(tok_begin + (st + 2) * TK, src);
    const float* as = smem + buf * STAGE + wk * 64 + i16 * 4;
    const float* bs = smem + buf * STAGE + TK * AW + wn * CT * 16 + i16;
#pragma unroll
    for (int s4 = 0; s4 < TK / 4; ++s4) {
      float4 a = *(const float4*)(as + (s4 * 4 + kq) * AW);
      float b[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) b[ct] = bs[(s4 * 4 + kq) * BW + ct * 16];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        acc[0][ct] = mfma16(a.x, b[ct], acc[0][ct]);
        acc[1][ct] = mfma16(a.y, b[ct], acc[1][ct]);
        acc[2][ct] = mfma16(a.z, b[ct], acc[2][ct]);
        acc[3][ct] = mfma16(a.w, b[ct], acc[3][ct]);
      }
    }
    if (ASMDMA) wait_dma();
    __syncthreads();   // vmcnt(0): the next stage has landed; every wave is done with this one
  }

  float* dst = part + (size_t)blockIdx.y * KD * F;
  const int rq = lane >> 4;
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      int f = (wn * CT + ct) * 16 + i16;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int kk = kk0 + wk * 64 + 4 * (rq * 4 + j) + e;
        if (kk < KD && f < F) dst[(size_t)kk * F + f] = acc[e][ct][j];
      }
    }
}

// per-token source offsets of the compacted list (in floats): offA = row of the padded embedding layout,
// offB = row of the conv-output-shaped tensors; entries past the count repeat the last valid token
__global__ void token_offsets_kernel(const int32_t* __restrict__ vidx, const int32_t* __restrict__ vcount, int M, int T,
                                     int Tp, int D, int F, int32_t* __restrict__ offA, int32_t* __restrict__ offB) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const int count = vcount ? vcount[0] : M;
  const int m = vidx ? vidx[min(i, max(count - 1, 0))] : i;          // no list: every token
  const int n = m / T, t = m - n * T;
  offA[i] = (n * Tp + t) * D;                                          // D: units per padded row (floats, or E3 pieces)
  offB[i] = m * F;
}
void launch_token_offsets(const int32_t* vidx, const int32_t* vcount, int M, int T, int Tp, int D, int F, int32_t* offA,
                          int32_t* offB, hipStream_t s) {
  if (M > 0) hipLaunchKernelGGL(token_offsets_kernel, dim3(cdiv(M, 256)), dim3(256), 0, s, vidx, vcount, M, T, Tp, D, F, offA, offB);
}

__global__ void reduce_slabs_kernel(const float* __restrict__ part, int nsplit, size_t count,
                                    float* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  float acc = 0.f;
  for (int s = 0; s < nsplit; ++s) acc += part[(size_t)s * count + i];
  out[i] = acc;
}

static int wgrad_splits(int kblocks, int fblocks) { return std::max(1, 256 / (kblocks * fblocks)); }

struct WgradGeom {
  int aw, bw, kblocks, fblocks, nsplit;
  bool big;
};
static WgradGeom wgrad_geom(int D, int ksize, int F) {
  WgradGeom g;
  int KD = ksize * D;
  g.big = (F == 400);
  g.aw = g.big ? 192 : 128;
  g.bw = g.big ? 400 : 64;
  g.kblocks = cdiv(KD, g.aw);
  g.fblocks = cdiv(F, g.bw);
  g.nsplit = wgrad_splits(g.kblocks, g.fblocks);
  return g;
}

size_t conv_wgrad_scratch_bytes(int D, int ksize, int F) {
  WgradGeom g = wgrad_geom(D, ksize, F);
  return (size_t)g.nsplit * ksize * D * F * sizeof(float);
}

void launch_conv_wgrad(const float* emb, const float* dout, float* dw, void* scratch, int n_titles, int T, int D,
                       int ksize, int F, const int32_t* vidx, const int32_t* vcount, hipStream_t s, const int32_t* offA,
                       const int32_t* offB, const float* zeros, ReduceJob* defer) {
  int M = n_titles * T, Tp = T + ksize - 1, KD = ksize * D;
  WgradGeom g = wgrad_geom(D, ksize, F);
  constexpr int TK = 16;
  int nsplit = std::min(g.nsplit, std::max(1, cdiv(M, TK)));   // splits beyond the token list write zero slabs
  float* part = (float*)scratch;
  if (g.big && offA && offB && zeros && vcount) {
    constexpr int WK = 3, WN = 5, CT = 5;
    size_t lds = 2 * TK * (64 * WK + 16 * CT * WN) * sizeof(float);
    static bool attr_set2 = false;
    if (!attr_set2) {
      (void)hipFuncSetAttribute((const void*)conv_wgrad_glds_kernel<WK, WN, CT, TK, false>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_set2 = true;
    }
    hipLaunchKernelGGL((conv_wgrad_glds_kernel<WK, WN, CT, TK, false>), dim3(g.kblocks, nsplit, 1),
                         dim3(WK * WN * 64), lds, s, emb, dout, part, KD, F, offA, offB, vcount, zeros);
  } else if (g.big) {
    constexpr int WK = 3, WN = 5, CT = 5;
    size_t lds = 2 * TK * (64 * WK + 16 * CT * WN) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<WK, WN, CT, TK>, hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds);
      attr_set = true;
    }
    hipLaunchKernelGGL((conv_wgrad_kernel<WK, WN, CT, TK>), dim3(g.kblocks, nsplit, g.fblocks), dim3(WK * WN * 64),
                       lds, s, emb, dout, part, M, T, Tp, D, KD, F, vidx, vcount);
  } else {
    constexpr int WK = 2, WN = 2, CT = 2;
    size_t lds = 2 * TK * (64 * WK + 16 * CT * WN) * sizeof(float);
    hipLaunchKernelGGL((conv_wgrad_kernel<WK, WN, CT, TK>), dim3(g.kblocks, nsplit, g.fblocks), dim3(WK * WN * 64),
                       lds, s, emb, dout, part, M, T, Tp, D, KD, F, vidx, vcount);
  }
  size_t count = (size_t)KD * F;
  if (defer) {
    *defer = ReduceJob{part, dw, 1, (int)count, (int)count, nsplit, count, 1.0f};
    return;
  }
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)cdivz(count, 256)), dim3(256), 0, s, part, nsplit, count, dw);
}

}  // namespace lstur
