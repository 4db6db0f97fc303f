// HBM-bound and glue kernels of the LSTUR path (gfx950): embedding gather (+dropout), operand
// packing, reductions, Keras-Adam (flat bucket and dense user-table sweep), scorer/softmax/CE.
#include "bf16_mfma.h"
#include "lstur_internal.h"

namespace lstur {

// ---------------------------------------------------------------------------------------------
// Word-embedding gather -> zero-padded per-title layout [N_t][pad_l + T + pad_r][D]
//   (keras.layers.Embedding + Dropout, reference task/paper.py:132-147; pad id 0 is gathered too)
// One 16-byte access per lane on both sides: a table row (D*4 B, 1200 B at D=300) is read by
// D/4 consecutive lanes, so every wave instruction covers whole 64-B segments of a row.
// ---------------------------------------------------------------------------------------------
// Four independent float4 per thread and pass, branch-free (pad rows are a select on an unconditional, clamped
// load): the four id loads, then the four row loads are in flight together, and the dropout hash of one piece
// overlaps the loads of the others -- a guarded load per grid-stride step left ONE dependent id -> row chain in flight
// per thread.
// BF16: the same gather with the result rounded to bf16 (round-to-nearest-even, bf16_mfma.h: cvt4 -- what the mhsa
// projections' loaders did with the fp32 rows on their way into LDS): half the bytes written here and read there
template <bool DROP, bool BF16 = false>
__global__ __launch_bounds__(256) void gather_kernel(const float4* __restrict__ table,
                                                     const int32_t* __restrict__ ids,
                                                     float4* __restrict__ out, uint32_t total4, int T, int Tp,
                                                     int D4, int pad_l, uint32_t thresh, uint32_t key,
                                                     float scale) {
  constexpr int U = 4;
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < total4; i0 += U * stride) {
    uint32_t c4[U], tok[U];
    bool inr[U], real[U];
    int id[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t i = min(i0 + u * stride, total4 - 1u);
      inr[u] = i0 + u * stride < total4;
      const uint32_t row = i / (uint32_t)D4;
      c4[u] = i - row * (uint32_t)D4;
      const uint32_t n = row / (uint32_t)Tp;
      const int t = (int)(row - n * (uint32_t)Tp) - pad_l;
      real[u] = t >= 0 && t < T;
      tok[u] = n * (uint32_t)T + (uint32_t)min(max(t, 0), T - 1);
      id[u] = ids[tok[u]];
    }
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = table[(size_t)id[u] * D4 + c4[u]];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float4 w = v[u];
      if (DROP) {
        const uint32_t e = (tok[u] * (uint32_t)D4 + c4[u]) * 4u;
        bool k[4];
        keep_quad(key, e, thresh, k);
        w.x = k[0] ? w.x * scale : 0.f;
        w.y = k[1] ? w.y * scale : 0.f;
        w.z = k[2] ? w.z * scale : 0.f;
        w.w = k[3] ? w.w * scale : 0.f;
      }
      if (!real[u]) w = make_float4(0.f, 0.f, 0.f, 0.f);
      if (inr[u]) {
        if (BF16) ((u16x4*)out)[i0 + u * stride] = cvt4(w);
        else out[i0 + u * stride] = w;
      }
    }
  }
}

void launch_gather_bf16(const float* table, const int32_t* ids, unsigned short* out, int n_titles, int T, int D, int pad_l,
                        int pad_r, float drop, uint32_t key, hipStream_t s) {
  int Tp = T + pad_l + pad_r, D4 = D / 4;
  size_t total4 = (size_t)n_titles * Tp * D4;
  if (total4 == 0) return;
  int blocks = (int)std::min<size_t>(cdivz(total4, 256 * 4), 256 * 16);
  if (drop > 0.f)
    hipLaunchKernelGGL((gather_kernel<true, true>), dim3(blocks), dim3(256), 0, s, (const float4*)table, ids, (float4*)out,
                       (uint32_t)total4, T, Tp, D4, pad_l, dropout_thresh(drop), key, 1.0f / (1.0f - drop));
  else
    hipLaunchKernelGGL((gather_kernel<false, true>), dim3(blocks), dim3(256), 0, s, (const float4*)table, ids, (float4*)out,
                       (uint32_t)total4, T, Tp, D4, pad_l, 0u, 0u, 1.0f);
}
void launch_gather(const float* table, const int32_t* ids, float* out, int n_titles, int T, int D, int pad_l,
                   int pad_r, float drop, uint32_t key, hipStream_t s) {
  int Tp = T + pad_l + pad_r, D4 = D / 4;
  size_t total4 = (size_t)n_titles * Tp * D4;
  if (total4 == 0) return;
  int blocks = (int)std::min<size_t>(cdivz(total4, 256 * 4), 256 * 16);
  if (drop > 0.f)
    hipLaunchKernelGGL(gather_kernel<true>, dim3(blocks), dim3(256), 0, s, (const float4*)table, ids,
                       (float4*)out, (uint32_t)total4, T, Tp, D4, pad_l, dropout_thresh(drop), key,
                       1.0f / (1.0f - drop));
  else
    hipLaunchKernelGGL(gather_kernel<false>, dim3(blocks), dim3(256), 0, s, (const float4*)table, ids,
                       (float4*)out, (uint32_t)total4, T, Tp, D4, pad_l, 0u, 0u, 1.0f);
}

// doc ids -> token ids through the resident title table
__global__ void expand_titles_kernel(const int32_t* __restrict__ table, const int32_t* __restrict__ doc_ids,
                                     int32_t* __restrict__ out, int total, int T) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int n = i / T, t = i - n * T;
  out[i] = table[(size_t)doc_ids[n] * T + t];
}
void launch_expand_titles(const int32_t* table, const int32_t* doc_ids, int32_t* out, int n, int T,
                          hipStream_t s) {
  int total = n * T;
  if (total) hipLaunchKernelGGL(expand_titles_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, table, doc_ids, out, total, T);
}

// ---------------------------------------------------------------------------------------------
// Stable compaction of the non-pad tokens: vidx[0..count) = ascending flat token indices with id != 0.
// The conv output of a pad token is multiplied by (id != 0) (reference task/paper.py:150-155) and its
// gradient is zero, so both conv contractions run over this list only; results are unchanged.
// Three tiny launches (count / scan / write) keep the order deterministic.
// ---------------------------------------------------------------------------------------------
constexpr int kCompactBlock = 1024;  // ids per block (256 threads x 4)

__device__ __forceinline__ int block_exclusive_scan_256(int v, int* total) {
  __shared__ int wsum[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  int base = 0;
  for (int k = 0; k < wv; ++k) base += wsum[k];
  *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  return base + inc - v;
}

__global__ __launch_bounds__(256) void compact_count_kernel(const int32_t* __restrict__ ids, int n,
                                                            int32_t* __restrict__ block_counts) {
  int i0 = blockIdx.x * kCompactBlock + threadIdx.x * 4;
  int c = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) c += (i0 + k < n && ids[i0 + k] != 0) ? 1 : 0;
  int total;
  block_exclusive_scan_256(c, &total);
  if (threadIdx.x == 0) block_counts[blockIdx.x] = total;
}
__global__ __launch_bounds__(256) void compact_scan_kernel(int32_t* __restrict__ block_counts, int nblocks,
                                                           int32_t* __restrict__ count_out) {
  // exclusive scan of block_counts in place (single block, chunks of 256)
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int b0 = 0; b0 < nblocks; b0 += 256) {
    int i = b0 + threadIdx.x;
    int v = i < nblocks ? block_counts[i] : 0;
    int total;
    int ex = block_exclusive_scan_256(v, &total);
    int c0 = carry;
    if (i < nblocks) block_counts[i] = c0 + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry = c0 + total;
    __syncthreads();
  }
  if (threadIdx.x == 0) count_out[0] = carry;
}
__global__ __launch_bounds__(256) void compact_write_kernel(const int32_t* __restrict__ ids, int n,
                                                            const int32_t* __restrict__ block_offsets,
                                                            int32_t* __restrict__ vidx) {
  int i0 = blockIdx.x * kCompactBlock + threadIdx.x * 4;
  bool v[4];
  int c = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[k] = i0 + k < n && ids[i0 + k] != 0;
    c += v[k] ? 1 : 0;
  }
  int total;
  int pos = block_offsets[blockIdx.x] + block_exclusive_scan_256(c, &total);
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (v[k]) vidx[pos++] = i0 + k;
}
// ids-form staging: expands the history and candidate doc ids through the resident title table AND counts the
// non-pad tokens of every 1024-id block (what compact_count_kernel would do) in the same launch.
__global__ __launch_bounds__(256) void expand_count_kernel(const int32_t* __restrict__ table,
                                                           const int32_t* __restrict__ hist_ids, int n_hist,
                                                           const int32_t* __restrict__ cand_ids, int n_total, int T,
                                                           int32_t* __restrict__ out, int32_t* __restrict__ block_counts) {
  const int n = n_total * T;
  const int i0 = blockIdx.x * kCompactBlock + threadIdx.x * 4;
  int c = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = i0 + k;
    if (i < n) {
      const int title = i / T, t = i - title * T;
      const int doc = title < n_hist ? hist_ids[title] : cand_ids[title - n_hist];
      const int id = table[(size_t)doc * T + t];
      out[i] = id;
      c += id != 0;
    }
  }
  int total;
  block_exclusive_scan_256(c, &total);
  if (threadIdx.x == 0) block_counts[blockIdx.x] = total;
}
void launch_expand_count(const int32_t* table, const int32_t* hist_ids, int n_hist, const int32_t* cand_ids, int n_cand,
                         int T, int32_t* out, int32_t* block_counts, hipStream_t s) {
  const int n = (n_hist + n_cand) * T;
  if (n > 0)
    hipLaunchKernelGGL(expand_count_kernel, dim3(cdiv(n, kCompactBlock)), dim3(256), 0, s, table, hist_ids, n_hist,
                       cand_ids, n_hist + n_cand, T, out, block_counts);
}
// Second half of the compaction with the block prefix computed in place (<= a few hundred block counts, each block
// sums its predecessors) and the conv kernel-gradient's per-token row offsets written along: offA = row of the padded
// embedding layout, offB = row of the conv-output-shaped tensors (entries past the count are never read).
__global__ __launch_bounds__(256) void compact_write_offsets_kernel(const int32_t* __restrict__ ids, int n,
                                                                    const int32_t* __restrict__ block_counts,
                                                                    int32_t* __restrict__ vidx,
                                                                    int32_t* __restrict__ count_out, int T, int Tp, int D,
                                                                    int F, int32_t* __restrict__ offA,
                                                                    int32_t* __restrict__ offB, int nblk,
                                                                    uint32_t* __restrict__ live, int split_at) {
  if ((int)blockIdx.x >= nblk) {   // trailing blocks: bit t of live[title] = token t is not a pad (t < 32): the attention
    const int title = ((int)blockIdx.x - nblk) * 256 + threadIdx.x;   // kernels read ONE scalar per title instead of ids
    if (title * T < n) {
      uint32_t m = 0;
      for (int t = 0; t < T && t < 32; ++t) m |= (ids[title * T + t] != 0 ? 1u : 0u) << t;
      live[title] = m;
    }
    return;
  }
  __shared__ int red[4];
  int part = 0;
  for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) part += block_counts[b];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  const int base = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  const int i0 = blockIdx.x * kCompactBlock + threadIdx.x * 4;
  bool v[4];
  int c = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[k] = i0 + k < n && ids[i0 + k] != 0;
    c += v[k] ? 1 : 0;
  }
  int total;
  int pos = base + block_exclusive_scan_256(c, &total);
  // count_out[1]: the non-pad tokens in front of token `split_at` (a title boundary: history | candidates) -- the compacted
  // list is ascending, so [0, count_out[1]) and [count_out[1], count_out[0]) are the two title ranges' tokens
  if (i0 <= split_at && split_at < i0 + 4) {
    int before = pos;
#pragma unroll
    for (int k = 0; k < 4; ++k) before += (i0 + k < split_at && v[k]) ? 1 : 0;
    count_out[1] = before;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (v[k]) {
      const int m = i0 + k;
      vidx[pos] = m;
      if (offA) {
        const int title = m / T, t = m - title * T;
        offA[pos] = (title * Tp + t) * D;
        offB[pos] = m * F;
      }
      ++pos;
    }
  if ((int)blockIdx.x == nblk - 1 && threadIdx.x == 0) {
    count_out[0] = base + total;
    if (split_at >= nblk * kCompactBlock || split_at < 0) count_out[1] = base + total;   // (no thread owns that index)
  }
}
size_t compact_scratch_ints(int n) { return (size_t)cdiv(n, kCompactBlock); }
void launch_compact_write_offsets(const int32_t* ids, int n, const int32_t* block_counts, int32_t* vidx, int32_t* count,
                                  int T, int Tp, int D, int F, int32_t* offA, int32_t* offB, hipStream_t s,
                                  uint32_t* live, int split_at) {
  if (n <= 0) return;
  const int nblk = cdiv(n, kCompactBlock), lblk = live ? cdiv(n / T, 256) : 0;
  hipLaunchKernelGGL(compact_write_offsets_kernel, dim3(nblk + lblk), dim3(256), 0, s, ids, n, block_counts, vidx, count, T,
                     Tp, D, F, offA, offB, nblk, live, split_at < 0 ? n : split_at);
}
void launch_compact_count(const int32_t* ids, int n, int32_t* block_counts, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(compact_count_kernel, dim3(cdiv(n, kCompactBlock)), dim3(256), 0, s, ids, n, block_counts);
}
void launch_compact_valid(const int32_t* ids, int n, int32_t* vidx, int32_t* count, int32_t* scratch, hipStream_t s) {
  if (n <= 0) return;
  int nb = cdiv(n, kCompactBlock);
  hipLaunchKernelGGL(compact_count_kernel, dim3(nb), dim3(256), 0, s, ids, n, scratch);
  hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(256), 0, s, scratch, nb, count);
  hipLaunchKernelGGL(compact_write_kernel, dim3(nb), dim3(256), 0, s, ids, n, scratch, vidx);
}

// ---------------------------------------------------------------------------------------------
// B[K,N] -> MFMA operand image P[ceil(K/16)][ceil16(N)][16]:  P[ch][n][kk] = B[ch*16+kk][n]
// A lane (c, kq) of mfma_f32_16x16x4 then reads ONE float4 P[ch][n0+c][4kq..4kq+3] and feeds
// element s to MFMA s of the chunk (the k order inside a chunk is permuted identically on the
// A side, which reads float4s along k).
// ---------------------------------------------------------------------------------------------
// The four k-quads of a column are stored in the order slot = ((n >> 2) & 3) ^ pi(kq), pi = (0, 3, 1, 2)
// (pack_b_slot, lstur_device.h): a workgroup copies chunks of P verbatim into LDS, and with this order the 16 lanes
// of every ds_read_b128 lane group ({0-3,12-15,20-27}, ...) hit 16 different 16-byte bank slots; the natural order
// made every such read a 2-way bank conflict (41 % of the conv kernel's LDS cycles).
size_t pack_b_floats(int K, int N) { return (size_t)cdiv(K, 16) * ceil16(N) * 16; }

template <bool TRANS>
__global__ void pack_b_kernel(const float* __restrict__ B, int ldb, int K, int N, int Np, float* __restrict__ P,
                              size_t total) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int n = (int)(i % Np);
  size_t r = i / Np;
  int kk = (int)(r % 16);
  int ch = (int)(r / 16);
  int k = ch * 16 + kk;
  float v = 0.f;
  if (k < K && n < N) v = TRANS ? B[(size_t)n * ldb + k] : B[(size_t)k * ldb + n];
  P[((size_t)ch * Np + n) * 16 + pack_b_slot(n, kk >> 2) * 4 + (kk & 3)] = v;
}
void launch_pack_b(const float* B, int ldb, int K, int N, float* packed, hipStream_t s) {
  size_t total = pack_b_floats(K, N);
  if (total)
    hipLaunchKernelGGL(pack_b_kernel<false>, dim3((unsigned)cdivz(total, 256)), dim3(256), 0, s, B, ldb, K, N,
                       ceil16(N), packed, total);
}
// same image for B = Bt^T where Bt is stored [N rows, K cols] with leading dim ldb
void launch_pack_b_t(const float* Bt, int ldb, int K, int N, float* packed, hipStream_t s) {
  size_t total = pack_b_floats(K, N);
  if (total)
    hipLaunchKernelGGL(pack_b_kernel<true>, dim3((unsigned)cdivz(total, 256)), dim3(256), 0, s, Bt, ldb, K, N,
                       ceil16(N), packed, total);
}

__global__ void transpose_kernel(const float* __restrict__ in, int ld_in, int rows, int cols,
                                 float* __restrict__ out) {
  __shared__ float tile[32][33];
  int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    int r = r0 + j, c = c0 + threadIdx.x;
    tile[j][threadIdx.x] = (r < rows && c < cols) ? in[(size_t)r * ld_in + c] : 0.f;
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    int c = c0 + j, r = r0 + threadIdx.x;
    if (c < cols && r < rows) out[(size_t)c * rows + r] = tile[threadIdx.x][j];
  }
}
void launch_transpose(const float* in, int ld_in, int rows_in, int cols_in, float* out, hipStream_t s) {
  if (rows_in == 0 || cols_in == 0) return;
  hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(cols_in, 32), cdiv(rows_in, 32)), dim3(32, 8), 0, s, in, ld_in,
                     rows_in, cols_in, out);
}

// out[j] = scale * sum_i x[i, j]   (fixed summation order -> bitwise reproducible)
// grid = (column blocks of 64, row slices); each block sums its row slice into part[slice][col].
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ x, int rows, int cols, int ld,
                                                      float* __restrict__ out, float scale, int rows_per_slice) {
  __shared__ float red[16][64];
  int c = blockIdx.x * 64 + threadIdx.x;
  int r0 = blockIdx.y * rows_per_slice, r1 = min(rows, r0 + rows_per_slice);
  float acc = 0.f;
  if (c < cols)
    for (int r = r0 + threadIdx.y; r < r1; r += 16) acc += x[(size_t)r * ld + c];
  red[threadIdx.y][threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.y == 0 && c < cols) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][threadIdx.x];
    out[(size_t)blockIdx.y * cols + c] = t * scale;
  }
}
void launch_colsum(const float* x, int rows, int cols, int ld, float* out, float scale, hipStream_t s) {
  if (cols) hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(cols, 64), 1), dim3(64, 16), 0, s, x, rows, cols, ld, out, scale, rows);
}
// two-stage variant for tall inputs: scratch holds kColsumSlices x cols partial sums
void launch_colsum2(const float* x, int rows, int cols, int ld, float* out, float scale, float* scratch,
                    hipStream_t s, ReduceJob* defer) {
  if (defer) defer->nslices = 0;
  if (!cols) return;
  if (rows < 256 || !scratch) return launch_colsum(x, rows, cols, ld, out, scale, s);
  int slices = std::min(kColsumSlices, cdiv(rows, 32));
  int rps = cdiv(rows, slices);
  slices = cdiv(rows, rps);
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(cols, 64), slices), dim3(64, 16), 0, s, x, rows, cols, ld, scratch, 1.0f, rps);
  if (defer) {
    *defer = ReduceJob{scratch, out, 1, cols, cols, slices, (size_t)cols, scale};
    return;
  }
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(cols, 64), 1), dim3(64, 16), 0, s, scratch, slices, cols, cols, out, scale, slices);
}

// all deferred reductions of a step in one launch (see ReduceJob, lstur_internal.h)
struct ReduceJobs {
  ReduceJob job[kMaxReduceJobs];
  int first_block[kMaxReduceJobs + 1];
  int njobs;
};
// a thread reduces 4 consecutive elements (one float4 per slice) where the job's shape allows, else one element
__device__ __forceinline__ bool reduce_job_vec4(const ReduceJob& q) {
  return (q.cols % 4 == 0) && (q.ld % 4 == 0) && (q.stride % 4 == 0) && ((((uintptr_t)q.src) | ((uintptr_t)q.dst)) & 15) == 0;
}
__global__ __launch_bounds__(256) void multi_reduce_kernel(ReduceJobs J) {
  int j = 0;
  while (j + 1 < J.njobs && (int)blockIdx.x >= J.first_block[j + 1]) ++j;
  const ReduceJob& q = J.job[j];
  const size_t count = (size_t)q.rows * q.cols;
  const size_t t = (size_t)(blockIdx.x - J.first_block[j]) * 256 + threadIdx.x;
  if (reduce_job_vec4(q)) {
    const size_t i = t * 4;
    if (i >= count) return;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // eight slice loads in flight, added in slice order (the order is part of the contract: bitwise reproducible)
    for (int s0 = 0; s0 < q.nslices; s0 += 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *(const float4*)(q.src + (size_t)min(s0 + u, q.nslices - 1) * q.stride + i);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (s0 + u < q.nslices) {
          acc.x += v[u].x;
          acc.y += v[u].y;
          acc.z += v[u].z;
          acc.w += v[u].w;
        }
      }
    }
    const int r = (int)(i / q.cols), c = (int)(i - (size_t)r * q.cols);
    float4* d = (float4*)(q.dst + (size_t)r * q.ld + c);
    float4 o = make_float4(q.alpha * acc.x, q.alpha * acc.y, q.alpha * acc.z, q.alpha * acc.w);
    if (q.beta != 0.f) {
      const float4 p = *d;
      o.x += q.beta * p.x;
      o.y += q.beta * p.y;
      o.z += q.beta * p.z;
      o.w += q.beta * p.w;
    }
    *d = o;
    return;
  }
  const size_t i = t;
  if (i >= count) return;
  float acc = 0.f;
  // sixteen slice loads in flight, added in slice order (one dependent load per slice made the batch means of loss /
  // accuracy -- 64 slices of one element -- the longest job of the launch: 31 us)
  for (int s0 = 0; s0 < q.nslices; s0 += 16) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = q.src[(size_t)min(s0 + u, q.nslices - 1) * q.stride + i];
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (s0 + u < q.nslices) acc += v[u];
  }
  const int r = (int)(i / q.cols), c = (int)(i - (size_t)r * q.cols);
  float* d = q.dst + (size_t)r * q.ld + c;
  *d = q.beta != 0.f ? q.alpha * acc + q.beta * *d : q.alpha * acc;
}
static bool reduce_job_vec4_host(const ReduceJob& q) {
  return (q.cols % 4 == 0) && (q.ld % 4 == 0) && (q.stride % 4 == 0) && ((((uintptr_t)q.src) | ((uintptr_t)q.dst)) & 15) == 0;
}
// false: more than kMaxReduceJobs non-empty jobs (nothing is launched: a dropped job would be a silently missing gradient)
bool launch_multi_reduce(const ReduceJob* jobs, int njobs, hipStream_t s) {
  ReduceJobs J;
  J.njobs = 0;
  int blocks = 0;
  for (int k = 0; k < njobs; ++k) {
    if (jobs[k].nslices == 0 || (size_t)jobs[k].rows * jobs[k].cols == 0) continue;   // (nslices < 0: dst = 0)
    if (J.njobs == kMaxReduceJobs) return false;
    J.first_block[J.njobs] = blocks;
    J.job[J.njobs++] = jobs[k];
    const size_t work = (size_t)jobs[k].rows * jobs[k].cols;
    blocks += (int)cdivz(reduce_job_vec4_host(jobs[k]) ? work / 4 : work, 256);
  }
  J.first_block[J.njobs] = blocks;
  if (blocks) hipLaunchKernelGGL(multi_reduce_kernel, dim3(blocks), dim3(256), 0, s, J);
  return true;
}

// ---------------------------------------------------------------------------------------------
// keras.optimizers.Adam (Keras 2.2.4; reference task/paper.py:657):
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr_t * m / (sqrt(v) + eps)
// with lr_t = lr sqrt(1-b2^t)/(1-b1^t) computed by the caller.
// ---------------------------------------------------------------------------------------------
// (no FMA contraction: every kernel that applies the update -- full sweep, zero-gradient sweep, single rows -- must round
//  alike whatever the compiler would fuse in its surroundings; separately rounded products are also what NumPy does)
__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float lr_t, float b1, float b2,
                                      float eps) {
#pragma clang fp contract(off)
  m = b1 * m + (1.0f - b1) * g;
  v = b2 * v + (1.0f - b2) * (g * g);
  p = p - lr_t * m / (sqrtf(v) + eps);
}

__global__ void adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                 float* __restrict__ v, size_t n, float lr_t, float b1, float b2, float eps) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float pi = p[i], mi = m[i], vi = v[i];
    adam1(pi, g[i], mi, vi, lr_t, b1, b2, eps);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
  }
}
// Adam over the dense bucket + the duplicate-row combine of the sparse user gradient (independent work) in one
// launch: blocks [0, flat_blocks) sweep the bucket, the rest handle one sample row per wave.
template <bool RF>
__device__ __forceinline__ void combine_row_t(const int32_t* __restrict__ rows, const float* __restrict__ rows_f,
                                              int32_t* __restrict__ rows_out, const float* __restrict__ g, int n, int dim,
                                              float* __restrict__ gsum, int32_t* __restrict__ slot, int i, int lane);
struct ImgState {          // the scales this launch writes the fp16-pair images with, the maxima it sees
  float sW, sU;
  uint32_t wmax, umax;
};
__device__ __forceinline__ void image_scatter(const ImagePlan& P, size_t i, float pi, ImgState& S);
__device__ __forceinline__ ImgState image_scale(const ImagePlan& P);
__device__ __forceinline__ void image_finish(const ImagePlan& P, const ImgState& S, bool first_block);
__global__ __launch_bounds__(256) void adam_flat_combine_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                                float* __restrict__ m, float* __restrict__ v, size_t n,
                                                                float lr_t, float b1, float b2, float eps,
                                                                int flat_blocks, const int32_t* __restrict__ rows,
                                                                const float* __restrict__ rg, int n_rows, int dim,
                                                                float* __restrict__ gsum, int32_t* __restrict__ slot,
                                                                ImagePlan P, const float* __restrict__ rows_f,
                                                                int32_t* __restrict__ rows_out) {
  if ((int)blockIdx.x >= flat_blocks) {
    const int i = (blockIdx.x - flat_blocks) * 4 + (threadIdx.x >> 6);
    if (i < n_rows) {
      if (rows_f) combine_row_t<true>(nullptr, rows_f, rows_out, rg, n_rows, dim, gsum, slot, i, threadIdx.x & 63);
      else combine_row_t<false>(rows, nullptr, nullptr, rg, n_rows, dim, gsum, slot, i, threadIdx.x & 63);
    }
    return;
  }
  ImgState S = image_scale(P);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)flat_blocks * 256) {
    float pi = p[i], mi = m[i], vi = v[i];
    adam1(pi, g[i], mi, vi, lr_t, b1, b2, eps);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
    image_scatter(P, i, pi, S);         // (an empty plan scatters nothing)
  }
  image_finish(P, S, blockIdx.x == 0);
}
void launch_adam_flat_combine(float* p, const float* g, float* m, float* v, size_t n, float lr_t, float b1, float b2,
                              float eps, const int32_t* rows, const float* rg, int n_rows, int dim, float* gsum,
                              int32_t* slot, hipStream_t s, const ImagePlan* plan, const float* rows_f, int32_t* rows_out) {
  const int fb = (int)std::min<size_t>(cdivz(n, 256), 4096);
  const int cb = cdiv(n_rows, 4);
  if (fb + cb)
    hipLaunchKernelGGL(adam_flat_combine_kernel, dim3(fb + cb), dim3(256), 0, s, p, g, m, v, n, lr_t, b1, b2, eps, fb, rows,
                       rg, n_rows, dim, gsum, slot, plan ? *plan : ImagePlan(), rows_f, rows_out);
}
// the updated element i of the bucket also goes into the weight images that hold it (ImagePlan, lstur_internal.h)
// f16x2 images: the scale comes from the maximum the previous pack saw, this launch leaves its own for the next one
// (the rule of convx.hip: convx_pack_w_kernel, restated here so that the image costs no launch of its own)
// (workgroups of 256 threads, EXEC full: every thread of the sweep's blocks calls both)
__device__ __forceinline__ uint32_t slot_max(const uint32_t* words, int first_word) {   // first_word: word of lane 0 of the slot
  uint32_t m = words[first_word + (int)(threadIdx.x & (kWmaxLanes - 1)) * kWmaxStride];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
  return m;
}
__device__ __forceinline__ ImgState image_scale(const ImagePlan& P) {
  ImgState S{1.0f, 1.0f, 0u, 0u};
  if (P.conv_cnt && P.np == 2)
    S.sW = pow2_scale(__float_as_uint(__uint_as_float(slot_max((const uint32_t*)P.cs, wmax_word(P.seq % 3, 0))) + P.grow), 13);
  if (P.H && P.gru_f16) {
    const uint32_t* tail = (const uint32_t*)(P.gimg + (size_t)P.KQ * 4 * 6 * P.H);
    S.sU = pow2_scale(__float_as_uint(__uint_as_float(slot_max(tail, gru_wmax_word(P.gru_seq % 3, 0))) + P.ggrow), 13);
  }
  return S;
}
// a workgroup leaves its maximum in word (workgroup % lanes) of slot `add` and (first workgroup) clears slot `clear`
__device__ __forceinline__ void slot_leave(uint32_t* words, uint32_t m, int add_word0, int clear_word0, bool first_block, uint32_t* wm) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
    if (m) atomicMax(words + add_word0 + (int)(blockIdx.x & (kWmaxLanes - 1)) * kWmaxStride, m);
  }
  if (first_block && threadIdx.x < kWmaxLanes) words[clear_word0 + (int)threadIdx.x * kWmaxStride] = 0u;
  __syncthreads();
}
__device__ __forceinline__ void image_finish(const ImagePlan& P, const ImgState& S, bool first_block) {
  __shared__ uint32_t wm[4];
  if (P.conv_cnt && P.np == 2) {
    slot_leave((uint32_t*)P.cs, S.wmax, wmax_word((P.seq + 1) % 3, 0), wmax_word((P.seq + 2) % 3, 0), first_block, wm);
    if (first_block && threadIdx.x == 0) {
      P.cs[kCsW] = S.sW;
      P.cs[kCsFwd] = 1.0f / (P.cs[kCsE] * S.sW);
    }
  }
  if (P.H && P.gru_f16) {
    float* tail = P.gimg + (size_t)P.KQ * 4 * 6 * P.H;
    slot_leave((uint32_t*)tail, S.umax, gru_wmax_word((P.gru_seq + 1) % 3, 0), gru_wmax_word((P.gru_seq + 2) % 3, 0), first_block, wm);
    if (first_block && threadIdx.x == 0) {
      tail[0] = S.sU;
      tail[1] = 1.0f / S.sU;
    }
  }
}
// the fp16 pair of v (scaled) at group (4 elements = 16 bytes), element j   (gru4.hip: gru4_put)
__device__ __forceinline__ void put_pair(float* img, size_t group4, int j, float v) {
  v = fminf(fmaxf(v, -65504.f), 65504.f);
  const _Float16 h = (_Float16)v;
  const _Float16 l = (_Float16)(v - (float)h);
  unsigned short* p = (unsigned short*)img + 2 * group4 + j;
  p[0] = __builtin_bit_cast(unsigned short, h);
  p[4] = __builtin_bit_cast(unsigned short, l);
}
__device__ __forceinline__ void image_scatter(const ImagePlan& P, size_t i, float pi, ImgState& S) {
    if (i - P.conv_off < P.conv_cnt) {            // conv_w[tap * D + d][f] -> Wx[part][tap * Dp + d][f]  (convx_pack_w_kernel)
      const size_t j = i - P.conv_off;
      const int f = (int)(j % P.F), td = (int)(j / P.F), tap = td / P.D, d = td - tap * P.D;
      const size_t idx = ((size_t)tap * P.Dp + d) * P.Np + f;
      if (P.np == 2) {
        u16x4 h, l;
        split2(make_float4(fminf(fmaxf(pi * S.sW, -65504.f), 65504.f), 0.f, 0.f, 0.f), h, l);
        P.w3[idx] = h[0];
        P.w3[P.w3_part + idx] = l[0];
        S.wmax = max(S.wmax, __float_as_uint(fabsf(pi)));
      } else {
        u16x4 h, mm, l;
        split3(make_float4(pi, 0.f, 0.f, 0.f), h, mm, l);
        P.w3[idx] = h[0];
        P.w3[P.w3_part + idx] = mm[0];
        P.w3[2 * P.w3_part + idx] = l[0];
      }
    }
    if (P.H && i - P.gru_off < (size_t)3 * P.H * P.H) {   // gru_u[k][c]: the forward image and a transposed one (gru4_pack_kernel)
      const size_t j = i - P.gru_off;
      const int H = P.H, c = (int)(j % (3 * H)), k = (int)(j / (3 * H));
      const size_t e = (size_t)P.KQ * 4 * H;
      const int which = 1 + c / H, kk = c - (which - 1) * H;                                             // itz | itr | ith
      const size_t g_zr = ((size_t)(k >> 2) * 2 * H + c) * 4, g_h = 2 * e + ((size_t)(k >> 2) * H + (c - 2 * H)) * 4;
      const size_t g_t = 2 * e + (size_t)which * e + ((size_t)(kk >> 2) * H + k) * 4;
      if (P.gru_f16) {      // the forward's images (izr, ih) as fp16 pairs; the BPTT's stay floats (gru4.hip)
        put_pair(P.gimg, c < 2 * H ? g_zr : g_h, k & 3, pi * S.sU);
        S.umax = max(S.umax, __float_as_uint(fabsf(pi)));
      } else {
        P.gimg[(c < 2 * H ? g_zr : g_h) + (k & 3)] = pi;
      }
      P.gimg[g_t + (kk & 3)] = pi;
    }
}
__global__ void adam_flat_pack_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                      float* __restrict__ v, size_t n, float lr_t, float b1, float b2, float eps, ImagePlan P) {
  ImgState S = image_scale(P);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float pi = p[i], mi = m[i], vi = v[i];
    adam1(pi, g[i], mi, vi, lr_t, b1, b2, eps);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
    image_scatter(P, i, pi, S);
  }
  image_finish(P, S, blockIdx.x == 0);
}
void launch_adam_flat_pack(float* p, const float* g, float* m, float* v, size_t n, float lr_t, float b1, float b2, float eps,
                           const ImagePlan& plan, hipStream_t s) {
  if (!n) return;
  const int blocks = (int)std::min<size_t>(cdivz(n, 256), 4096);
  hipLaunchKernelGGL(adam_flat_pack_kernel, dim3(blocks), dim3(256), 0, s, p, g, m, v, n, lr_t, b1, b2, eps, plan);
}
void launch_adam_flat(float* p, const float* g, float* m, float* v, size_t n, float lr_t, float b1, float b2,
                      float eps, hipStream_t s) {
  if (!n) return;
  int blocks = (int)std::min<size_t>(cdivz(n, 256), 4096);
  hipLaunchKernelGGL(adam_flat_kernel, dim3(blocks), dim3(256), 0, s, p, g, m, v, n, lr_t, b1, b2, eps);
}

// Dense sweep of the user table: TF-1 Keras densifies the embedding gradient, so every row's m and v
// decay and every row moves each step.  HBM-bound: 3 reads + 3 writes / elem.
// UNMARKED: only the rows without a slot (not in this batch) move, with their zero gradient -- the first half of the
// fused step's sweep, which needs nothing from the step's backward (launch_adam_rows finishes the batch's rows).
template <int VEC, bool UNMARKED = false>
__global__ __launch_bounds__(256) void adam_table_kernel(float* __restrict__ p, float* __restrict__ m,
                                                         float* __restrict__ v, const int32_t* __restrict__ slot,
                                                         const float* __restrict__ gsum, size_t total, int dimv,
                                                         float lr_t, float b1, float b2, float eps) {
  // a block sweeps ONE contiguous range of the table (256-element pieces, one after the other): a grid-strided sweep
  // touched 6 streams x 8192 blocks of pages at once and fell to 4.1 TB/s on the 2.4 GB of a 1 M-row table
  const size_t per = (total + gridDim.x - 1) / gridDim.x, lo = (size_t)blockIdx.x * per, hi = lo + per < total ? lo + per : total;
  for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    size_t row = i / dimv;
    int c = (int)(i - row * dimv);
    int sl = slot[row];
    if (UNMARKED && sl >= 0) continue;
    if (VEC == 4) {
      float4 pp = ((float4*)p)[i], mm = ((float4*)m)[i], vv = ((float4*)v)[i];
      float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
      if (sl >= 0) g = ((const float4*)gsum)[(size_t)sl * dimv + c];
      adam1(pp.x, g.x, mm.x, vv.x, lr_t, b1, b2, eps);
      adam1(pp.y, g.y, mm.y, vv.y, lr_t, b1, b2, eps);
      adam1(pp.z, g.z, mm.z, vv.z, lr_t, b1, b2, eps);
      adam1(pp.w, g.w, mm.w, vv.w, lr_t, b1, b2, eps);
      ((float4*)p)[i] = pp;
      ((float4*)m)[i] = mm;
      ((float4*)v)[i] = vv;
    } else {
      float pp = p[i], mm = m[i], vv = v[i];
      float g = sl >= 0 ? gsum[(size_t)sl * dimv + c] : 0.f;
      adam1(pp, g, mm, vv, lr_t, b1, b2, eps);
      p[i] = pp;
      m[i] = mm;
      v[i] = vv;
    }
  }
}
void launch_adam_table(float* p, float* m, float* v, const int32_t* slot, const float* gsum, int n_rows, int dim,
                       float lr_t, float b1, float b2, float eps, hipStream_t s) {
  if (!n_rows) return;
  if (dim % 4 == 0) {
    size_t total = (size_t)n_rows * (dim / 4);
    int blocks = (int)std::min<size_t>(cdivz(total, 256), 256 * 32);
    hipLaunchKernelGGL(adam_table_kernel<4>, dim3(blocks), dim3(256), 0, s, p, m, v, slot, gsum, total, dim / 4,
                       lr_t, b1, b2, eps);
  } else {
    size_t total = (size_t)n_rows * dim;
    int blocks = (int)std::min<size_t>(cdivz(total, 256), 256 * 32);
    hipLaunchKernelGGL(adam_table_kernel<1>, dim3(blocks), dim3(256), 0, s, p, m, v, slot, gsum, total, dim, lr_t,
                       b1, b2, eps);
  }
}

void launch_adam_table_unmarked(float* p, float* m, float* v, const int32_t* slot, int n_rows, int dim, float lr_t,
                                float b1, float b2, float eps, int max_blocks, hipStream_t s) {
  if (!n_rows) return;
  if (max_blocks <= 0) max_blocks = 256 * 32;
  if (dim % 4 == 0) {
    size_t total = (size_t)n_rows * (dim / 4);
    int blocks = (int)std::min<size_t>(cdivz(total, 256), max_blocks);
    hipLaunchKernelGGL((adam_table_kernel<4, true>), dim3(blocks), dim3(256), 0, s, p, m, v, slot, (const float*)nullptr,
                       total, dim / 4, lr_t, b1, b2, eps);
  } else {
    size_t total = (size_t)n_rows * dim;
    int blocks = (int)std::min<size_t>(cdivz(total, 256), max_blocks);
    hipLaunchKernelGGL((adam_table_kernel<1, true>), dim3(blocks), dim3(256), 0, s, p, m, v, slot, (const float*)nullptr,
                       total, dim, lr_t, b1, b2, eps);
  }
}
// the batch's rows: marked before the unmarked sweep (any non-negative value; launch_user_grad_combine then writes the
// first-occurrence index), updated from their combined gradients afterwards -- one wave per sample, first occurrences only
__global__ void mark_rows_kernel(const int32_t* __restrict__ rows, int n, int32_t* __restrict__ slot) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) slot[rows[i]] = i;
}
void launch_user_rows_mark(const int32_t* rows, int n, int32_t* slot, hipStream_t s) {
  if (n) hipLaunchKernelGGL(mark_rows_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, rows, n, slot);
}
__global__ __launch_bounds__(64) void adam_rows_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                       const int32_t* __restrict__ rows, const int32_t* __restrict__ slot,
                                                       const float* __restrict__ gsum, int dim, float lr_t, float b1,
                                                       float b2, float eps) {
  const int i = blockIdx.x, r = rows[i];
  if (slot[r] != i) return;
  for (int c = threadIdx.x; c < dim; c += 64) {
    const size_t e = (size_t)r * dim + c;
    float pp = p[e], mm = m[e], vv = v[e];
    adam1(pp, gsum[(size_t)i * dim + c], mm, vv, lr_t, b1, b2, eps);
    p[e] = pp;
    m[e] = mm;
    v[e] = vv;
  }
}
void launch_adam_rows(float* p, float* m, float* v, const int32_t* rows, const int32_t* slot, const float* gsum, int n,
                      int dim, float lr_t, float b1, float b2, float eps, hipStream_t s) {
  if (n) hipLaunchKernelGGL(adam_rows_kernel, dim3(n), dim3(64), 0, s, p, m, v, rows, slot, gsum, dim, lr_t, b1, b2, eps);
}
// Deferred zero-gradient sweep (data-parallel step, lstur_capi.hip): the rows a forward is about to read that the pending
// sweep has not reached take their zero-gradient update here and are marked, so that the sweep skips them.  One wave per
// sample; of several samples with the same row one claims it.
__global__ __launch_bounds__(64) void defer_rows_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                        float* __restrict__ p2, float* __restrict__ m2, float* __restrict__ v2,
                                                        const int32_t* __restrict__ rows, int32_t* __restrict__ slot, int dim,
                                                        float lr_t, float b1, float b2, float eps) {
  const int i = blockIdx.x, r = rows[i];
  int old = 0;
  if (threadIdx.x == 0) old = atomicCAS(&slot[r], -1, 0x40000000 + i);
  old = __shfl(old, 0, 64);
  if (old != -1) return;                    // swept with its gradient already (the step's union), or claimed by another sample
  for (int c = threadIdx.x; c < dim; c += 64) {
    const size_t e = (size_t)r * dim + c;
    float pp = p[e], mm = m[e], vv = v[e];
    adam1(pp, 0.f, mm, vv, lr_t, b1, b2, eps);
    p[e] = pp; m[e] = mm; v[e] = vv;
    if (p2) {
      pp = p2[e]; mm = m2[e]; vv = v2[e];
      adam1(pp, 0.f, mm, vv, lr_t, b1, b2, eps);
      p2[e] = pp; m2[e] = mm; v2[e] = vv;
    }
  }
}
void launch_defer_rows(float* p, float* m, float* v, float* p2, float* m2, float* v2, const int32_t* rows, int n, int32_t* slot,
                       int dim, float lr_t, float b1, float b2, float eps, hipStream_t s) {
  if (n) hipLaunchKernelGGL(defer_rows_kernel, dim3(n), dim3(64), 0, s, p, m, v, p2, m2, v2, rows, slot, dim, lr_t, b1, b2, eps);
}

// ---------------------------------------------------------------------------------------------
// LAZY dense Adam for large user tables.  TF-1 Keras densifies the embedding gradient, so every row takes an update every
// step -- for a row outside the batch the zero-gradient one, which depends on nothing but the row's own (p, m, v) and the
// step's lr_t.  Instead of sweeping N_u rows per step (24 B of HBM traffic per element: 4.8 GB at 1 M users, 1.1 ms), a row
// remembers the last iteration it is current for (last[row]) and takes its pending zero-gradient updates, in order, with
// the logged lr_t of each, when it is next READ (forward gather) or next UPDATED with a gradient: the same sequence of
// fp32 operations per element as the dense sweep (same adam1, no FMA contraction), hence bit-identical values -- only the
// time they are computed moves, and the work turns from HBM traffic into registers.  lstur_flush_tables brings every row
// up to date (before the host reads the table).
// ---------------------------------------------------------------------------------------------
// One THREAD per (sample, column): a row that waited 2 000 iterations is a 2 000-long dependent chain of sqrt / divide per
// element, so the parallelism has to come from the elements (a wave per row took 1.7 ms for 512 rows at 1 M users; this
// form ~0.3 ms).  first[i] (smallest j with rows[j] == rows[i], lazy_first_kernel) picks one sample per distinct row; the
// rows' `last` is advanced by lazy_mark_kernel once every table has been brought up.
__global__ void lazy_first_kernel(const int32_t* __restrict__ rows, int n, int32_t* __restrict__ first) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int r = rows[i];
  int f = i;
  for (int j = 0; j < i; ++j)
    if (rows[j] == r) {
      f = j;
      break;
    }
  first[i] = f;
}
__global__ __launch_bounds__(256) void lazy_catchup_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                          const int32_t* __restrict__ rows, const int32_t* __restrict__ first,
                                                          int n, const int32_t* __restrict__ last,
                                                          const float* __restrict__ lrt, int upto, int dim, float b1, float b2,
                                                          float eps, int mask) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * dim) return;
  const int i = idx / dim, c = idx - i * dim;
  if (first[i] != i) return;
  const int r = rows[i], l = last[r];
  if (l >= upto) return;
  const size_t e = (size_t)r * dim + c;
  float pp = p[e], mm = m[e], vv = v[e];
  for (int tt = l + 1; tt <= upto; ++tt) adam1(pp, 0.f, mm, vv, lrt[tt & mask], b1, b2, eps);
  p[e] = pp;
  m[e] = mm;
  v[e] = vv;
}
__global__ void lazy_mark_kernel(const int32_t* __restrict__ rows, const int32_t* __restrict__ first, int n,
                                 int32_t* __restrict__ last, int upto) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && first[i] == i && last[rows[i]] < upto) last[rows[i]] = upto;
}
// rows[0..n) current for iteration `upto` in table (p, m, v) and, when p2 != NULL, in the second table
void launch_lazy_catchup(float* p, float* m, float* v, float* p2, float* m2, float* v2, const int32_t* rows, int n,
                         int32_t* first, int32_t* last, const float* lrt, int upto, int dim, float b1, float b2, float eps,
                         int mask, hipStream_t s) {
  if (n <= 0 || upto <= 0) return;
  hipLaunchKernelGGL(lazy_first_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, rows, n, first);
  hipLaunchKernelGGL(lazy_catchup_kernel, dim3(cdiv(n * dim, 256)), dim3(256), 0, s, p, m, v, rows, (const int32_t*)first, n,
                     (const int32_t*)last, lrt, upto, dim, b1, b2, eps, mask);
  if (p2)
    hipLaunchKernelGGL(lazy_catchup_kernel, dim3(cdiv(n * dim, 256)), dim3(256), 0, s, p2, m2, v2, rows, (const int32_t*)first,
                       n, (const int32_t*)last, lrt, upto, dim, b1, b2, eps, mask);
  hipLaunchKernelGGL(lazy_mark_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, rows, (const int32_t*)first, n, last, upto);
}
// iteration t for the batch's rows (first occurrences: slot[row] == i, set by the row combine): pending zero-gradient
// updates up to t - 1, then the step with the combined gradient; one thread per (sample, column).  lazy_rows_mark_kernel
// then advances `last` and logs lr_t for the catch-ups to come.
__global__ __launch_bounds__(256) void lazy_rows_adam_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                            const int32_t* __restrict__ rows, const int32_t* __restrict__ slot,
                                                            const float* __restrict__ gsum, int n, int dim,
                                                            const int32_t* __restrict__ last, const float* __restrict__ lrt,
                                                            int t, float lr_t, float b1, float b2, float eps, int mask) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * dim) return;
  const int i = idx / dim, c = idx - i * dim, r = rows[i];
  if (slot[r] != i) return;
  const int l = last[r];
  const size_t e = (size_t)r * dim + c;
  float pp = p[e], mm = m[e], vv = v[e];
  for (int tt = l + 1; tt < t; ++tt) adam1(pp, 0.f, mm, vv, lrt[tt & mask], b1, b2, eps);
  adam1(pp, gsum[(size_t)i * dim + c], mm, vv, lr_t, b1, b2, eps);
  p[e] = pp;
  m[e] = mm;
  v[e] = vv;
}
__global__ void lazy_rows_mark_kernel(const int32_t* __restrict__ rows, const int32_t* __restrict__ slot, int n,
                                      int32_t* __restrict__ last, float* __restrict__ lrt, int t, float lr_t, int mask) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) lrt[t & mask] = lr_t;
  if (i < n && slot[rows[i]] == i) last[rows[i]] = t;
}
void launch_lazy_rows_adam(float* p, float* m, float* v, const int32_t* rows, const int32_t* slot, const float* gsum, int n,
                           int dim, const int32_t* last, const float* lrt, int t, float lr_t, float b1, float b2, float eps,
                           int mask, hipStream_t s) {
  if (n > 0)
    hipLaunchKernelGGL(lazy_rows_adam_kernel, dim3(cdiv(n * dim, 256)), dim3(256), 0, s, p, m, v, rows, slot, gsum, n, dim, last,
                       lrt, t, lr_t, b1, b2, eps, mask);
}
void launch_lazy_rows_mark(const int32_t* rows, const int32_t* slot, int n, int32_t* last, float* lrt, int t, float lr_t,
                           int mask, hipStream_t s) {
  hipLaunchKernelGGL(lazy_rows_mark_kernel, dim3(std::max(1, cdiv(n, 256))), dim3(256), 0, s, rows, slot, n, last, lrt, t, lr_t,
                     mask);
}
// every row up to iteration `upto` (before the host reads the table); the caller then sets last[] = upto
__global__ __launch_bounds__(256) void lazy_flush_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                         const int32_t* __restrict__ last, const float* __restrict__ lrt,
                                                         int upto, size_t total, int dim, float b1, float b2, float eps,
                                                         int mask) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int l = last[e / dim];
    if (l >= upto) continue;
    float pp = p[e], mm = m[e], vv = v[e];
    for (int tt = l + 1; tt <= upto; ++tt) adam1(pp, 0.f, mm, vv, lrt[tt & mask], b1, b2, eps);
    p[e] = pp;
    m[e] = mm;
    v[e] = vv;
  }
}
void launch_lazy_flush(float* p, float* m, float* v, const int32_t* last, const float* lrt, int upto, int n_rows, int dim,
                       float b1, float b2, float eps, int mask, hipStream_t s) {
  const size_t total = (size_t)n_rows * dim;
  if (total && upto > 0)
    hipLaunchKernelGGL(lazy_flush_kernel, dim3((unsigned)std::min<size_t>(cdivz(total, 256), 256 * 32)), dim3(256), 0, s, p, m,
                       v, last, lrt, upto, total, dim, b1, b2, eps, mask);
}

// first[i] = smallest j with rows[j] == rows[i]
__global__ void first_occurrence_kernel(const int32_t* __restrict__ rows, int n, int32_t* __restrict__ first) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int r = rows[i], f = i;
  for (int j = 0; j < i; ++j)
    if (rows[j] == r) {
      f = j;
      break;
    }
  first[i] = f;
}
// gsum[i,:] = sum over j >= i (ascending) with rows[j] == rows[i] of g[j,:], for the first occurrence i of each row.
// One wave per sample row; matches are found 64 rows at a time by ballot, so the usual case (no duplicate) costs
// n/64 loads instead of n.
// rows_f != NULL: the row numbers arrive as floats (the reduced tail of a data-parallel step, read in place); every row's
// number is then also left as an int32 in rows_out for the kernels that follow
template <bool RF>
__device__ __forceinline__ void combine_row_t(const int32_t* __restrict__ rows, const float* __restrict__ rows_f,
                                              int32_t* __restrict__ rows_out, const float* __restrict__ g, int n, int dim,
                                              float* __restrict__ gsum, int32_t* __restrict__ slot, int i, int lane);
__device__ __forceinline__ void combine_row(const int32_t* __restrict__ rows, const float* __restrict__ g, int n, int dim,
                                            float* __restrict__ gsum, int32_t* __restrict__ slot, int i, int lane) {
  combine_row_t<false>(rows, nullptr, nullptr, g, n, dim, gsum, slot, i, lane);
}
template <bool RF>
__device__ __forceinline__ void combine_row_t(const int32_t* __restrict__ rows, const float* __restrict__ rows_f,
                                              int32_t* __restrict__ rows_out, const float* __restrict__ g, int n, int dim,
                                              float* __restrict__ gsum, int32_t* __restrict__ slot, int i, int lane) {
  auto R = [&](int j) { return RF ? (int)(rows_f[j] + 0.5f) : rows[j]; };
  const int r = R(i);
  if (RF && lane == 0) rows_out[i] = r;
  bool dup = false;
  for (int j = lane; j < i; j += 64) dup |= R(j) == r;
  if (__any(dup)) return;   // not the first occurrence
  for (int c0 = 0; c0 < dim; c0 += 512) {   // 512 columns per pass
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = (c0 + lane + 64 * k < dim) ? g[(size_t)i * dim + c0 + lane + 64 * k] : 0.f;
    for (int j0 = (i + 1) & ~63; j0 < n; j0 += 64) {
      const int j = j0 + lane;
      unsigned long long mask = __ballot(j > i && j < n && R(min(j, n - 1)) == r);
      while (mask) {   // ascending sample order
        const int jj = j0 + __builtin_ctzll(mask);
        mask &= mask - 1;
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (c0 + lane + 64 * k < dim) acc[k] += g[(size_t)jj * dim + c0 + lane + 64 * k];
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (c0 + lane + 64 * k < dim) gsum[(size_t)i * dim + c0 + lane + 64 * k] = acc[k];
  }
  if (lane == 0) slot[r] = i;
}
__global__ __launch_bounds__(64) void combine_rows_kernel(const int32_t* __restrict__ rows, const float* __restrict__ g,
                                                          int n, int dim, float* __restrict__ gsum,
                                                          int32_t* __restrict__ slot) {
  combine_row(rows, g, n, dim, gsum, slot, blockIdx.x, threadIdx.x);
}
void launch_user_grad_combine(const int32_t* rows, const float* g, int n, int dim, int32_t* first, float* gsum,
                              int32_t* slot, hipStream_t s) {
  (void)first;
  if (n) hipLaunchKernelGGL(combine_rows_kernel, dim3(n), dim3(64), 0, s, rows, g, n, dim, gsum, slot);
}
// ---- data-parallel exchange of the sparse user-row gradients inside the dense bucket's ONE all-reduce -------------
// tail (behind the dense gradients in the caller's bucket) = [world x B row numbers as float | world x B x Ue gradients
// (| the same of a second table)]; a rank fills its own B-slot of every region and zeroes the others, so the sum over ranks
// IS the concatenation (x + 0 + ... + 0 = x) -- and with full batches (n == B) the reduced gradient region is the row-gradient
// matrix [world * n, Ue] as it stands (lstur_adam_step_dp reads it in place).
__global__ void dp_stage_kernel(const int32_t* __restrict__ rows, const float* __restrict__ g, const float* __restrict__ g2,
                                int n, int B, int Ue, float* __restrict__ tail, int rank, int world) {
  const int ntab = g2 ? 2 : 1, WB = world * B;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= WB * (1 + ntab * Ue)) return;
  float v = 0.f;
  if (i < WB) {
    const int k = i / B, e = i - k * B;
    if (k == rank && e < n) v = (float)rows[e];
  } else {
    const int x = i - WB, tab = x / (WB * Ue), y = x - tab * (WB * Ue), j = y / Ue, k = j / B, e = j - k * B;
    if (k == rank && e < n) v = (tab ? g2 : g)[(size_t)e * Ue + (y - j * Ue)];
  }
  tail[i] = v;
}
// g2 != NULL: a second table's row gradients travel behind the first's (iigru / iicgru)
void launch_dp_stage(const int32_t* rows, const float* g, const float* g2, int n, int B, int Ue, float* tail, int rank,
                     int world, hipStream_t s) {
  const int total = world * B * (1 + (g2 ? 2 : 1) * Ue);
  hipLaunchKernelGGL(dp_stage_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, rows, g, g2, n, B, Ue, tail, rank, world);
}
// reduced tail -> rows int32 [world*n], grads [world*n, Ue] (and grads2) in rank order (every rank passed the same n)
__global__ void dp_unpack_kernel(const float* __restrict__ tail, int n, int B, int Ue, int world, int32_t* __restrict__ rows,
                                 float* __restrict__ g, float* __restrict__ g2) {
  const int ntab = g2 ? 2 : 1, WB = world * B, Wn = world * n;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Wn * (1 + ntab * Ue)) return;
  if (i < Wn) {
    const int k = i / n, e = i - k * n;
    rows[i] = (int32_t)(tail[k * B + e] + 0.5f);
  } else {
    const int x = i - Wn, tab = x / (Wn * Ue), y = x - tab * (Wn * Ue), j = y / Ue, k = j / n, e = j - k * n;
    (tab ? g2 : g)[y] = tail[(size_t)WB + (size_t)tab * WB * Ue + (size_t)(k * B + e) * Ue + (y - j * Ue)];
  }
}
void launch_dp_unpack(const float* tail, int n, int B, int Ue, int world, int32_t* rows, float* g, float* g2, hipStream_t s) {
  const int total = world * n * (1 + (g2 ? 2 : 1) * Ue);
  if (total) hipLaunchKernelGGL(dp_unpack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, tail, n, B, Ue, world, rows, g, g2);
}
// the staging as jobs of the backward pass's ONE reduction launch (a "reduction" of one slice is a copy, of a negative
// slice count a zero fill): rank's slot of every region, zeros in front of and behind it.  rows_f: the batch's row numbers
// as floats (the history-mask launch leaves them).  Returns the number of jobs written (<= 9).
int dp_stage_jobs(const float* rows_f, const float* g, const float* g2, int n, int B, int Ue, float* tail, int rank, int world,
                  ReduceJob* jobs) {
  int nj = 0;
  auto region = [&](const float* src, float* base, int width) {   // a region of world * B * width floats
    const int lo = rank * B * width, used = n * width, total = world * B * width;
    if (lo) jobs[nj++] = ReduceJob{src, base, 1, lo, lo, -1, 0, 0.f, 0.f};
    jobs[nj++] = ReduceJob{src, base + lo, 1, used, used, 1, 0, 1.0f, 0.f};
    if (total - lo - used) jobs[nj++] = ReduceJob{src, base + lo + used, 1, total - lo - used, total - lo - used, -1, 0, 0.f, 0.f};
  };
  region(rows_f, tail, 1);
  region(g, tail + (size_t)world * B, Ue);
  if (g2) region(g2, tail + (size_t)world * B * (1 + Ue), Ue);
  return nj;
}

__global__ void slot_clear_kernel(const int32_t* __restrict__ rows, int n, int32_t* __restrict__ slot) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) slot[rows[i]] = -1;
}
void launch_user_slot_clear(const int32_t* rows, int n, int32_t* slot, hipStream_t s) {
  if (n) hipLaunchKernelGGL(slot_clear_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, rows, n, slot);
}
__global__ void fill_i32_kernel(int32_t* p, int32_t v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
void launch_fill_i32(int32_t* p, int32_t v, size_t n, hipStream_t s) {
  if (n) hipLaunchKernelGGL(fill_i32_kernel, dim3((unsigned)std::min<size_t>(cdivz(n, 256), 4096)), dim3(256), 0, s, p, v, n);
}

// ---------------------------------------------------------------------------------------------
// models.ComputeMasking(0) on the clicked titles + multiply (reference models.py:20-30,
// task/paper.py:644-645) and the user encoder's Masking() step mask (paper.py:592).  One wave per
// history slot.
// ---------------------------------------------------------------------------------------------
// The trailing blocks of the same launch gather the user-embedding rows of the batch (paper.py:589-591) and keep the
// row numbers for the sparse table update.
__global__ __launch_bounds__(256) void hist_mask_kernel(const int32_t* __restrict__ ids, float* __restrict__ hv,
                                                        uint8_t* __restrict__ hmask, uint8_t* __restrict__ gmask,
                                                        int nW, int T, int U, int mask_blocks,
                                                        const float* __restrict__ table, const int32_t* __restrict__ user,
                                                        int32_t* __restrict__ user_rows, float* __restrict__ uvec, int n,
                                                        int Ue, float* __restrict__ user_rows_f) {
  if ((int)blockIdx.x >= mask_blocks) {
    const int b = (blockIdx.x - mask_blocks) * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
    if (b >= n) return;
    const int r = user[b];
    if (l == 0) {
      user_rows[b] = r;
      if (user_rows_f) user_rows_f[b] = (float)r;   // (the data-parallel staging copies these into the gradient bucket's tail)
    }
    for (int c = l; c < Ue; c += 64) uvec[(size_t)b * Ue + c] = table[(size_t)r * Ue + c];
    return;
  }
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= nW) return;
  bool any_id = false;
  for (int t = lane; t < T; t += 64) any_id |= ids[(size_t)row * T + t] != 0;
  bool hm = __any(any_id);
  bool nz = false;
  for (int c = lane; c < U; c += 64) {
    float v = hv[(size_t)row * U + c];
    v = hm ? v : 0.f;
    hv[(size_t)row * U + c] = v;
    nz |= v != 0.f;
  }
  bool gm = __any(nz);
  if (lane == 0) {
    hmask[row] = hm;
    gmask[row] = gm;
  }
}
void launch_hist_mask(const int32_t* hist_ids, float* hv, uint8_t* hmask, uint8_t* gmask, int nW, int T, int U,
                      hipStream_t s, const float* table, const int32_t* user, int32_t* user_rows, float* uvec, int n,
                      int Ue, float* user_rows_f) {
  const int mb = cdiv(nW, 4), ub = table ? cdiv(n, 4) : 0;
  if (mb + ub)
    hipLaunchKernelGGL(hist_mask_kernel, dim3(mb + ub), dim3(256), 0, s, hist_ids, hv, hmask, gmask, nW, T, U, mb, table,
                       user, user_rows, uvec, n, Ue, user_rows_f);
}

__global__ void scale_rows_u8_kernel(float* __restrict__ x, const uint8_t* __restrict__ mask, size_t total, int cols) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total && !mask[i / cols]) x[i] = 0.f;
}
void launch_scale_rows_u8(float* x, const uint8_t* mask, int rows, int cols, hipStream_t s) {
  size_t total = (size_t)rows * cols;
  if (total) hipLaunchKernelGGL(scale_rows_u8_kernel, dim3((unsigned)cdivz(total, 256)), dim3(256), 0, s, x, mask, total, cols);
}

__global__ void gather_rows_kernel(const float* __restrict__ table, const int32_t* __restrict__ rows,
                                   float* __restrict__ out, int n, int dim) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * dim) return;
  int b = i / dim, c = i - b * dim;
  out[i] = table[(size_t)rows[b] * dim + c];
}
void launch_gather_rows(const float* table, const int32_t* rows, float* out, int n, int dim, hipStream_t s) {
  if (n * dim) hipLaunchKernelGGL(gather_rows_kernel, dim3(cdiv(n * dim, 256)), dim3(256), 0, s, table, rows, out, n, dim);
}

__global__ void concat2_kernel(const float* __restrict__ a, int ac, const float* __restrict__ b, int bc,
                               float* __restrict__ out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int w = ac + bc;
  if (i >= n * w) return;
  int r = i / w, c = i - r * w;
  out[i] = c < ac ? a[(size_t)r * ac + c] : b[(size_t)r * bc + (c - ac)];
}
void launch_concat2(const float* a, int a_cols, const float* bm, int b_cols, float* out, int n, hipStream_t s) {
  int total = n * (a_cols + b_cols);
  if (total) hipLaunchKernelGGL(concat2_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, a, a_cols, bm, b_cols, out, n);
}
// ---- listwise score models ddot / dnn (Seq2VecPaperSoftmax._score_model, task/paper.py:448-455) -----------------
__global__ void tanh_inplace_kernel(float* __restrict__ x, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = tanhf(x[i]);
}
void launch_tanh_inplace(float* x, size_t n, hipStream_t s) {
  if (n) hipLaunchKernelGGL(tanh_inplace_kernel, dim3((unsigned)cdivz(n, 256)), dim3(256), 0, s, x, n);
}
// g *= 1 - y^2   (y = tanh output)
__global__ void tanh_bwd_kernel(float* __restrict__ g, const float* __restrict__ y, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) g[i] *= 1.0f - y[i] * y[i];
}
void launch_tanh_bwd(float* g, const float* y, size_t n, hipStream_t s) {
  if (n) hipLaunchKernelGGL(tanh_bwd_kernel, dim3((unsigned)cdivz(n, 256)), dim3(256), 0, s, g, y, n);
}
// out[b*K1 + j] = [u[b] | d[b*K1 + j]]   (keras concatenate of the user vector with every candidate, paper.py:449)
__global__ void concat_rep_kernel(const float* __restrict__ u, int uc, int K1, const float* __restrict__ d, int dc,
                                  float* __restrict__ out, int rows) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, w = uc + dc;
  if (i >= rows * w) return;
  const int r = i / w, c = i - r * w;
  out[i] = c < uc ? u[(size_t)(r / K1) * uc + c] : d[(size_t)r * dc + (c - uc)];
}
void launch_concat_rep(const float* u, int uc, int K1, const float* d, int dc, float* out, int rows, hipStream_t s) {
  const int total = rows * (uc + dc);
  if (total) hipLaunchKernelGGL(concat_rep_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, u, uc, K1, d, dc, out, rows);
}
// dhid[r][i] = ds[r] * w2[i] * (pre[r][i] > 0)     (Dense(1) on a ReLU hidden layer)
__global__ void mlp_dhid_kernel(const float* __restrict__ pre, const float* __restrict__ ds, const float* __restrict__ w2,
                                float* __restrict__ dhid, int rows, int U) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * U) return;
  const int r = i / U, c = i - r * U;
  dhid[i] = pre[i] > 0.f ? ds[r] * w2[c] : 0.f;
}
void launch_mlp_dhid(const float* pre, const float* ds, const float* w2, float* dhid, int rows, int U, hipStream_t s) {
  if (rows * U) hipLaunchKernelGGL(mlp_dhid_kernel, dim3(cdiv(rows * U, 256)), dim3(256), 0, s, pre, ds, w2, dhid, rows, U);
}
// dcat [n*K1, uc + dc] -> du[b] = sum_j dcat[b*K1 + j][:uc] (fixed order), dd[b*K1 + j] = dcat[..][uc:]
__global__ void split_sum_kernel(const float* __restrict__ dcat, int uc, int dc, int K1, float* __restrict__ du,
                                 float* __restrict__ dd, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, w = uc + dc;
  if (i < n * uc) {
    const int b = i / uc, c = i - b * uc;
    float acc = 0.f;
    for (int j = 0; j < K1; ++j) acc += dcat[(size_t)(b * K1 + j) * w + c];
    du[i] = acc;
  }
  if (i < n * K1 * dc) {
    const int r = i / dc, c = i - r * dc;
    dd[i] = dcat[(size_t)r * w + uc + c];
  }
}
void launch_split_sum(const float* dcat, int uc, int dc, int K1, float* du, float* dd, int n, hipStream_t s) {
  const int total = std::max(n * uc, n * K1 * dc);
  if (total) hipLaunchKernelGGL(split_sum_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, dcat, uc, dc, K1, du, dd, n);
}
// niavg user encoder (models.GlobalAveragePoolingMaskSupport, models.py:433-435): u[b] = sum_w x[b,w] / (cnt_b + eps);
// backward (du != NULL): dx[b,w] = mask[b,w] * du[b] / (cnt_b + eps).  x rows of masked steps are already zero.
__global__ void masked_mean_kernel(const float* __restrict__ x, const uint8_t* __restrict__ mask, float* __restrict__ u,
                                   const float* __restrict__ du, float* __restrict__ dx, int n, int W, int U) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * U) return;
  const int b = i / U, c = i - b * U;
  float cnt = 0.f;
  for (int w = 0; w < W; ++w) cnt += mask[b * W + w] ? 1.f : 0.f;
  const float inv = 1.0f / (cnt + LSTUR_EPS);
  if (!du) {
    float acc = 0.f;
    for (int w = 0; w < W; ++w) acc += x[((size_t)b * W + w) * U + c];
    u[i] = acc * inv;
  } else {
    const float g = du[i] * inv;
    for (int w = 0; w < W; ++w) dx[((size_t)b * W + w) * U + c] = mask[b * W + w] ? g : 0.f;
  }
}
void launch_masked_mean(const float* x, const uint8_t* mask, float* u, const float* du, float* dx, int n, int W, int U,
                        hipStream_t s) {
  if (n * U) hipLaunchKernelGGL(masked_mean_kernel, dim3(cdiv(n * U, 256)), dim3(256), 0, s, x, mask, u, du, dx, n, W, U);
}
// out = a + b
__global__ void add2_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] + b[i];
}
void launch_add2(const float* a, const float* b, float* out, size_t n, hipStream_t s) {
  if (n) hipLaunchKernelGGL(add2_kernel, dim3((unsigned)cdivz(n, 256)), dim3(256), 0, s, a, b, out, n);
}
__global__ void fill_f32_kernel(float* __restrict__ p, float v, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
void launch_fill_f32(float* p, float v, size_t n, hipStream_t s) {
  if (n) hipLaunchKernelGGL(fill_f32_kernel, dim3((unsigned)cdivz(n, 256)), dim3(256), 0, s, p, v, n);
}
// whole-row inverted dropout (keras Dropout(p, noise_shape=(None, 1)), the `dgru` user embedding, paper.py:610):
// row b is kept iff keep_elem(key, b), kept rows are scaled by 1 / (1 - p)
__global__ void drop_rows_kernel(float* __restrict__ x, int n, int dim, uint32_t key, uint32_t thresh, float scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * dim) return;
  const int b = i / dim;
  x[i] = keep_elem(key, (uint32_t)b, thresh) ? x[i] * scale : 0.f;
}
void launch_drop_rows(float* x, int n, int dim, float p, uint32_t key, hipStream_t s) {
  if (n * dim && p > 0.f)
    hipLaunchKernelGGL(drop_rows_kernel, dim3(cdiv(n * dim, 256)), dim3(256), 0, s, x, n, dim, key, dropout_thresh(p),
                       1.0f / (1.0f - p));
}

__global__ void split2_kernel(const float* __restrict__ in, int ac, int bc, float* __restrict__ a,
                              float* __restrict__ b, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int w = ac + bc;
  if (i >= n * w) return;
  int r = i / w, c = i - r * w;
  if (c < ac)
    a[(size_t)r * ac + c] = in[i];
  else
    b[(size_t)r * bc + (c - ac)] = in[i];
}
void launch_split2(const float* in, int a_cols, int b_cols, float* a, float* bm, int n, hipStream_t s) {
  int total = n * (a_cols + b_cols);
  if (total) hipLaunchKernelGGL(split2_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, in, a_cols, b_cols, a, bm, n);
}

// ---------------------------------------------------------------------------------------------
// dot scorer over the 1+K candidates + softmax + Keras categorical_crossentropy on probabilities
// (reference task/paper.py:446-447, 460-464, 658) with label [1,0,..,0], and the gradient wrt u, d.
// One wave per sample.
// ---------------------------------------------------------------------------------------------
constexpr int kMaxCands = 64;

__global__ __launch_bounds__(256) void score_kernel(const float* __restrict__ u, const float* __restrict__ d,
                                                    float* __restrict__ logits, float* __restrict__ probs,
                                                    float* __restrict__ loss_i, float* __restrict__ correct_i,
                                                    float* __restrict__ du, float* __restrict__ dd, int n, int K1,
                                                    int U, float inv_n) {
  __shared__ float zs[4][kMaxCands];
  int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int b = blockIdx.x * 4 + wv;
  if (b >= n) return;
  const float* ub = u + (size_t)b * U;
  const float* db = d + (size_t)b * K1 * U;
  for (int j = 0; j < K1; ++j) {
    float acc = 0.f;
    for (int c = lane; c < U; c += 64) acc += ub[c] * db[(size_t)j * U + c];
    acc = wave_sum(acc);
    if (lane == 0) zs[wv][j] = acc;
  }
  __builtin_amdgcn_wave_barrier();
  float zmax = -INFINITY;
  for (int j = 0; j < K1; ++j) zmax = fmaxf(zmax, zs[wv][j]);
  float esum = 0.f;
  for (int j = 0; j < K1; ++j) esum += expf(zs[wv][j] - zmax);
  float S = 0.f, p0 = 0.f, pmax = 0.f;
  for (int j = 0; j < K1; ++j) {
    float pj = expf(zs[wv][j] - zmax) / esum;
    S += pj;
    if (j == 0) p0 = pj;
    pmax = fmaxf(pmax, pj);
  }
  float q0 = p0 / S;
  float cq0 = fminf(fmaxf(q0, LSTUR_EPS), 1.0f - LSTUR_EPS);
  bool inside = (q0 >= LSTUR_EPS) && (q0 <= 1.0f - LSTUR_EPS);
  float gq0 = inside ? -1.0f / cq0 : 0.f;
  // dL/dp_j = gq_j / S - (gq0 p0) / S^2 ; dz_j = p_j (dp_j - sum_i dp_i p_i)
  float common = (gq0 * p0) / (S * S);
  float dot = 0.f;
  for (int j = 0; j < K1; ++j) {
    float pj = expf(zs[wv][j] - zmax) / esum;
    float dpj = (j == 0 ? gq0 / S : 0.f) - common;
    dot += dpj * pj;
  }
  if (lane == 0) {
    loss_i[b] = -logf(cq0);
    correct_i[b] = (p0 >= pmax) ? 1.f : 0.f;
  }
  for (int j = lane; j < K1; j += 64) {
    logits[(size_t)b * K1 + j] = zs[wv][j];
    probs[(size_t)b * K1 + j] = expf(zs[wv][j] - zmax) / esum;
  }
  if (du == nullptr) return;
  for (int c = lane; c < U; c += 64) {
    float uc = ub[c], acc = 0.f;
    for (int j = 0; j < K1; ++j) {
      float pj = expf(zs[wv][j] - zmax) / esum;
      float dpj = (j == 0 ? gq0 / S : 0.f) - common;
      float dz = pj * (dpj - dot) * inv_n;
      acc += dz * db[(size_t)j * U + c];
      dd[((size_t)b * K1 + j) * U + c] = dz * uc;
    }
    du[(size_t)b * U + c] = acc;
  }
}
// The same arithmetic (same summation orders: bitwise the same results) for K1 <= KM candidates and U <= 64 UC with the user
// vector and all candidate vectors of the sample in registers: every load is in flight at once, the softmax terms are
// evaluated once, nothing is read twice.  (The general kernel above walks the candidates one dependent load-reduce round
// trip after the other and re-evaluates expf in every loop: 15 us on the step's critical path for 64 x 5 dot products.)
template <int KM, int UC>
__global__ __launch_bounds__(256) void score_fast_kernel(const float* __restrict__ u, const float* __restrict__ d,
                                                         float* __restrict__ logits, float* __restrict__ probs,
                                                         float* __restrict__ loss_i, float* __restrict__ correct_i,
                                                         float* __restrict__ du, float* __restrict__ dd, int n, int K1,
                                                         int U, float inv_n) {
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + wv;
  if (b >= n) return;
  const __amdgpu_buffer_rsrc_t rsU = make_rsrc(u + (size_t)b * U, (uint32_t)U * 4u);
  float uv[UC];
#pragma unroll
  for (int c = 0; c < UC; ++c) uv[c] = buf_load1(rsU, (uint32_t)(lane + 64 * c) * 4u, 0);      // past U: 0
  score_row<KM, UC>(uv, d, logits, probs, loss_i, correct_i, du, dd, b, lane, K1, U, inv_n);
}
void launch_score(const float* u, const float* d, float* logits, float* probs, float* loss_i, float* correct_i,
                  float* du, float* dd, int n, int K1, int U, float inv_n, hipStream_t s) {
  if (!n) return;
  if (K1 <= 8 && U <= 256)   // (launch_k: a fork event armed by the step rides on this launch's completion)
    launch_k((score_fast_kernel<8, 4>), dim3(cdiv(n, 4)), dim3(256), 0, s, u, d, logits, probs, loss_i, correct_i, du, dd, n, K1, U,
             inv_n);
  else
    hipLaunchKernelGGL(score_kernel, dim3(cdiv(n, 4)), dim3(256), 0, s, u, d, logits, probs, loss_i, correct_i, du, dd, n, K1, U, inv_n);
}

__global__ __launch_bounds__(256) void predict_score_kernel(const float* __restrict__ u, const float* __restrict__ d,
                                                            float* __restrict__ scores, int n, int U) {
  int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= n) return;
  float acc = 0.f;
  for (int c = lane; c < U; c += 64) acc += u[(size_t)b * U + c] * d[(size_t)b * U + c];
  acc = wave_sum(acc);
  if (lane == 0) scores[b] = 1.0f / (1.0f + expf(-acc));
}
void launch_predict_score(const float* u, const float* d, float* scores, int n, int U, hipStream_t s) {
  if (n) hipLaunchKernelGGL(predict_score_kernel, dim3(cdiv(n, 4)), dim3(256), 0, s, u, d, scores, n, U);
}

// ---- pointwise head: Seq2VecPaper.score_encoder / Seq2VecPaperDot + Seq2Vec.loss -------------------
//   loss_i = -0.5 (1+K) (y log(p + 1e-8) gain + (1-y) log(1 - p + 1e-8) / K)        task/seq2vec.py:213-216
// One wave per sample.  ds = dloss_i/ds_i * scale.
__device__ __forceinline__ void point_loss(float sc, float y, float gain, float K, float scale, float* prob,
                                           float* loss, float* ok, float* ds) {
  const float p = 1.0f / (1.0f + expf(-sc));
  const float c = 0.5f * (1.0f + K), e = 1e-8f;
  *prob = p;
  *loss = -c * (y * logf(p + e) * gain + (1.0f - y) * logf(1.0f - p + e) / K);
  *ok = ((p > 0.5f) == (y > 0.5f)) ? 1.f : 0.f;
  const float dp = -c * (y * gain / (p + e) - (1.0f - y) / (K * (1.0f - p + e)));
  *ds = dp * p * (1.0f - p) * scale;
}
__global__ __launch_bounds__(256) void point_dot_kernel(const float* __restrict__ u, const float* __restrict__ d,
                                                        const float* __restrict__ labels, float* __restrict__ logits,
                                                        float* __restrict__ probs, float* __restrict__ loss_i,
                                                        float* __restrict__ ok_i, float* __restrict__ du,
                                                        float* __restrict__ dd, int n, int U, float gain, float K,
                                                        float scale) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= n) return;
  float acc = 0.f;
  for (int c = lane; c < U; c += 64) acc += u[(size_t)b * U + c] * d[(size_t)b * U + c];
  acc = wave_sum(acc);
  float p, l, ok, ds;
  point_loss(acc, labels[b], gain, K, scale, &p, &l, &ok, &ds);
  if (lane == 0) {
    logits[b] = acc;
    probs[b] = p;
    loss_i[b] = l;
    ok_i[b] = ok;
  }
  if (du)
    for (int c = lane; c < U; c += 64) {
      du[(size_t)b * U + c] = ds * d[(size_t)b * U + c];
      dd[(size_t)b * U + c] = ds * u[(size_t)b * U + c];
    }
}
void launch_point_dot(const float* u, const float* d, const float* labels, float* logits, float* probs, float* loss_i,
                      float* ok_i, float* du, float* dd, int n, int U, float gain, int K, float scale, hipStream_t s) {
  if (n)
    hipLaunchKernelGGL(point_dot_kernel, dim3(cdiv(n, 4)), dim3(256), 0, s, u, d, labels, logits, probs, loss_i, ok_i,
                       du, dd, n, U, gain, (float)K, scale);
}
__global__ __launch_bounds__(256) void point_mlp_out_kernel(const float* __restrict__ pre, const float* __restrict__ w2,
                                                            const float* __restrict__ b2,
                                                            const float* __restrict__ labels, float* __restrict__ logits,
                                                            float* __restrict__ probs, float* __restrict__ loss_i,
                                                            float* __restrict__ ok_i, float* __restrict__ dsv,
                                                            float* __restrict__ dhid, int n, int U, float gain, float K,
                                                            float scale) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= n) return;
  float acc = 0.f;
  for (int c = lane; c < U; c += 64) acc += fmaxf(pre[(size_t)b * U + c], 0.f) * w2[c];
  acc = wave_sum(acc) + b2[0];
  float p, l, ok, ds;
  point_loss(acc, labels ? labels[b] : 0.f, gain, K, scale, &p, &l, &ok, &ds);
  if (lane == 0) {
    logits[b] = acc;
    probs[b] = p;
    if (loss_i) {
      loss_i[b] = l;
      ok_i[b] = ok;
    }
    if (dsv) dsv[b] = ds;
  }
  if (dhid)
    for (int c = lane; c < U; c += 64) dhid[(size_t)b * U + c] = pre[(size_t)b * U + c] > 0.f ? ds * w2[c] : 0.f;
}
void launch_point_mlp_out(const float* pre, const float* w2, const float* b2, const float* labels, float* logits,
                          float* probs, float* loss_i, float* ok_i, float* ds, float* dhid, int n, int U, float gain,
                          int K, float scale, hipStream_t s) {
  if (n)
    hipLaunchKernelGGL(point_mlp_out_kernel, dim3(cdiv(n, 4)), dim3(256), 0, s, pre, w2, b2, labels, logits, probs,
                       loss_i, ok_i, ds, dhid, n, U, gain, (float)K, scale);
}
// dw2[j] = sum_i relu(pre[i][j]) ds[i] (samples in ascending order: reproducible); db2 = sum_i ds[i]
__global__ void point_w2_grad_kernel(const float* __restrict__ pre, const float* __restrict__ ds, float* __restrict__ dw2,
                                     float* __restrict__ db2, int n, int U) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j > U) return;
  float acc = 0.f;
  if (j < U) {
    for (int i = 0; i < n; ++i) acc += fmaxf(pre[(size_t)i * U + j], 0.f) * ds[i];
    dw2[j] = acc;
  } else {
    for (int i = 0; i < n; ++i) acc += ds[i];
    db2[0] = acc;
  }
}
void launch_point_w2_grad(const float* pre, const float* ds, float* dw2, float* db2, int n, int U, hipStream_t s) {
  hipLaunchKernelGGL(point_w2_grad_kernel, dim3(cdiv(U + 1, 64)), dim3(64), 0, s, pre, ds, dw2, db2, n, U);
}

__global__ __launch_bounds__(256) void mean2_kernel(const float* __restrict__ a, const float* __restrict__ b, int n,
                                                    float* __restrict__ stats) {
  __shared__ float ra[4], rb[4];
  float sa = 0.f, sb = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    sa += a[i];
    sb += b[i];
  }
  sa = wave_sum(sa);
  sb = wave_sum(sb);
  if ((threadIdx.x & 63) == 0) {
    ra[threadIdx.x >> 6] = sa;
    rb[threadIdx.x >> 6] = sb;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    stats[0] = (ra[0] + ra[1] + ra[2] + ra[3]) / (float)n;
    stats[1] = (rb[0] + rb[1] + rb[2] + rb[3]) / (float)n;
  }
}
void launch_mean2(const float* a, const float* b, int n, float* stats, hipStream_t s) {
  if (n) hipLaunchKernelGGL(mean2_kernel, dim3(1), dim3(256), 0, s, a, b, n, stats);
}

}  // namespace lstur
