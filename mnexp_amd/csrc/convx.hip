// Conv1D('same', relu) x (id != 0) of the cnnatt news encoder (task/paper.py:132-155) as an fp32-GRADE contraction on the
// 16-bit matrix cores, with BOTH operands PRE-SPLIT into NP 16-bit parts ("split-operand conv", convx):
//
//   NP = 2  "f16x2"  two fp16 parts h | l of the power-of-two scaled value, THREE products hh + hl + lh per fp32-grade
//                    product (bf16_mfma.h: split2).  The default: half the matrix-core work of the bf16 triple and 4 bytes
//                    per staged element.  Scales (powers of two, exact): embeddings from max |word table| (computed when
//                    the table is bound / changed), weight from max |conv_w| (tracked by the pack launches), conv-output
//                    gradient from max |dC| (tracked by the attention backward); all live in a small device block `cs`.
//   NP = 3  "bf16x3" three bf16 parts h | m | l, SIX products hh + hm + mh + hl + lh + mm (split3); no scales.  Kept for the
//                    trainable word table (its maximum moves every step) and as the A/B partner (LSTUR_CONV_SCHEME=bf16x3).
//
// This file holds the whole family: the embedding gather that writes the split layout, the weight pack, the forward
// contraction and the kernel gradient.
//
//   * the gather (gatherx_kernel; Embedding + Dropout, paper.py:132-147) writes the parts of every embedding element straight
//     into the padded per-title layout "Ex": [title][T + k - 1][Dp / 8][part][8] 16-bit (Dp = D rounded up to 8; the pad
//     elements d >= D are zeros), i.e. the NP 16-byte pieces of an 8-element k-chunk lie side by side, the window of token
//     (n, t) is the contiguous run of k * Dp / 8 chunks that starts at padded row n * Tp + t, and the contraction index is
//     k' = tap * Dp + d;
//   * the kernel weight is split once per step into Wx [part][Kp][Np] (convx_pack_w: rows in k' order, zero rows at d >= D;
//     in the training step the dense Adam launch writes the image itself, misc.hip: image_scatter).
//   So the forward's k-loop carries no conversion arithmetic at all: 16-byte copies HBM/L2 -> registers -> LDS, LDS fragment
//   reads, MFMAs.
//
// Forward schedule: the B fragments of column tile j + 1 are read while the products of tile j run (two register sets,
// static indices), the column-tile count is a scalar (readfirstlane) so its tests are s_cbranch, and the tile loop exists
// once per live tile count (straight-line code, counted lgkmcnt).
// Forward geometry: 8 waves, 128 token rows x all F (<= 448) columns, one workgroup per CU; wave = 32 rows x up to 7 column
// tiles; the two waves of a SIMD take the two column groups of the same rows (7 + 6 tiles at F = 400: every SIMD carries 13).
// Row quantisation: tiles of 128 rows on 256 CUs come in rounds; tiles beyond the last whole round are cut along k into S
// slices (S = min(8, CUs / tiles left)) that run side by side and leave raw partial sums in slabs; convx_fwd_fix sums the
// slabs of a tile in slice order (bitwise reproducible) and applies the epilogue.
// Measured and rejected alternatives of rounds 1-2 (4-wave workgroups, 512-register waves, tap sharing, 16x16x32 tiles,
// other work splits of the kernel gradient) are recorded with their numbers in DESIGN.md section 4.
#include <type_traits>

#include "bf16_mfma.h"
#include "lstur_internal.h"

namespace lstur {

// k-depth of a forward stage.  16 for both schemes: a 32-deep stage for the fp16 pair (two stages of 128 x 32 + 32 x 448 images
// in two parts just fit the 160 KB of LDS; half the barriers and fragment-read restarts) was built and measured at the same
// time within the box-to-box spread (hot 136.0 against 138.7 us, cold 151.9 against 150.6 us, tools/bench_convx.py) at 256
// VGPRs + 4 spilled: the kernel is not bound by its barriers (the ablation says so too: -3 us without them).  The kernel
// keeps the stage depth as a parameter.
__host__ __device__ constexpr int fx_bk(int np) { return np == 2 ? 16 : 16; }

int convx_dp(int D) { return (D + 7) / 8 * 8; }
static int fx_kp(int D, int ksize, int np) { return cdiv(ksize * convx_dp(D), fx_bk(np)) * fx_bk(np); }
static int fx_np(int F) { return cdiv(F, 32) * 32; }
// 16-bit elements of Ex for n_titles titles (+ slack: the k tail of the last window and clamped loads stay inside)
size_t convx_e_elems(int n_titles, int T, int D, int ksize, int np) {
  return (size_t)n_titles * (T + ksize - 1) * convx_dp(D) * np + (size_t)np * 256;
}
size_t convx_w_elems(int D, int ksize, int F, int np) { return (size_t)np * fx_kp(D, ksize, np) * fx_np(F); }
void convx_w_dims(int D, int ksize, int F, int np, int* Kp, int* Np) {
  *Kp = fx_kp(D, ksize, np);
  *Np = fx_np(F);
}
bool convx_supported(int D, int ksize, int F) { return D % 4 == 0 && F >= 1 && F <= 448 && ksize >= 1 && ksize <= 8; }

template <int NP>
__device__ __forceinline__ f32x16 mfma_parts(u16x8 a, u16x8 b, f32x16 c) {
  if constexpr (NP == 2) return mfma32h(a, b, c);
  else return mfma32b(a, b, c);
}
// the products of one 32 x 32 x 16 tile step, small terms first
template <int NP>
__device__ __forceinline__ f32x16 tile_products(const u16x8 (&a)[NP], const u16x8 (&b)[NP], f32x16 c) {
  if constexpr (NP == 2) {
    c = mfma_parts<2>(a[1], b[0], c);       // l h
    c = mfma_parts<2>(a[0], b[1], c);       // h l
    c = mfma_parts<2>(a[0], b[0], c);       // h h
  } else {
    c = mfma_parts<3>(a[2], b[0], c);       // l h
    c = mfma_parts<3>(a[0], b[2], c);       // h l
    c = mfma_parts<3>(a[1], b[1], c);       // m m
    c = mfma_parts<3>(a[1], b[0], c);       // m h
    c = mfma_parts<3>(a[0], b[1], c);       // h m
    c = mfma_parts<3>(a[0], b[0], c);       // h h
  }
  return c;
}

// ---------------------------------------------------------------------------------------------
// Scale block (NP = 2): see lstur_internal.h for the word indices.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, size_t n, uint32_t* __restrict__ out) {
  uint32_t m = 0;
  const bool vec = ((uintptr_t)x & 15) == 0;
  const size_t n4 = vec ? n / 4 : 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = ((const float4*)x)[i];
    m = max(max(m, __float_as_uint(fabsf(v.x))), max(__float_as_uint(fabsf(v.y)), max(__float_as_uint(fabsf(v.z)), __float_as_uint(fabsf(v.w)))));
  }
  for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    m = max(m, __float_as_uint(fabsf(x[i])));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
// bits of max |x[i]| (non-negative floats order like their bit patterns) -> atomicMax into *out_bits (zero it first)
void launch_absmax(const float* x, size_t n, uint32_t* out_bits, hipStream_t s) {
  if (!n) return;
  const int blocks = (int)std::min<size_t>(cdivz(n, 1024), 1024);
  hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(256), 0, s, x, n, out_bits);
}
// the same over the rows a compacted list names (vidx / vcount NULL: rows [0, M)): rows outside the list are not read
__global__ __launch_bounds__(256) void absmax_rows_kernel(const float* __restrict__ x, int cols, const int32_t* __restrict__ vidx,
                                                          const int32_t* __restrict__ vcount, int M, uint32_t* __restrict__ out) {
  const size_t total = (size_t)(vcount ? vcount[0] : M) * cols;
  uint32_t m = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t r = i / cols, c = i - r * cols;
    m = max(m, __float_as_uint(fabsf(x[(size_t)(vidx ? vidx[r] : (int)r) * cols + c])));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
void launch_absmax_rows(const float* x, int rows, int cols, const int32_t* vidx, const int32_t* vcount, uint32_t* out_bits,
                        hipStream_t s) {
  if (rows <= 0 || cols <= 0) return;
  const int blocks = (int)std::min<size_t>(cdivz((size_t)rows * cols, 1024), 1024);
  hipLaunchKernelGGL(absmax_rows_kernel, dim3(blocks), dim3(256), 0, s, x, cols, vidx, vcount, rows, out_bits);
}
__global__ void escale_kernel(float* __restrict__ cs, float drop_scale) {
  const uint32_t* cu = (const uint32_t*)cs;
  // the gather multiplies by 1 / (1 - p) first: the largest staged magnitude is max |table| / (1 - p) -> [2^14, 2^15)
  const float mx = __uint_as_float(cu[kCsEmax]) * drop_scale;
  const float sE = pow2_scale(__float_as_uint(mx), 14);
  cs[kCsE] = sE;
  cs[kCsInvE] = 1.0f / sE;
  cs[kCsFwd] = 1.0f / (sE * cs[kCsW]);
}
// embedding scale from max |table| (NP = 2 contexts: when the word table is bound or changed)
void launch_convx_escale(const float* table, size_t count, float drop, float* cs, hipStream_t s) {
  (void)hipMemsetAsync((uint32_t*)cs + kCsEmax, 0, sizeof(uint32_t), s);
  launch_absmax(table, count, (uint32_t*)cs + kCsEmax, s);
  hipLaunchKernelGGL(escale_kernel, dim3(1), dim3(1), 0, s, cs, drop > 0.f ? 1.0f / (1.0f - drop) : 1.0f);
}
// a fresh block: every scale 1, every maximum 0
__global__ void cs_init_kernel(float* __restrict__ cs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= kCsWords) return;
  cs[i] = (i == kCsE || i == kCsW || i == kCsFwd || i == kCsInvE) ? 1.0f : 0.0f;
}
void launch_convx_scales_init(float* cs, hipStream_t s) {
  hipLaunchKernelGGL(cs_init_kernel, dim3(cdiv(kCsWords, 256)), dim3(256), 0, s, cs);
}

// ---------------------------------------------------------------------------------------------
// Gather: Embedding + Dropout (paper.py:132-147) into Ex.  One thread = one 8-element chunk: 32 bytes of a table row in,
// NP contiguous 16-byte pieces out; dropout mask and 1 / (1 - p) are applied in fp32 BEFORE the split (then the scale of
// the fp16 pair, exact).  A wave's 64 chunks cross LDS once so that every store instruction writes 1 KB of consecutive
// bytes (lane-strided granules reached HBM as partial lines).
// Rows that no window of a non-pad token covers are SKIPPED (neither read nor written): the conv kernels run over the
// non-pad tokens only (see compact_*), so with SKIP a padded row tp is needed iff some token t' in [t - pad_r, t + pad_l]
// (t = tp - pad_l) of the title is non-pad -- 30 % fewer bytes at config 2 (titles are zero-padded at the end, histories at
// the front).  The buffer must hold finite values from the start (the k tail of a window can touch a skipped row under a
// zero weight): the step zeroes it once.
// ---------------------------------------------------------------------------------------------
// LIVE: which rows are covered comes from the title's non-pad bit mask (bit t of live[title], T <= 32; the compaction kernel
// leaves it) instead of the neighbours' ids: behind a step's traffic those id loads come from HBM and the table load hangs on
// them -- 45 -> 35 us from cold caches (tools/probe_gather.hip, profiles/r03_probe_gather.txt).
template <int NP, bool DROP, bool SKIP, bool LIVE>
__global__ __launch_bounds__(256) void gatherx_kernel(const float4* __restrict__ table, const int32_t* __restrict__ ids,
                                                      const uint32_t* __restrict__ live_mask, uint4* __restrict__ out,
                                                      uint32_t total8, int T, int Tp, int D4, int C8, int pad_l, int pad_r,
                                                      uint32_t thresh, uint32_t key, float scale,
                                                      const float* __restrict__ cs) {
  __shared__ uint4 xbuf[4][64 * NP];
  // two chunks (four 16-byte table loads) in flight per thread: 54 registers = 8 waves per SIMD; four chunks cost three of
  // the eight waves and ran 2 us slower from cold caches, one or eight chunks 3 to 6 us slower
  constexpr int U = 2;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t stride = gridDim.x * blockDim.x;
  const float sE = NP == 2 ? cs[kCsE] : 1.0f;
  for (uint32_t b0 = blockIdx.x * blockDim.x; b0 < total8; b0 += U * stride) {     // block-uniform trip count
    const uint32_t i0 = b0 + threadIdx.x;
    uint32_t c8[U], tok[U];
    bool inr[U], real[U];
    int id[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t i = min(i0 + u * stride, total8 - 1u);
      inr[u] = i0 + u * stride < total8;
      const uint32_t row = i / (uint32_t)C8;
      c8[u] = i - row * (uint32_t)C8;
      const uint32_t n = row / (uint32_t)Tp;
      const int t = (int)(row - n * (uint32_t)Tp) - pad_l;
      real[u] = t >= 0 && t < T;
      tok[u] = n * (uint32_t)T + (uint32_t)min(max(t, 0), T - 1);
      id[u] = ids[tok[u]];
      if (SKIP && LIVE) {
        const int lo = max(t - pad_r, 0), hi = min(t + pad_l, T - 1);
        const uint32_t win = hi >= lo ? ((0xFFFFFFFFu >> (31 - (hi - lo))) << lo) : 0u;
        inr[u] = inr[u] && (live_mask[n] & win) != 0u;
      } else if (SKIP) {
        bool need = false;
        for (int tt = max(t - pad_r, 0); tt <= min(t + pad_l, T - 1); ++tt) need |= ids[n * (uint32_t)T + (uint32_t)tt] != 0;
        inr[u] = inr[u] && need;
      }
    }
    float4 va[U], vb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      va[u] = table[(size_t)id[u] * D4 + min((int)(2u * c8[u]), D4 - 1)];
      vb[u] = table[(size_t)id[u] * D4 + min((int)(2u * c8[u] + 1u), D4 - 1)];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned long long live = __ballot(inr[u]);            // which chunks of this wave are written
      if (live == 0ull) continue;                                   // (wave-uniform: no hashing / split for skipped rows)
      float4 w[2] = {va[u], vb[u]};
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const uint32_t c4 = 2u * c8[u] + hf;
        if (DROP) {
          const uint32_t e = (tok[u] * (uint32_t)D4 + c4) * 4u;
          bool k[4];
          keep_quad(key, e, thresh, k);
          w[hf].x = k[0] ? w[hf].x * scale : 0.f;
          w[hf].y = k[1] ? w[hf].y * scale : 0.f;
          w[hf].z = k[2] ? w[hf].z * scale : 0.f;
          w[hf].w = k[3] ? w[hf].w * scale : 0.f;
        }
        if (!real[u] || (int)c4 >= D4) w[hf] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (NP == 2) w[hf] = make_float4(w[hf].x * sE, w[hf].y * sE, w[hf].z * sE, w[hf].w * sE);
      }
      auto pk = [](u16x4 a, u16x4 b) {
        return make_uint4((uint32_t)a[0] | ((uint32_t)a[1] << 16), (uint32_t)a[2] | ((uint32_t)a[3] << 16),
                          (uint32_t)b[0] | ((uint32_t)b[1] << 16), (uint32_t)b[2] | ((uint32_t)b[3] << 16));
      };
      if constexpr (NP == 2) {
        u16x4 h0, l0, h1, l1;
        split2(w[0], h0, l0);
        split2(w[1], h1, l1);
        xbuf[wv][lane * 2 + 0] = pk(h0, h1);
        xbuf[wv][lane * 2 + 1] = pk(l0, l1);
      } else {
        u16x4 h0, m0, l0, h1, m1, l1;
        split3(w[0], h0, m0, l0);
        split3(w[1], h1, m1, l1);
        xbuf[wv][lane * 3 + 0] = pk(h0, h1);
        xbuf[wv][lane * 3 + 1] = pk(m0, m1);
        xbuf[wv][lane * 3 + 2] = pk(l0, l1);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      uint4* o = out + (size_t)(b0 + wv * 64 + u * stride) * NP;    // first piece of the wave's 64 chunks
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        const int pc = q * 64 + lane;                               // piece pc belongs to chunk (= lane) pc / NP
        const uint4 v = xbuf[wv][pc];
        if ((live >> (pc / NP)) & 1ull) {
          // streamed once, read much later (by the conv kernels of the next step): non-temporal, no cache allocation
          typedef unsigned int u32x4nt __attribute__((ext_vector_type(4)));
          __builtin_nontemporal_store(*(const u32x4nt*)&v, (u32x4nt*)(o + pc));
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

void launch_gatherx(const float* table, const int32_t* ids, unsigned short* ex, int n_titles, int T, int D, int pad_l,
                    int pad_r, float drop, uint32_t key, bool skip_unread, int np, const float* cs, hipStream_t s,
                    int max_blocks, const uint32_t* live) {
  const int Tp = T + pad_l + pad_r, D4 = D / 4, C8 = (D + 7) / 8;
  const size_t total8 = (size_t)n_titles * Tp * C8;
  if (total8 == 0) return;
  if (T > 32 || !skip_unread) live = nullptr;   // (one 32-bit word per title)
  // (max_blocks: a background launch beside latency-bound kernels takes a fraction of the chip on purpose)
  const int blocks = (int)std::min<size_t>(cdivz(total8, 256 * 2), max_blocks > 0 ? max_blocks : 256 * 32);
  const uint32_t th = drop > 0.f ? dropout_thresh(drop) : 0u;
  const float sc = drop > 0.f ? 1.0f / (1.0f - drop) : 1.0f;
  auto go = [&](auto kern) {
    last_launch();
    launch_k(kern, dim3(blocks), dim3(256), 0, s, (const float4*)table, ids, live, (uint4*)ex, (uint32_t)total8, T, Tp, D4, C8,
             pad_l, pad_r, th, key, sc, cs);
  };
  auto pick = [&](auto np_c, auto drop_c) {
    constexpr int NP = decltype(np_c)::value;
    constexpr bool DR = decltype(drop_c)::value;
    if (!skip_unread) go(gatherx_kernel<NP, DR, false, false>);
    else if (live) go(gatherx_kernel<NP, DR, true, true>);
    else go(gatherx_kernel<NP, DR, true, false>);
  };
  if (np == 2) {
    if (drop > 0.f) pick(std::integral_constant<int, 2>{}, std::true_type{});
    else pick(std::integral_constant<int, 2>{}, std::false_type{});
  } else {
    if (drop > 0.f) pick(std::integral_constant<int, 3>{}, std::true_type{});
    else pick(std::integral_constant<int, 3>{}, std::false_type{});
  }
}

// ---------------------------------------------------------------------------------------------
// Weight pack: W [ksize * D, F] fp32 -> NP 16-bit parts [NP][Kp][Np] in k' = tap * Dp + d order, zero padded.
// NP = 2: the scale comes from the maximum the PREVIOUS pack saw (slot seq % 3 of the rotating maxima) plus `grow`, a bound
// on how far one optimizer step can have moved any weight since (Keras Adam: |dp| <= lr_t |m| / sqrt(v) <= lr_t (1 - b1) /
// sqrt(1 - b2) / (1 - b1 / sqrt(b2)), capi: adam_growth_bound; 0 for a fresh pack whose slot holds the maximum of W as it
// is): (max + grow) * sW in [2^13, 2^14), so no element can leave the fp16 range (the clamp is a second fence, never taken).
// This pack leaves its own maximum in slot (seq + 1) % 3 and clears slot (seq + 2) % 3 -- one pass, no grid-wide dependency,
// the same rule in the stand-alone kernel below and in the dense Adam launch that writes the image in the training step
// (misc.hip: image_scatter).
// ---------------------------------------------------------------------------------------------
// the maximum a rotating slot holds (every lane of the wave gets it; EXEC must be full)
__device__ __forceinline__ uint32_t wmax_read(const uint32_t* cu, int slot, int lane) {
  uint32_t m = cu[wmax_word(slot, lane & (kWmaxLanes - 1))];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
  return m;
}
// a workgroup of 256 threads leaves its maximum in slot `add` and (first workgroup) clears slot `clear`
__device__ __forceinline__ void wmax_leave(uint32_t* cu, uint32_t m, int add, int clear) {
  __shared__ uint32_t wm[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
    if (m) atomicMax(cu + wmax_word(add, blockIdx.x & (kWmaxLanes - 1)), m);
  }
  if (blockIdx.x == 0 && threadIdx.x < kWmaxLanes) cu[wmax_word(clear, threadIdx.x)] = 0u;
}
template <int NP>
__global__ __launch_bounds__(256) void convx_pack_w_kernel(const float* __restrict__ W, int D, int Dp, int ksize, int F, int Kp, int Np,
                                    unsigned short* __restrict__ Wx, float* __restrict__ cs, int seq, float grow) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool in = i < (size_t)Kp * Np;
  const int n = in ? (int)(i % Np) : 0, kp = in ? (int)(i / Np) : 0;
  const int tap = kp / Dp, d = kp - tap * Dp;
  const float v = (in && tap < ksize && d < D && n < F) ? W[((size_t)tap * D + d) * F + n] : 0.f;
  if constexpr (NP == 2) {
    uint32_t* cu = (uint32_t*)cs;
    const float sW = pow2_scale(__float_as_uint(__uint_as_float(wmax_read(cu, seq % 3, threadIdx.x & 63)) + grow), 13);
    u16x4 h, l;
    const float vs = fminf(fmaxf(v * sW, -65504.f), 65504.f);
    split2(make_float4(vs, 0.f, 0.f, 0.f), h, l);
    if (in) {
      Wx[i] = h[0];
      Wx[(size_t)Kp * Np + i] = l[0];
    }
    wmax_leave(cu, __float_as_uint(fabsf(v)), (seq + 1) % 3, (seq + 2) % 3);
    if (i == 0) {
      cs[kCsW] = sW;
      cs[kCsFwd] = 1.0f / (cs[kCsE] * sW);
    }
  } else {
    if (!in) return;
    u16x4 h, m, l;
    split3(make_float4(v, 0.f, 0.f, 0.f), h, m, l);
    Wx[i] = h[0];
    Wx[(size_t)Kp * Np + i] = m[0];
    Wx[(size_t)2 * Kp * Np + i] = l[0];
  }
}
// fresh: the maximum is taken from W itself first (a first pack, or the caller changed the weights)
void launch_convx_pack_w(const float* W, int D, int ksize, int F, unsigned short* Wx, int np, float* cs, int seq, bool fresh,
                         float grow, hipStream_t s) {
  const int Kp = fx_kp(D, ksize, np), Np = fx_np(F);
  const size_t total = (size_t)Kp * Np;
  const dim3 grid((unsigned)cdivz(total, 256));
  if (np == 2) {
    if (fresh) {
      (void)hipMemsetAsync((uint32_t*)cs + kCsWmax, 0, (size_t)(kCsWords - kCsWmax) * sizeof(uint32_t), s);
      launch_absmax(W, (size_t)ksize * D * F, (uint32_t*)cs + wmax_word(seq % 3, 0), s);
    }
    hipLaunchKernelGGL(convx_pack_w_kernel<2>, grid, dim3(256), 0, s, W, D, convx_dp(D), ksize, F, Kp, Np, Wx, cs, seq,
                       fresh ? 0.f : grow);
  } else {
    hipLaunchKernelGGL(convx_pack_w_kernel<3>, grid, dim3(256), 0, s, W, D, convx_dp(D), ksize, F, Kp, Np, Wx, cs, seq, 0.f);
  }
}

__host__ __device__ constexpr int fx_rows(int RWM) { return 32 * RWM; }   // RWM = row waves x row tiles per wave
__host__ __device__ constexpr int fx_sb(int CW, int NTW) { return tr_stride(CW * NTW * 32); }
// one stage: NP A images [BM][BK] (rows of BK / 8 sixteen-byte pieces, swizzled so that the 16 lanes of a ds_read_b128
// group hit 16 different bank slots) + NP B images [BK][SB]
__host__ __device__ constexpr int fx_stage(int NP, int RW, int CW, int NTW) {
  return NP * fx_rows(RW) * fx_bk(NP) + NP * fx_bk(NP) * fx_sb(CW, NTW) + 64 * 8;   // + a 16-byte trash slot per lane
}
// 16-byte piece p of row r inside an A image row (BK = 16: 2 pieces, 32-byte rows; BK = 32: 4 pieces, 64-byte rows)
__host__ __device__ constexpr int fx_apiece(int BK, int r, int p) { return BK == 16 ? (p ^ ((r >> 3) & 1)) : (p ^ ((r >> 2) & 3)); }
__host__ __device__ constexpr size_t fx_lds(int NP, int RW, int CW, int NTW) {
  return 2 * fx_stage(NP, RW, CW, NTW) * sizeof(unsigned short) + 2 * fx_rows(RW) * sizeof(int32_t);
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

// OPT != 0 (only with -DLSTUR_ABLATE, the bench-only library): timing-only ablations with WRONG results
template <int NP, int RW, int CW, int NTW, int OPT = 0>
__global__ __launch_bounds__(RW* CW * 64, 2) void convx_fwd_kernel(
    const unsigned short* __restrict__ Ex, uint32_t e_bytes, const unsigned short* __restrict__ Wx,
    const float* __restrict__ bias, const int32_t* __restrict__ ids, float* __restrict__ C,
    const int32_t* __restrict__ vidx, const int32_t* __restrict__ vcount, int M, int T, int Tp, int Dp8, int F, int Kp,
    int Np, int ncu, float* __restrict__ slabs, const float* __restrict__ cs) {
  constexpr int BK = fx_bk(NP), KP8 = BK / 8, KSUB = BK / 16;
  constexpr int BM = fx_rows(RW), NTH = RW * CW * 64, SB = fx_sb(CW, NTW), BCOLS = CW * NTW * 32;
  constexpr int A_IMG = BM * BK, B_IMG = BK * SB, STAGE = fx_stage(NP, RW, CW, NTW);
  constexpr int A_PCS = BM * KP8 * NP, A_PER = (A_PCS + NTH - 1) / NTH;     // 16-byte pieces (row, k-eighth, part) of a step
  constexpr int B_CH = NP * BK * (BCOLS / 8), B_PER = (B_CH + NTH - 1) / NTH;
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
  const int NT = Np / 32;                                  // column tiles of 32
  // waves w and w + NW/2 share a SIMD: give them the SAME rows and the two column groups, so that with an odd number of
  // column tiles (F = 400: 7 + 6) every SIMD carries the same 13 tiles per step instead of 14 on two SIMDs and 12 on two
  const int wm = wv % RW, wn = wv / RW;
  const int nt_lo = wn * NTW;
  const int my_nt = max(0, min(NTW, NT - nt_lo));          // scalar (wave-uniform by construction)

  // A loader role: piece idx = row * KP8 NP + q, q = k-eighth * NP + part: the pieces of a row and step are 2 BK NP contiguous bytes
  uint32_t a_off[A_PER];
  int a_lds[A_PER];
#pragma unroll
  for (int p = 0; p < A_PER; ++p) {
    const int idx = tid + p * NTH, arow = idx / (KP8 * NP), q = idx - arow * (KP8 * NP), seg = q / NP, part = q - seg * NP;
    const bool use = A_PCS % NTH == 0 || idx < A_PCS;
    const int i = min(m0 + min(arow, BM - 1), count - 1);
    const int m = vidx ? vidx[i] : i;
    const int title = m / T, t = m - title * T;
    a_off[p] = use ? ((uint32_t)(title * Tp + t) * (uint32_t)Dp8 * (uint32_t)NP + (uint32_t)q) * 16u : kBufOob;
    a_lds[p] = use ? part * A_IMG + arow * BK + (fx_apiece(BK, arow, seg) << 3) : trash;
    if (use && q == 0) {
      rowm[arow] = (m0 + arow < count) ? m : -1;
      rowkeep[arow] = ids[m] != 0 ? 1.f : 0.f;
    }
  }
  // B loader role: pieces of 16 bytes (8 columns) of the NP x [16][BCOLS] window of the weight parts
  uint32_t b_goff[B_PER];
  int b_loff[B_PER];
#pragma unroll
  for (int i = 0; i < B_PER; ++i) {
    const int idx = tid + i * NTH;
    constexpr int C8 = BCOLS / 8;
    const int img = idx / (BK * C8), rem = idx - img * (BK * C8), krow = rem / C8, ch = rem - krow * C8;
    const bool use = idx < B_CH && ch * 8 < Np;
    b_goff[i] = use ? (((uint32_t)img * (uint32_t)Kp + (uint32_t)krow) * (uint32_t)Np + (uint32_t)(ch * 8)) * 2u : kBufOob;
    b_loff[i] = use ? NP * A_IMG + img * B_IMG + krow * SB + ch * 8 : trash;
  }
  const __amdgpu_buffer_rsrc_t rsE = make_rsrc(Ex, e_bytes);
  const __amdgpu_buffer_rsrc_t rsW = make_rsrc(Wx, (uint32_t)((size_t)NP * Kp * Np * 2));

  f32x16 acc[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  float4 ra[A_PER];
  float4 rb[B_PER];
  int it0_ = 0;
  auto load = [&](int k0) {
    if ((OPT & 4) && k0 >= (it0_ + 2) * BK) return;     // timing-only ablation: no global loads inside the loop
#pragma unroll
    for (int p = 0; p < A_PER; ++p) ra[p] = buf_load4(rsE, a_off[p], (uint32_t)k0 * (2u * NP));
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

  const int nit_all = Kp / BK;
  const int it0 = sliced ? ks * nit_all / tp.S : 0, it1 = sliced ? (ks + 1) * nit_all / tp.S : nit_all;
  it0_ = it0;
  const int nit = it1 - it0;                       // >= 1 for nit_all >= S (S <= 8; guarded by the launcher)
  load(it0 * BK);
  store(0);
  if (nit > 1) load((it0 + 1) * BK);
  __syncthreads();

  // fragment addresses (elements): A = row-major [row][BK] with the 16-byte pieces swizzled; B by transposed reads
  const int arow = 32 * wm + (lane & 31);
  int a_frag[KSUB];
#pragma unroll
  for (int ss = 0; ss < KSUB; ++ss) a_frag[ss] = arow * BK + (fx_apiece(BK, arow, 2 * ss + (lane >> 5)) << 3);
  // The k-loop exists once per number NA of live column tiles of the wave (a compile-time constant inside): straight-line
  // code with no test in it, so the compiler counts its LDS reads exactly (lgkmcnt(N), not (0)) and keeps the reads of
  // tile j + 1 in flight under the products of tile j.
  auto kloop = [&](auto na_tag) {
    constexpr int NA = decltype(na_tag)::value;
    for (int it = 0; it < nit; ++it) {
      if (it + 1 < nit) {
        if (!(OPT & 8) || it == 0) store((it & 1) ^ 1);
        if (it + 2 < nit) load((it0 + it + 2) * BK);
      }
      if constexpr (NA > 0) {
        const unsigned short* st = smem + (it & 1) * STAGE;
        const unsigned short* bp = st + NP * A_IMG;
        // the stage's KSUB 16-deep sub-steps x NA column tiles as one straight-line sequence: the B fragments of item
        // i + 1 are read while the products of item i run
        u16x8 af[KSUB][NP];
#pragma unroll
        for (int ss = 0; ss < KSUB; ++ss)
#pragma unroll
          for (int q = 0; q < NP; ++q) af[ss][q] = *(const u16x8*)(st + q * A_IMG + a_frag[ss]);
        auto frags = [&](int i, u16x8 (&b)[NP]) {
          const int ss = i / NA, j = i - ss * NA;
          const int x0 = (wn * NTW + j) * 32;                // column inside this workgroup's window
#pragma unroll
          for (int q = 0; q < NP; ++q) b[q] = tr_frag(bp + q * B_IMG, SB, 16 * ss, x0, lane);
        };
        constexpr int NI = KSUB * NA;
        u16x8 bA[NP], bB[NP];
        if (OPT & 32) {                      // timing-only ablation: no B fragment reads (operands = whatever A holds)
#pragma unroll
          for (int q = 0; q < NP; ++q) bA[q] = af[0][0];
#pragma unroll
          for (int i = 0; i < NI; ++i) {
            asm volatile("" : "+v"(bA[0]), "+v"(bA[1]));
            acc[i % NA] = tile_products<NP>(af[i / NA], bA, acc[i % NA]);
          }
        } else {
          frags(0, bA);
#pragma unroll
          for (int i = 0; i < NI; i += 2) {
            if (i + 1 < NI) frags(i + 1, bB);
            acc[i % NA] = tile_products<NP>(af[i / NA], bA, acc[i % NA]);
            if (i + 2 < NI) frags(i + 2, bA);
            if (i + 1 < NI) acc[(i + 1) % NA] = tile_products<NP>(af[(i + 1) / NA], bB, acc[(i + 1) % NA]);
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
    // raw (still scaled) partial sums of this k slice: slab [(tile - full) * S + ks][BM][BCOLS]
    float* slab = slabs + ((size_t)(tile - tp.full) * tp.S + ks) * (size_t)(BM * BCOLS);
    const __amdgpu_buffer_rsrc_t rsS = make_rsrc(slab, (uint32_t)(BM * BCOLS * 4));
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const int col = (wn * NTW + j) * 32 + c32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lr = 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * hh;
        buf_store1(acc[j][r], rsS, j < my_nt ? (uint32_t)(lr * BCOLS + col) * 4u : kBufOob, 0);
      }
    }
    return;
  }
  // epilogue: un-scale (NP = 2; exact), + bias, ReLU, x (id != 0); rows scattered to the token's place, branch-free
  // through a buffer descriptor
  const float unscale = NP == 2 ? cs[kCsFwd] : 1.0f;
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
    for (int r = 0; r < 16; ++r) {
      const float a = NP == 2 ? acc[j][r] * unscale : acc[j][r];
      buf_store1(fmaxf(a + bv, 0.f) * okeep[r], rsC,
                 (cok && orow[r] >= 0) ? ((uint32_t)orow[r] * (uint32_t)F + (uint32_t)col) * 4u : kBufOob, 0);
    }
  }
}

// Epilogue of the k-sliced tiles: out[token] = relu(bias + unscale * sum over slices, in slice order) x (id != 0).
// A few workgroup columns walk the sliced tiles; one thread = one column quad of one row, all slice loads in flight at once
__global__ __launch_bounds__(256) void convx_fwd_fix_kernel(const float* __restrict__ slabs, const float* __restrict__ bias,
                                                            const int32_t* __restrict__ ids, float* __restrict__ C,
                                                            const int32_t* __restrict__ vidx,
                                                            const int32_t* __restrict__ vcount, int M, int F, int ncu,
                                                            const float* __restrict__ cs) {
  constexpr int BM = 128, BCOLS = 448;
  const int count = vcount ? vcount[0] : M;
  const TailPlan tp = tail_plan(count, BM, ncu);
  if (tp.S <= 1) return;
  const float unscale = cs ? cs[kCsFwd] : 1.0f;
  // (the number of sliced tiles is a device-side value: a grid sized for the most there can be, ncu / 2, spent ~5 us
  //  dispatching 8 000 workgroups that returned at once)
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
      if (col + e < F) C[(size_t)m * F + col + e] = fmaxf(o[e] * unscale + bias[col + e], 0.f) * keep;
  }
}

size_t convx_fwd_scratch_bytes(int ncu) { return (size_t)ncu * 128 * 448 * sizeof(float); }
static int convx_ncu() {
  static int ncu = 0;
#ifdef LSTUR_ABLATE
  if (const char* e = getenv("LSTUR_PROBE_NCU")) return atoi(e);   // (timing probes: kernels planned for a chip that shares CUs)
#endif
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t prop;
    ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
              ? prop.multiProcessorCount : 256;
  }
  return ncu;
}

// scratch: convx_fwd_scratch_bytes(CUs) for the k-sliced tail tiles (NULL: no tail slicing).
// variant (bench-only library, -DLSTUR_ABLATE): 10 .. 15 timing-only ablations with WRONG results (tools/bench_convx.py):
// 10 no global loads in the loop, 11 + no LDS stores, 12 + no barrier, 13 + no B fragment reads, 14 only "no barrier",
// 15 only "no B reads".  The product library knows variant 0 only.
void launch_convx_fwd(const unsigned short* ex, const unsigned short* Wx, const float* bias, const int32_t* ids, float* out,
                      int n_titles, int T, int D, int ksize, int F, const int32_t* vidx, const int32_t* vcount, int np,
                      const float* cs, void* scratch, hipStream_t s, int variant) {
  const int M = n_titles * T;
  if (M == 0) return;
  const int Tp = T + ksize - 1, Dp = convx_dp(D), Np = fx_np(F), Kp = fx_kp(D, ksize, np);
  const uint32_t e_bytes = (uint32_t)std::min(convx_e_elems(n_titles, T, D, ksize, np) * 2, (size_t)0xFFFFFFFFu);
  const int tiles = cdiv(M, 128);
  // tail slicing needs at least one k-step per slice; without scratch every tile is a plain tile (ncu = "infinite")
  const int ncu = (scratch && Kp / fx_bk(np) >= 8) ? std::min(convx_ncu(), 256) : 0;   // scratch holds <= 256 slabs; 0: no slicing
  // grid: every tile as a plain tile, or whole rounds + at most ncu slices (the device decides from the token count)
  const int grid = tiles + ncu;
  if (ncu <= 0) last_launch();
  auto go = [&](auto kern, size_t lds) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    launch_k(kern, dim3(grid), dim3(512), lds, s, ex, e_bytes, Wx, bias, ids, out, vidx, vcount, M, T, Tp, Dp / 8, F, Kp, Np, ncu,
             (float*)scratch, cs);
  };
  (void)variant;
#ifdef LSTUR_ABLATE
  if (const char* e = getenv("LSTUR_CONVX_FWD_VARIANT")) variant = atoi(e);   // (bench-only library: tools/bench_convx.py --ablate)
  if (variant >= 10 && variant <= 15 && np == 2) {
    switch (variant) {
      case 10: go(convx_fwd_kernel<2, 4, 2, 7, 4>, fx_lds(2, 4, 2, 7)); break;
      case 11: go(convx_fwd_kernel<2, 4, 2, 7, 12>, fx_lds(2, 4, 2, 7)); break;
      case 12: go(convx_fwd_kernel<2, 4, 2, 7, 28>, fx_lds(2, 4, 2, 7)); break;
      case 13: go(convx_fwd_kernel<2, 4, 2, 7, 60>, fx_lds(2, 4, 2, 7)); break;
      case 14: go(convx_fwd_kernel<2, 4, 2, 7, 16>, fx_lds(2, 4, 2, 7)); break;
      default: go(convx_fwd_kernel<2, 4, 2, 7, 32>, fx_lds(2, 4, 2, 7)); break;
    }
  } else
#endif
  if (np == 2) go(convx_fwd_kernel<2, 4, 2, 7, 0>, fx_lds(2, 4, 2, 7));
  else go(convx_fwd_kernel<3, 4, 2, 7, 0>, fx_lds(3, 4, 2, 7));
  if (ncu > 0) {
    last_launch();
    launch_k(convx_fwd_fix_kernel, dim3(8, 64), dim3(256), 0, s, (const float*)scratch, bias, ids, out, vidx, vcount, M, F,
             ncu, np == 2 ? cs : (const float*)nullptr);
  }
}


// ---------------------------------------------------------------------------------------------
// Kernel gradient dW[k', F] = im2col(E)^T . dC over the (compacted) tokens: the A operand (the tokens' windows) comes
// PRE-SPLIT from Ex -- 16-byte copies, no conversion arithmetic; the conv-output gradient dC (fp32, written by the
// attention backward) is split on its way into LDS (NP = 2: scaled by the power of two that puts max |dC| -- tracked by
// the attention backward in cs[kCsGmax + gslot] -- at 2^14).  The contraction index is the token and both operands lie
// token-major in HBM, so both LDS tiles are row-major [token][x] images read with transposed LDS reads.
// Workgroup = 8 waves, tile 256 (k') x 224 (F): wave w owns row tile w and up to 7 column tiles; 16 tokens per step.
// Waves whose row tile lies past k' = ksize * Dp and column tiles past F issue NO MFMAs.
// The k-loop exists once per live column-tile count (compile-time): straight-line code, counted LDS waits.
// Tokens are cut into slabs (one per workgroup row) reduced in fixed order by the step's multi_reduce: one job per tap
// (rows d < D of tap j land on conv_w rows j * D + d).
// ---------------------------------------------------------------------------------------------
constexpr int GX_TM = 8, GX_TN = 7, GX_BK = 16, GX_THREADS = GX_TM * 64;
constexpr int GX_BMR = GX_TM * 32, GX_BNC = GX_TN * 32;                       // 256 x 224
constexpr int GX_SA = tr_stride(GX_BMR), GX_SB = tr_stride(GX_BNC);          // 288 / 224 elements
// The part images of A lie a few 16-byte bank groups apart (mod 256 bytes): the loader's 16-byte stores go out in
// w = chunk * NP + part order, so eight consecutive lanes then cover eight different bank groups (with images a multiple of
// 256 bytes apart the parts of a chunk hit the same banks: an NP-way conflict on every A store)
__host__ __device__ constexpr int gx_a_img(int NP) { return GX_BK * GX_SA + (NP == 2 ? 32 : 24); }
constexpr int GX_B_IMG = GX_BK * GX_SB;
static_assert((GX_BK * GX_SA * 2) % 256 == 0, "part skew assumes image sizes of whole bank rows");
__host__ __device__ constexpr int gx_stage(int NP) { return NP * gx_a_img(NP) + NP * GX_B_IMG + 64 * 4; }   // + an 8-byte trash slot per lane
__host__ __device__ constexpr size_t gx_lds(int NP) { return 2 * gx_stage(NP) * sizeof(unsigned short); }
constexpr int GX_BPC = (GX_BK * (GX_BNC / 4) + GX_THREADS - 1) / GX_THREADS;  // 2 float4 per thread and step

static int gx_mblocks(int Kq) { return cdiv(cdiv(Kq, 32), GX_TM); }
static int gx_nblocks(int F) { return cdiv(cdiv(F, 32), GX_TN); }
// slabs: about one workgroup per CU (or whole multiples), at least 512 tokens each
static int gx_slabs(int M, int Kq, int F) {
  const int per = gx_mblocks(Kq) * gx_nblocks(F);
  // ONE round whatever the token count: a workgroup keeps its tile in registers over as many tokens as its slab has, and
  // every further round of slabs is 1.46 MB more per tile to write and to reduce (B = 512: eight rounds of 256 slabs cost
  // the kernel 8 % and the step's reduction 60 us against one round of 32 slabs, 4.10 -> 3.99 ms per step)
  const int rounds = 1;
  int want = rounds * 256 / per / 8 * 8;                           // whole multiples of 8 slabs: the same share for every XCD
#ifdef LSTUR_ABLATE
  if (const char* e = getenv("LSTUR_PROBE_NCU")) want = rounds * atoi(e) / per;
#endif
  return std::max(1, std::min(std::max(8, want), std::max(1, M / 512)));
}
// (twice the slab count of one launch + one: a step may give the candidate titles' tokens a launch of their own and reduce
//  its slabs into one ahead of time)
size_t convx_wgrad_scratch_bytes(int n_titles, int T, int D, int ksize, int F) {
  return (size_t)(3 * gx_slabs(n_titles * T, ksize * convx_dp(D), F) + 1) * ksize * convx_dp(D) * F * sizeof(float);
}
int convx_wgrad_slabs(int plan_titles, int T, int D, int ksize, int F) {
  return gx_slabs(plan_titles * T, ksize * convx_dp(D), F);
}
size_t convx_wgrad_slab_floats(int D, int ksize, int F) { return (size_t)ksize * convx_dp(D) * F; }

// OPT != 0 (bench-only library): timing-only ablations (WRONG results).  SIDE: the same kernel under another name -- the
// launches over a title sub-range that a step runs beside the recurrence show up as their own line in rocprof summaries
// and PMC passes
template <int NP, int OPT = 0, bool SIDE = false>
__global__ __launch_bounds__(GX_THREADS, 2) void convx_wgrad_kernel(
    const unsigned short* __restrict__ Ex, uint32_t e_bytes, const float* __restrict__ G, uint32_t g_bytes,
    float* __restrict__ slabs, const int32_t* __restrict__ vcount, int M, int F, int Kq, int mblocks, int nblocks, int nslabs,
    const int32_t* __restrict__ offA, const int32_t* __restrict__ offB, int range_sel, int slab_base,
    const float* __restrict__ cs, const float* __restrict__ gparts, int g_n) {
  constexpr int A_IMG = gx_a_img(NP), STAGE = gx_stage(NP), APC = NP;
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // tokens [lo, count) of the compacted list: all of it, or a title range between the boundaries the compaction counted
  // (vcount[0] total, [1] history | candidates).  range_sel: 0 all, 1 history, 2 candidates
  const int count = vcount ? vcount[range_sel == 1 ? 1 : 0] : M;
  const int lo = (vcount && range_sel == 2) ? vcount[1] : 0;
  // block -> (slab, m-block, n-block); XCD-aware: workgroup ids go round-robin over the 8 XCDs, so the blocks of one slab
  // get ids 8 apart: they run on ONE XCD at about the same time and share the slab's tokens through its L2
  const int per = mblocks * nblocks;
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int slab = (seq / per) * 8 + xcd, rem = seq - (seq / per) * per, mb = rem / nblocks, nb = rem - mb * nblocks;
  if (slab >= nslabs) return;
  const int per_slab = (count - lo + nslabs - 1) / nslabs, tpad = (per_slab + GX_BK - 1) / GX_BK * GX_BK;
  const int t_begin = lo + slab * tpad, t_end = min(count, t_begin + tpad);
  const int kd0 = mb * GX_BMR, f0 = nb * GX_BNC;
  const __amdgpu_buffer_rsrc_t rsE = make_rsrc(Ex, e_bytes), rsG = make_rsrc(G, g_bytes);
  // live work of this wave: its row tile must start inside k' < Kq; column tiles inside F
  const bool row_live = kd0 + wv * 32 < Kq;
  const int na = row_live ? max(0, min(GX_TN, (F - f0 + 31) / 32)) : 0;
  // NP = 2: power-of-two scale of dC (exact to undo) from the maxima the attention backward's workgroups left, one per
  // title of this launch's range: every workgroup takes the maximum of the g_n entries itself (a few loads per thread, no
  // atomics, no zeroing: same-address atomicMax from 14 000 waves cost the attention backward 100 us)
  float sG = 1.0f;
  if constexpr (NP == 2) {
    float gm = 0.f;
    for (int i = tid; i < g_n; i += GX_THREADS) gm = fmaxf(gm, gparts[i]);
    gm = wave_max(gm);
    float* gred = (float*)smem;
    if (lane == 0) gred[wv] = gm;
    __syncthreads();
    gm = gred[0];
#pragma unroll
    for (int i = 1; i < GX_TM; ++i) gm = fmaxf(gm, gred[i]);
    __syncthreads();                                   // (smem is the stage buffer from here on)
    sG = pow2_scale(__float_as_uint(gm), 14);
  }

  f32x16 acc[GX_TN];
#pragma unroll
  for (int j = 0; j < GX_TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // loader roles: a thread serves ONE token row of the step (row = tid / 32), so a step costs it one index load per operand
  // (offA / offB of its token).  A: pieces w = c + 32 p (p < NP) of the 32 NP of its row (w = chunk * NP + part: 512 NP
  // contiguous bytes of the token's window in Ex); B: float4 slots c + 32 p (p < 2) -- 64 slots for the 56 of a 224-column
  // tile; the other 8 load through the out-of-range offset and store to a trash slot
  static_assert(GX_BPC == 2 && GX_THREADS == 32 * GX_BK, "row-per-32-threads loader grid");
  const int l_row = tid >> 5, l_c = tid & 31;
  int a_lds[APC];
#pragma unroll
  for (int p = 0; p < APC; ++p) {
    const int w = l_c + 32 * p, chunk = w / NP, part = w - chunk * NP;
    a_lds[p] = part * A_IMG + l_row * GX_SA + chunk * 8;
  }
  const uint32_t a_w16 = (uint32_t)l_c * 16u + (uint32_t)kd0 * (2u * NP);   // byte offset inside the window
  int b_lds[GX_BPC];
#pragma unroll
  for (int p = 0; p < GX_BPC; ++p)
    b_lds[p] = (l_c + 32 * p) * 4 < GX_BNC ? NP * A_IMG + l_row * GX_SB + (l_c + 32 * p) * 4 : STAGE - 64 * 4 + (tid & 63) * 4;
  // Row offsets of a step's pieces come in two halves so that no wait sits between the prefetch loads: `fetch_rows`
  // only issues the index loads (clamped, branch-free), `make_offsets` turns them into byte offsets one step later.
  // A token past the slab's end contributes 0 through the out-of-range offset of G.
  uint32_t a_off, b_off[GX_BPC];
  int a_raw, b_raw;
  auto fetch_rows = [&](int t0) {
    if ((OPT & 1) && t0 > t_begin) return;          // ablation: the row offsets of the first step for every step
    a_raw = offA[min(t0 + l_row, count - 1)];
    b_raw = offB[min(t0 + l_row, count - 1)];
  };
  auto make_offsets = [&](int t0) {
    a_off = (uint32_t)a_raw * 16u + a_w16;                                               // offA: 16-byte pieces
#pragma unroll
    for (int p = 0; p < GX_BPC; ++p) {
      const int col = f0 + (l_c + 32 * p) * 4;
      b_off[p] = (t0 + l_row < t_end && col < F && col < f0 + GX_BNC) ? ((uint32_t)b_raw + (uint32_t)col) * 4u : kBufOob;   // offB: floats
    }
  };
  float4 ra[APC], rb[GX_BPC];
  int loads_done = 0;
  auto load = [&]() {
    if ((OPT & 4) && loads_done >= 2) return;       // ablation: no global loads inside the loop
    ++loads_done;
#pragma unroll
    for (int p = 0; p < APC; ++p) ra[p] = buf_load4(rsE, a_off, (uint32_t)(p * 512));
#pragma unroll
    for (int p = 0; p < GX_BPC; ++p) rb[p] = buf_load4(rsG, b_off[p], 0);
  };
  auto store = [&](int buf) {
    unsigned short* st = smem + buf * STAGE;
#pragma unroll
    for (int p = 0; p < APC; ++p) *(float4*)(st + a_lds[p]) = ra[p];
#pragma unroll
    for (int p = 0; p < GX_BPC; ++p) {
      const bool in_tile = (l_c + 32 * p) * 4 < GX_BNC;
      unsigned short* d = st + b_lds[p];
      if constexpr (NP == 2) {
        u16x4 h, l;
        split2(make_float4(rb[p].x * sG, rb[p].y * sG, rb[p].z * sG, rb[p].w * sG), h, l);
        *(u16x4*)d = h;
        *(u16x4*)(d + (in_tile ? GX_B_IMG : 0)) = l;
      } else {
        u16x4 h, m, l;
        split3(rb[p], h, m, l);
        *(u16x4*)d = h;
        *(u16x4*)(d + (in_tile ? GX_B_IMG : 0)) = m;
        *(u16x4*)(d + (in_tile ? 2 * GX_B_IMG : 0)) = l;
      }
    }
  };

  const int nit = t_end > t_begin ? (t_end - t_begin + GX_BK - 1) / GX_BK : 0;
  if (nit > 0) {
    fetch_rows(t_begin);
    make_offsets(t_begin);
    load();                                          // step 0
    fetch_rows(t_begin + GX_BK);
    store(0);
    make_offsets(t_begin + GX_BK);
    load();                                          // step 1
    fetch_rows(t_begin + 2 * GX_BK);
    __syncthreads();
  }
  auto kloop = [&](auto na_tag) {
    constexpr int NA = decltype(na_tag)::value;
    for (int it = 0; it < nit; ++it) {
      if (it + 1 < nit) {
        if (!(OPT & 8) || it == 0) store((it & 1) ^ 1);   // step it + 1 (its loads were issued a step ago)
        make_offsets(t_begin + (it + 2) * GX_BK);      // from the rows fetched a step ago
        load();                                        // step it + 2
        fetch_rows(t_begin + (it + 3) * GX_BK);        // consumed in the next step
      }
      if constexpr (NA > 0) {
        const unsigned short* st = smem + (it & 1) * STAGE;
        const unsigned short* bp = st + NP * A_IMG;
        u16x8 af[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) af[q] = tr_frag(st + q * A_IMG, GX_SA, 0, wv * 32, lane);
        auto frags = [&](int j, u16x8 (&b)[NP]) {
#pragma unroll
          for (int q = 0; q < NP; ++q) b[q] = tr_frag(bp + q * GX_B_IMG, GX_SB, 0, j * 32, lane);
        };
        u16x8 bA[NP], bB[NP];
        frags(0, bA);
#pragma unroll
        for (int j = 0; j < NA; j += 2) {
          if (j + 1 < NA) frags(j + 1, bB);
          acc[j] = tile_products<NP>(af, bA, acc[j]);
          if (j + 2 < NA) frags(j + 2, bA);
          if (j + 1 < NA) acc[j + 1] = tile_products<NP>(af, bB, acc[j + 1]);
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

  // slab [Kq, F] of this token range (zeros when the range is empty: the reduction reads every slab), un-scaled (exact)
  const float unscale = NP == 2 ? cs[kCsInvE] * (1.0f / sG) : 1.0f;
  const int hh = lane >> 5, c32 = lane & 31;
  const __amdgpu_buffer_rsrc_t rsC = make_rsrc(slabs + (size_t)(slab_base + slab) * Kq * F, (uint32_t)((size_t)Kq * F * 4));
#pragma unroll
  for (int j = 0; j < GX_TN; ++j) {
    const int f = f0 + j * 32 + c32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kd = kd0 + wv * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
      buf_store1(NP == 2 ? acc[j][r] * unscale : acc[j][r], rsC,
                 (f < F && kd < Kq) ? ((uint32_t)kd * (uint32_t)F + (uint32_t)f) * 4u : kBufOob, 0);
    }
  }
}

// dw [ksize * D, F] through jobs[0 .. ksize) (one per tap; slabs in `scratch`, convx_wgrad_scratch_bytes); dout rows of pad
// tokens are not read.  offA: per-token window offset in 16-byte pieces of Ex ((title * Tp + t) * Dp / 8 * np), offB:
// per-token row offset in floats of dout (token * F) -- launch_compact_write_offsets with rowA = Dp / 8 * np.  Returns the job
// count.  A step may cut the token list at a title boundary (range_sel 2: the tokens behind vcount[1], 1: those in front)
// and give each part its own launch: the parts' slabs lie one after the other (slab_base = slabs of the launches before,
// plan_titles = the titles of this part for the slab count) and the LAST launch writes the jobs over all of them
// (jobs == NULL: none).  gparts [g_n] (np == 2): maxima of |dout| whose largest bounds every row this launch reads.
int launch_convx_wgrad(const unsigned short* ex, const float* dout, float* dw, void* scratch, int n_titles, int T, int D,
                       int ksize, int F, const int32_t* vcount, hipStream_t s, ReduceJob* jobs, const int32_t* offA,
                       const int32_t* offB, int np, const float* cs, const float* gparts, int g_n, float alpha, int range_sel,
                       int slab_base, int plan_titles, int* slabs_out, int variant) {
  const int M = n_titles * T, Dp = convx_dp(D), Kq = ksize * Dp;
  if (slabs_out) *slabs_out = 0;
  if (M == 0) {
    for (int j = 0; jobs && j < ksize; ++j) jobs[j].nslices = 0;
    return jobs ? ksize : 0;
  }
  // (the scratch is sized for the slab count of all titles; a title sub-range plans fewer)
  const int Mp = (plan_titles > 0 ? plan_titles : n_titles) * T;
  const int mblocks = gx_mblocks(Kq), nblocks = gx_nblocks(F), ns = gx_slabs(Mp, Kq, F);
  const uint32_t e_bytes = (uint32_t)std::min(convx_e_elems(n_titles, T, D, ksize, np) * 2, (size_t)0xFFFFFFFFu);
  const uint32_t g_bytes = (uint32_t)std::min((size_t)M * F * 4, (size_t)0xFFFFFFFFu);
  const dim3 grid(cdiv(ns, 8) * 8 * mblocks * nblocks);
  const bool side = vcount && range_sel == 2;
  auto go = [&](auto kern, size_t lds) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    last_launch();
    launch_k(kern, grid, dim3(GX_THREADS), lds, s, ex, e_bytes, dout, g_bytes, (float*)scratch, vcount, M, F, Kq, mblocks, nblocks,
             ns, offA, offB, vcount ? range_sel : 0, slab_base, cs, gparts, g_n);
  };
  (void)variant;
#ifdef LSTUR_ABLATE
  if (const char* e = getenv("LSTUR_CONVX_WGRAD_VARIANT")) variant = atoi(e);
  if (variant && np == 2) {
    switch (variant) {
      case 1: go(convx_wgrad_kernel<2, 1>, gx_lds(2)); break;
      case 4: go(convx_wgrad_kernel<2, 4>, gx_lds(2)); break;
      case 7: go(convx_wgrad_kernel<2, 7>, gx_lds(2)); break;
      default: go(convx_wgrad_kernel<2, 15>, gx_lds(2)); break;
    }
  } else
#endif
  if (np == 2) side ? go(convx_wgrad_kernel<2, 0, true>, gx_lds(2)) : go(convx_wgrad_kernel<2, 0, false>, gx_lds(2));
  else side ? go(convx_wgrad_kernel<3, 0, true>, gx_lds(3)) : go(convx_wgrad_kernel<3, 0, false>, gx_lds(3));
  if (slabs_out) *slabs_out = ns;
  if (!jobs) return 0;
  for (int j = 0; j < ksize; ++j)
    jobs[j] = ReduceJob{(const float*)scratch + (size_t)j * Dp * F, dw + (size_t)j * D * F, D, F, F, slab_base + ns, (size_t)Kq * F, alpha, 0.f};
  return ksize;
}

}  // namespace lstur
