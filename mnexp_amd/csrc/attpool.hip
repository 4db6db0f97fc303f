// Masking -> Dropout -> additive attention pool over title tokens (gfx950), forward and backward.
//
// Reference: keras.layers.Masking() + Dropout + models.SimpleAttentionMaskSupport
// (task/paper.py:157-158, models.py:474-489):
//   mask_t = any_f(c[t,f] != 0);  x = dropout(c);  a_t = tanh(x_t . k + b);
//   e_t = exp(a_t) * mask_t;  w_t = e_t / (sum_t e_t + 1e-7);  pooled_f = sum_t x[t,f] w_t
// No max-subtraction (tanh-bounded), an all-masked title pools to 0.
//
// HBM-bound streaming kernels, no LDS tiles and no block barriers on the hot path:
//   forward : att_score (one wave per token row: x.k by wavefront shuffle)  ->  att_pool (one thread per
//             (title, 4 filters): normalises the T weights and accumulates the pooled row)
//   backward: att_bwd_dw (wave per token: dw_t = dp . x_t) -> att_bwd_ds (thread per title) ->
//             att_bwd_dc (persistent blocks over token ranges: dconv rows + parameter-gradient partials
//             kept in registers).
// Rows of pad tokens (id == 0) are zero by construction and are neither read nor written when `ids` is given
// (the conv kernels skip them too).  Dropout masks are re-derived from the counter hash in every pass.
#include "lstur_device.h"
#include "lstur_internal.h"

namespace lstur {

__device__ __forceinline__ float4 drop4(float4 v, uint32_t key, uint32_t e, uint32_t thresh, float scale) {
  if (thresh) {          // e is a multiple of 4: two hashes decide the four elements
    bool k[4];
    keep_quad(key, e, thresh, k);
    v.x = k[0] ? v.x * scale : 0.f;
    v.y = k[1] ? v.y * scale : 0.f;
    v.z = k[2] ? v.z * scale : 0.f;
    v.w = k[3] ? v.w * scale : 0.f;
  }
  return v;
}

// does the gradient pass through this (dropped) activation?  relu: kept AND c > 0 (the conv's ReLU, cnnatt);
// !relu: kept (a token stage without ReLU -- the mhsa encoder; an exact 0 there reads as dropped)
__device__ __forceinline__ bool grad_gate(float x, int relu) { return relu ? x > 0.f : x != 0.f; }

// ---- forward 1: per-token score.  a[row] = tanh(x.k + b), e[row] = exp(a) * any(c != 0) ------------------
template <int VEC>
__global__ __launch_bounds__(256) void att_score_kernel(const float* __restrict__ c, const int32_t* __restrict__ ids,
                                                        const float* __restrict__ att_w,
                                                        const float* __restrict__ att_b, float* __restrict__ a_out,
                                                        float* __restrict__ e_out, int rows, int F, uint32_t thresh,
                                                        uint32_t key, float scale) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  if (ids && ids[row] == 0) {
    if (lane == 0) {
      a_out[row] = 0.f;
      e_out[row] = 0.f;
    }
    return;
  }
  const float* src = c + (size_t)row * F;
  const uint32_t e0 = (uint32_t)row * (uint32_t)F;
  float s = 0.f;
  bool nz = false;
  if (VEC == 4) {
    for (int f4 = lane; f4 < F / 4; f4 += 64) {
      float4 v = ((const float4*)src)[f4];
      nz |= (v.x != 0.f) | (v.y != 0.f) | (v.z != 0.f) | (v.w != 0.f);
      v = drop4(v, key, e0 + f4 * 4, thresh, scale);
      float4 k = ((const float4*)att_w)[f4];
      s += v.x * k.x + v.y * k.y + v.z * k.z + v.w * k.w;
    }
  } else {
    for (int f = lane; f < F; f += 64) {
      float v = src[f];
      nz |= v != 0.f;
      if (thresh) v = keep_elem(key, e0 + f, thresh) ? v * scale : 0.f;
      s += v * att_w[f];
    }
  }
  s = wave_sum(s);
  bool m = __any(nz);
  if (lane == 0) {
    float a = tanhf(s + att_b[0]);
    a_out[row] = a;
    e_out[row] = m ? expf(a) : 0.f;
  }
}

// ---- forward 2: per-title normalisation + pool.  `w` holds e on entry, the attention weights on exit -------
template <int VEC>
__global__ __launch_bounds__(128) void att_pool_kernel(const float* __restrict__ c, float* __restrict__ w,
                                                       float* __restrict__ pooled, int T, int F, uint32_t thresh,
                                                       uint32_t key, float scale) {
  const int n = blockIdx.x;
  float* wn = w + (size_t)n * T;
  float S = 0.f;
  for (int t = 0; t < T; ++t) S += wn[t];
  const float inv = 1.0f / (S + LSTUR_EPS);
  const float* cn = c + (size_t)n * T * F;
  const int cols = VEC == 4 ? F / 4 : F;
  for (int j = threadIdx.x; j < cols; j += 128) {
    if (VEC == 4) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int t = 0; t < T; ++t) {
        float wt = wn[t] * inv;
        if (wt != 0.f) {   // masked / pad rows carry weight 0 and may hold unwritten memory
          float4 v = ((const float4*)(cn + (size_t)t * F))[j];
          v = drop4(v, key, ((uint32_t)n * T + t) * (uint32_t)F + j * 4, thresh, scale);
          acc.x += v.x * wt;
          acc.y += v.y * wt;
          acc.z += v.z * wt;
          acc.w += v.w * wt;
        }
      }
      ((float4*)(pooled + (size_t)n * F))[j] = acc;
    } else {
      float acc = 0.f;
      for (int t = 0; t < T; ++t) {
        float wt = wn[t] * inv;
        if (wt != 0.f) {
          float v = cn[(size_t)t * F + j];
          if (thresh) v = keep_elem(key, ((uint32_t)n * T + t) * (uint32_t)F + j, thresh) ? v * scale : 0.f;
          acc += v * wt;
        }
      }
      pooled[(size_t)n * F + j] = acc;
    }
  }
  __syncthreads();  // every thread has read the e values
  for (int t = threadIdx.x; t < T; t += 128) wn[t] = wn[t] * inv;
}

// ---------------------------------------------------------------------------------------------
// Title-resident forward and backward (T <= 32 tokens, F <= 512, F % 4 == 0): ONE workgroup per title keeps the
// whole title -- 8 token rows per wave, 2 float4 per lane and row, dropout already applied -- in registers, so the
// conv output is read once per pass (the streaming kernels above read it twice forward and twice backward) and a
// pass is one launch: scores by wavefront shuffle -> T weights through LDS -> pooled row / gradient rows from the
// registers.  Cross-wave sums go through LDS in fixed wave order (bitwise reproducible).
// ---------------------------------------------------------------------------------------------
constexpr int kTitleTok = 8;   // token rows per wave (4 waves)
static bool title_path(int T, int F) { return T <= 4 * kTitleTok && F <= 512 && F % 4 == 0 && !getenv("LSTUR_ATT_STREAMING"); }
bool attpool_title_path(int T, int F) { return title_path(T, F); }

__device__ __forceinline__ float dot4(const float4& a, const float4& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// The title's non-pad bit mask (bit t = token t exists and is not a pad), in an SGPR.  `live` (one word per title,
// written by the token compaction) makes it ONE scalar load; without it the wave derives it from the ids by ballot.
// (The per-row `ids[row]` reads this replaces were a vector-load round trip in FRONT of the row loads: the kernels are
// latency chains per workgroup -- ids -> rows -> score -> barrier -> pool -- at 3-4 workgroups per CU, and that hop was a
// quarter of the chain.)
__device__ __forceinline__ uint32_t title_mask(const uint32_t* __restrict__ live, const int32_t* __restrict__ ids, int n, int T,
                                               int lane) {
  const uint32_t all = T >= 32 ? 0xFFFFFFFFu : ((1u << T) - 1u);
  if (live) return __builtin_amdgcn_readfirstlane(live[n]) & all;
  if (!ids) return all;
  const unsigned long long b = __ballot(lane < T && ids[(size_t)n * T + min(lane, T - 1)] != 0);
  return (uint32_t)b & all;
}
// rows of this wave: t = wv + 4 i (live iff bit t of lm).  x = dropout(c[row]) (0 for dead rows); nz = "my 8 values of row i
// hold a non-zero" (before dropout)
// v * scale where the keep bit is set, else 0: bit -> all-ones / zero word (v_bfe_i32) -> AND, 2 instructions per element
__device__ __forceinline__ float keep_scaled(float v, uint32_t bits, int pos, float scale) {
  const int m = __builtin_amdgcn_sbfe((int)bits, pos, 1);
  return __int_as_float(__float_as_int(v * scale) & m);
}
// kbits (nullable): the keep decisions of this title's conv-output dropout, precomputed by launch_attpool_keep_bits for the
// same (key, thresh): word h of thread (wv, lane) holds bit 4 i + e for element e of piece h of row i.  Both attention
// passes are VALU-bound once the id loads are gone, and two hashes per piece (three quarter-rate integer multiplies each)
// were a quarter of their instruction stream; the bits are made off the critical path (with the next batch's staging).
template <bool NZ = true>
__device__ __forceinline__ void title_load(const float* __restrict__ c, uint32_t lm, int n, int T, int F, int wv, int lane,
                                           uint32_t thresh, uint32_t key, float scale, float4 (&x)[kTitleTok][2],
                                           bool (&nz)[kTitleTok], const uint32_t* __restrict__ kbits = nullptr,
                                           int hn = -1) {   // hn: the title's index in the reference's order (hash position)
  const int F4 = F / 4;
  uint32_t kb[2] = {0u, 0u};
  if (kbits) {
    kb[0] = kbits[((size_t)n * 2 + 0) * 256 + threadIdx.x];
    kb[1] = kbits[((size_t)n * 2 + 1) * 256 + threadIdx.x];
  }
  // all 16 row pieces in flight at once: buffer loads with the out-of-range offset for pad / missing rows and columns
  // past F (a guarded load is a branch + s_waitcnt each, i.e. 16 serialised round trips)
  const __amdgpu_buffer_rsrc_t rsc = make_rsrc(c + (size_t)n * T * F, (uint32_t)T * (uint32_t)F * 4u);
#pragma unroll
  for (int i = 0; i < kTitleTok; ++i) {
    const int t = wv + 4 * i;
    const bool lv = (lm >> t) & 1u;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int f4 = lane + 64 * h;
      x[i][h] = buf_load4(rsc, (lv && f4 < F4) ? ((uint32_t)t * (uint32_t)F + (uint32_t)f4 * 4u) * 4u : kBufOob, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < kTitleTok; ++i) {
    const int t = wv + 4 * i;
    const uint32_t row = (uint32_t)(hn >= 0 ? hn : n) * (uint32_t)T + (uint32_t)t;
    bool any = false;
    if (kbits) {            // (dead rows were loaded as zeros and stay zero: no branch)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float4 v = x[i][h];
        if (NZ) any |= (v.x != 0.f) | (v.y != 0.f) | (v.z != 0.f) | (v.w != 0.f);
        v.x = keep_scaled(v.x, kb[h], 4 * i + 0, scale);
        v.y = keep_scaled(v.y, kb[h], 4 * i + 1, scale);
        v.z = keep_scaled(v.z, kb[h], 4 * i + 2, scale);
        v.w = keep_scaled(v.w, kb[h], 4 * i + 3, scale);
        x[i][h] = v;
      }
    } else if ((lm >> t) & 1u) {   // wave-uniform (scalar): pad / missing rows cost no hash work
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float4 v = x[i][h];
        if (NZ) any |= (v.x != 0.f) | (v.y != 0.f) | (v.z != 0.f) | (v.w != 0.f);
        x[i][h] = drop4(v, key, row * (uint32_t)F + (uint32_t)(lane + 64 * h) * 4u, thresh, scale);
      }
    }
    nz[i] = any;
  }
}
// Sums of 8 per-lane partials over the wave, one row per 8-lane group: 10 cross-lane exchanges instead of 8 x 6.  On
// return every lane of group g = lane >> 3 holds the wave total of p[g].  (Halving exchange: at distance 32 the lower half
// keeps rows 0-3 and hands 4-7 over, and so on; three plain butterfly steps finish inside the group.)
__device__ __forceinline__ float rows8_sum(const float (&p)[kTitleTok], int lane) {
  float q4[4], q2[2];
  const bool hi5 = lane & 32, hi4 = lane & 16, hi3 = lane & 8;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float keep = hi5 ? p[4 + j] : p[j], send = hi5 ? p[j] : p[4 + j];
    q4[j] = keep + __shfl_xor(send, 32, 64);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float keep = hi4 ? q4[2 + j] : q4[j], send = hi4 ? q4[j] : q4[2 + j];
    q2[j] = keep + __shfl_xor(send, 16, 64);
  }
  float r = (hi3 ? q2[1] : q2[0]) + __shfl_xor(hi3 ? q2[0] : q2[1], 8, 64);
  r += __shfl_xor(r, 4, 64);
  r += __shfl_xor(r, 2, 64);
  r += __shfl_xor(r, 1, 64);
  return r;
}

// keep bits of the conv-output dropout in the title kernels' thread layout (see title_load); dead rows and columns past F: 0
__global__ __launch_bounds__(256) void att_keep_bits_kernel(const uint32_t* __restrict__ live, uint32_t* __restrict__ kbits,
                                                            int T, int F, uint32_t thresh, uint32_t key) {
  const int n = blockIdx.x, lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), F4 = F / 4;
  const int hn = n;
  const uint32_t lm = title_mask(live, nullptr, n, T, lane);
  uint32_t kb[2] = {0u, 0u};
#pragma unroll
  for (int i = 0; i < kTitleTok; ++i) {
    const int t = wv + 4 * i;
    if (!((lm >> t) & 1u)) continue;
    const uint32_t row = (uint32_t)hn * (uint32_t)T + (uint32_t)t;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (lane + 64 * h >= F4) continue;
      bool k[4];
      keep_quad(key, row * (uint32_t)F + (uint32_t)(lane + 64 * h) * 4u, thresh, k);
      kb[h] |= ((k[0] ? 1u : 0u) | (k[1] ? 2u : 0u) | (k[2] ? 4u : 0u) | (k[3] ? 8u : 0u)) << (4 * i);
    }
  }
  kbits[((size_t)n * 2 + 0) * 256 + threadIdx.x] = kb[0];
  kbits[((size_t)n * 2 + 1) * 256 + threadIdx.x] = kb[1];
}
size_t attpool_keep_bits_words(int n_titles) { return (size_t)n_titles * 512; }
bool launch_attpool_keep_bits(const uint32_t* live, uint32_t* kbits, int n_titles, int T, int F, float drop, uint32_t key,
                              hipStream_t s) {
  if (!n_titles || !live || !kbits || !(drop > 0.f) || !title_path(T, F)) return false;
  hipLaunchKernelGGL(att_keep_bits_kernel, dim3(n_titles), dim3(256), 0, s, live, kbits, T, F, dropout_thresh(drop), key);
  return true;
}

__global__ __launch_bounds__(256) void att_fwd_title_kernel(const float* __restrict__ c, const int32_t* __restrict__ ids,
                                                            const uint32_t* __restrict__ live,
                                                            const float* __restrict__ att_w,
                                                            const float* __restrict__ att_b, float* __restrict__ a_out,
                                                            float* __restrict__ w_out, float* __restrict__ pooled, int T,
                                                            int F, uint32_t thresh, uint32_t key, float scale,
                                                            const uint32_t* __restrict__ kbits) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // e[32] | part[4][F]
  float* eL = sm;
  float* part = sm + 32;
  const int n = blockIdx.x, lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), F4 = F / 4;
  const uint32_t lm = title_mask(live, ids, n, T, lane);
  float4 x[kTitleTok][2];
  bool nz[kTitleTok];
  title_load(c, lm, n, T, F, wv, lane, thresh, key, scale, x, nz, kbits, n);
  float4 kf[2];
  const __amdgpu_buffer_rsrc_t rs_aw = make_rsrc(att_w, (uint32_t)F * 4u);
#pragma unroll
  for (int h = 0; h < 2; ++h) kf[h] = buf_load4(rs_aw, (uint32_t)(lane + 64 * h) * 16u, 0);   // past F: zeros
  const float b = att_b[0];
  // scores: the 8 rows' dot products are reduced together; group g = lane >> 3 ends up with row g, so tanh / exp run once
  // per wave instead of once per row (each was a whole-wave instruction stream for one lane's value)
  float ps[kTitleTok];
  uint32_t mrow = 0;   // bit i: row i holds a non-zero before dropout (Masking: any(c != 0))
#pragma unroll
  for (int i = 0; i < kTitleTok; ++i) {
    ps[i] = dot4(x[i][0], kf[0]) + dot4(x[i][1], kf[1]);
    mrow |= __any(nz[i]) ? (1u << i) : 0u;
  }
  const float srow = rows8_sum(ps, lane);
  {
    const int g = lane >> 3, t = wv + 4 * g;
    const bool lv = (lm >> t) & 1u;
    const float a = lv ? tanhf(srow + b) : 0.f;
    const float e = (lv && ((mrow >> g) & 1u)) ? expf(a) : 0.f;
    if ((lane & 7) == 0 && t < T) {
      a_out[(size_t)n * T + t] = a;
      eL[t] = e;
    }
  }
  __syncthreads();
  float S = 0.f;
  for (int t = 0; t < T; ++t) S += eL[t];
  const float inv = 1.0f / (S + LSTUR_EPS);
  if ((int)threadIdx.x < T) w_out[(size_t)n * T + threadIdx.x] = eL[threadIdx.x] * inv;
  float4 acc[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
#pragma unroll
  for (int i = 0; i < kTitleTok; ++i) {
    const int t = wv + 4 * i;
    const float wt = t < T ? eL[t] * inv : 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      acc[h].x += x[i][h].x * wt;
      acc[h].y += x[i][h].y * wt;
      acc[h].z += x[i][h].z * wt;
      acc[h].w += x[i][h].w * wt;
    }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h)
    if (lane + 64 * h < F4) ((float4*)(part + wv * F))[lane + 64 * h] = acc[h];
  __syncthreads();
  for (int j = threadIdx.x; j < F4; j += 256) {
    float4 r = ((float4*)part)[j];
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      const float4 q = ((float4*)(part + k * F))[j];
      r.x += q.x;
      r.y += q.y;
      r.z += q.z;
      r.w += q.w;
    }
    ((float4*)(pooled + (size_t)n * F))[j] = r;
  }
}

// the value lane `l` (wave-uniform) holds, as a scalar
__device__ __forceinline__ float lane_value(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
// (per-row scalars -- weight, 1 - a^2, ds -- live in LANE t of every wave and reach the row loops through v_readlane, i.e.
//  as SGPRs: the kernel sits at 128 VGPRs = 4 workgroups per CU; with them in per-row registers it was 132 = 3)
// tpb titles per workgroup (title0 + blockIdx.x * tpb ...): the per-title partial sums (datt_w, dconv_b, datt_b, max |dc|)
// are accumulated over the workgroup's titles -- per wave in LDS, no barrier -- and reduced over the four waves ONCE: one
// partial-sum row and one maximum per workgroup (row row0 + blockIdx.x) instead of one per title.
__global__ __launch_bounds__(256) void att_bwd_title_kernel(const float* __restrict__ c, const int32_t* __restrict__ ids,
                                                            const uint32_t* __restrict__ live,
                                                            const float* __restrict__ att_w, const float* __restrict__ a,
                                                            const float* __restrict__ w,
                                                            const float* __restrict__ dpooled, float* __restrict__ dc,
                                                            float* __restrict__ parts, int T, int F, uint32_t thresh,
                                                            uint32_t key, float scale, int ldp, int off_aw, int off_cb,
                                                            int off_ab, int relu, const uint32_t* __restrict__ kbits,
                                                            int title0, float* __restrict__ gparts, int tpb, int title_end,
                                                            int row0) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // dw[2][32] | red[4][2][F]
  float* red = sm + 64;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), F4 = F / 4;
  const __amdgpu_buffer_rsrc_t rs_aw = make_rsrc(att_w, (uint32_t)F * 4u);
  float4 kf[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    kf[h] = buf_load4(rs_aw, (uint32_t)(lane + 64 * h) * 16u, 0);   // columns past F read zeros (descriptor range)
    if (lane + 64 * h < F4) {                                        // this thread's own slots of its wave's two rows
      ((float4*)(red + (wv * 2 + 0) * F))[lane + 64 * h] = make_float4(0.f, 0.f, 0.f, 0.f);
      ((float4*)(red + (wv * 2 + 1) * F))[lane + 64 * h] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  float gm = 0.f;                                        // max |dc| of this thread's elements (the kernel gradient's fp16 scale)
  float gab = 0.f;                                       // datt_b (wave 0)
  for (int k = 0; k < tpb; ++k) {
    const int n = title0 + (int)blockIdx.x * tpb + k;
    if (n >= title_end) break;                           // (block-uniform)
    float* dwL = sm + (k & 1) * 32;                      // two buffers: a wave may be one barrier ahead of another
    const uint32_t lm = title_mask(live, ids, n, T, lane);
    float4 x[kTitleTok][2];
    bool nz[kTitleTok];
    title_load<false>(c, lm, n, T, F, wv, lane, thresh, key, scale, x, nz, kbits, n);
    float4 dp[2];
    const __amdgpu_buffer_rsrc_t rs_dp = make_rsrc(dpooled + (size_t)n * F, (uint32_t)F * 4u);
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(w + (size_t)n * T, (uint32_t)T * 4u);
    const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(a + (size_t)n * T, (uint32_t)T * 4u);
#pragma unroll
    for (int h = 0; h < 2; ++h) dp[h] = buf_load4(rs_dp, (uint32_t)(lane + 64 * h) * 16u, 0);
    // lane t: the title's attention weight (0 for masked / pad rows, and past T) and tanh output of token t
    const float w_t = buf_load1(rs_w, (uint32_t)lane * 4u, 0), a_t = buf_load1(rs_a, (uint32_t)lane * 4u, 0);
    {   // dw_t = dp . x_t for the wave's 8 rows at once (group g = lane >> 3 holds row g)
      float ps[kTitleTok];
#pragma unroll
      for (int i = 0; i < kTitleTok; ++i) ps[i] = dot4(x[i][0], dp[0]) + dot4(x[i][1], dp[1]);
      const float srow = rows8_sum(ps, lane);
      const int t = wv + 4 * (lane >> 3);
      if ((lane & 7) == 0 && t < T) dwL[t] = srow;
    }
    __syncthreads();
    // every wave: lane t holds dw_t (0 where the weight is 0: those rows contribute nothing), then ds_t
    const float dw_t = (lane < T && w_t != 0.f) ? dwL[lane] : 0.f;
    const float sum = wave_sum(dw_t * w_t);
    const float ds_t = (dw_t - sum) * w_t * (1.0f - a_t * a_t);
    const __amdgpu_buffer_rsrc_t rs_dc = make_rsrc(dc + (size_t)n * T * F, (uint32_t)T * (uint32_t)F * 4u);
    float4 gaw[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)}, gcb[2] = {gaw[0], gaw[0]};
#pragma unroll
    for (int i = 0; i < kTitleTok; ++i) {
      const int t = wv + 4 * i;
      if (!((lm >> t) & 1u)) continue;                   // pad / missing token: its dconv row is neither read nor written
      const float wt = lane_value(w_t, t), dst = lane_value(ds_t, t);   // SGPRs
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (wt != 0.f) {                                 // (scalar branch)
          const float4 xv = x[i][h];
          gaw[h].x += dst * xv.x;
          gaw[h].y += dst * xv.y;
          gaw[h].z += dst * xv.z;
          gaw[h].w += dst * xv.w;
          o.x = grad_gate(xv.x, relu) ? (dp[h].x * wt + dst * kf[h].x) * scale : 0.f;
          o.y = grad_gate(xv.y, relu) ? (dp[h].y * wt + dst * kf[h].y) * scale : 0.f;
          o.z = grad_gate(xv.z, relu) ? (dp[h].z * wt + dst * kf[h].z) * scale : 0.f;
          o.w = grad_gate(xv.w, relu) ? (dp[h].w * wt + dst * kf[h].w) * scale : 0.f;
          gcb[h].x += o.x;
          gcb[h].y += o.y;
          gcb[h].z += o.z;
          gcb[h].w += o.w;
          gm = fmaxf(fmaxf(gm, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        }
        buf_store4(o, rs_dc, (lane + 64 * h < F4) ? ((uint32_t)t * (uint32_t)F + (uint32_t)(lane + 64 * h) * 4u) * 4u : kBufOob, 0);
      }
    }
    // this title's sums into the wave's rows (each thread its own slots: no barrier), titles in order
#pragma unroll
    for (int h = 0; h < 2; ++h)
      if (lane + 64 * h < F4) {
        float4* pa = (float4*)(red + (wv * 2 + 0) * F) + lane + 64 * h;
        float4* pc = (float4*)(red + (wv * 2 + 1) * F) + lane + 64 * h;
        const float4 ra = *pa, rc = *pc;
        *pa = make_float4(ra.x + gaw[h].x, ra.y + gaw[h].y, ra.z + gaw[h].z, ra.w + gaw[h].w);
        *pc = make_float4(rc.x + gcb[h].x, rc.y + gcb[h].y, rc.z + gcb[h].z, rc.w + gcb[h].w);
      }
    if (wv == 0)     // datt_b += sum_t ds_t, in lane order (lanes past T hold 0)
      for (int t = 0; t < T; ++t) gab += lane_value(ds_t, t);
  }
  __syncthreads();   // every wave's rows are complete; the dw buffers are free
  if (gparts) {      // max |dc| of this workgroup's titles (fmaxf drops NaNs: a NaN gradient shows in the loss, not in the scale)
    gm = wave_max(gm);
    if (lane == 0) sm[wv] = gm;
  }
  float* pr = parts + (size_t)(row0 + (int)blockIdx.x) * ldp;
  for (int j = threadIdx.x; j < 2 * F; j += 256) {
    const int which = j / F, f = j - which * F;
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) r += red[(k * 2 + which) * F + f];
    pr[(which ? off_cb : off_aw) + f] = r;
  }
  if (wv == 0 && lane == 0) pr[off_ab] = gab;
  if (gparts) {
    __syncthreads();
    if (threadIdx.x == 0) gparts[row0 + (int)blockIdx.x] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
  }
}

void launch_attpool_fwd(const float* c, const int32_t* ids, const float* att_w, const float* att_b, float* pooled,
                        float* a, float* w, int n_titles, int T, int F, float drop, uint32_t key, hipStream_t s,
                        const uint32_t* live, const uint32_t* kbits) {
  if (!n_titles) return;
  uint32_t thresh = drop > 0.f ? dropout_thresh(drop) : 0u;
  float scale = drop > 0.f ? 1.0f / (1.0f - drop) : 1.0f;
  int rows = n_titles * T;
  if (title_path(T, F)) {
    hipLaunchKernelGGL(att_fwd_title_kernel, dim3(n_titles), dim3(256), (32 + 4 * F) * sizeof(float), s, c, ids, live, att_w,
                       att_b, a, w, pooled, T, F, thresh, key, scale, thresh ? kbits : nullptr);
    return;
  }
  if (F % 4 == 0) {
    hipLaunchKernelGGL(att_score_kernel<4>, dim3(cdiv(rows, 4)), dim3(256), 0, s, c, ids, att_w, att_b, a, w, rows, F,
                       thresh, key, scale);
    hipLaunchKernelGGL(att_pool_kernel<4>, dim3(n_titles), dim3(128), 0, s, c, w, pooled, T, F, thresh, key, scale);
  } else {
    hipLaunchKernelGGL(att_score_kernel<1>, dim3(cdiv(rows, 4)), dim3(256), 0, s, c, ids, att_w, att_b, a, w, rows, F,
                       thresh, key, scale);
    hipLaunchKernelGGL(att_pool_kernel<1>, dim3(n_titles), dim3(128), 0, s, c, w, pooled, T, F, thresh, key, scale);
  }
}

// ---------------------------------------------------------------------------------------------
// Backward through pool, attention, dropout and the ReLU of the conv output.
//   dw_t = sum_f dp_f x_tf ; ds_t = (dw_t - sum_t' dw_t' w_t') w_t (1 - a_t^2)
//   dx_tf = dp_f w_t + ds_t k_f ; dc_tf = dx_tf * keep*scale * [c_tf > 0]
// parts[block] holds datt_w[F], dconv_b[F], datt_b at caller-chosen offsets of an ldp-wide row (reduced by the
// caller with colsum; the training step lays the row out like the gradient bucket: ONE reduction).
// ---------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void att_bwd_dw_kernel(const float* __restrict__ c, const float* __restrict__ w,
                                                         const float* __restrict__ dpooled, float* __restrict__ dw,
                                                         int rows, int T, int F, uint32_t thresh, uint32_t key,
                                                         float scale) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  if (w[row] == 0.f) {   // masked / pad row: contributes nothing and may hold unwritten memory
    if (lane == 0) dw[row] = 0.f;
    return;
  }
  const int n = row / T;
  const float* src = c + (size_t)row * F;
  const float* dp = dpooled + (size_t)n * F;
  const uint32_t e0 = (uint32_t)row * (uint32_t)F;
  float s = 0.f;
  if (VEC == 4) {
    for (int f4 = lane; f4 < F / 4; f4 += 64) {
      float4 v = drop4(((const float4*)src)[f4], key, e0 + f4 * 4, thresh, scale);
      float4 d = ((const float4*)dp)[f4];
      s += v.x * d.x + v.y * d.y + v.z * d.z + v.w * d.w;
    }
  } else {
    for (int f = lane; f < F; f += 64) {
      float v = src[f];
      if (thresh) v = keep_elem(key, e0 + f, thresh) ? v * scale : 0.f;
      s += v * dp[f];
    }
  }
  s = wave_sum(s);
  if (lane == 0) dw[row] = s;
}

// dw (in) -> ds (out, in place)
__global__ void att_bwd_ds_kernel(float* __restrict__ dw, const float* __restrict__ w, const float* __restrict__ a,
                                  int n_titles, int T) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n_titles) return;
  float* d = dw + (size_t)n * T;
  const float* wn = w + (size_t)n * T;
  const float* an = a + (size_t)n * T;
  float sum = 0.f;
  for (int t = 0; t < T; ++t) sum += d[t] * wn[t];
  for (int t = 0; t < T; ++t) {
    float av = an[t];
    d[t] = (d[t] - sum) * wn[t] * (1.0f - av * av);
  }
}

int attpool_bwd_parts(int n_titles) { return n_titles > 0 ? n_titles : 1; }   // one partial-sum row per title

// RSUB row sub-streams per block (256 threads = RSUB x 128 column slots): more loads in flight per CU for this
// purely bandwidth-bound pass; the sub-streams' partial sums are combined through LDS in fixed order.
constexpr int kDcRsub = 2;

template <int VEC>
__global__ __launch_bounds__(128 * kDcRsub) void att_bwd_dc_kernel(
    const float* __restrict__ c, const int32_t* __restrict__ ids, const float* __restrict__ att_w,
    const float* __restrict__ w, const float* __restrict__ ds, const float* __restrict__ dpooled,
    float* __restrict__ dc, float* __restrict__ parts, int rows, int T, int F, uint32_t thresh, uint32_t key,
    float scale, int ldp, int off_aw, int off_cb, int off_ab, int relu, float* __restrict__ gparts) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [2][F] partials of sub-stream 1
  float gm = 0.f;
  const int cols = VEC == 4 ? F / 4 : F;
  const int per = (rows + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * per, r1 = min(rows, r0 + per);
  const int rsub = threadIdx.x / 128, tcol = threadIdx.x - rsub * 128;
  float* pr = parts + (size_t)blockIdx.x * ldp;
  float* pr_aw = pr + off_aw;   // datt_w[F]
  float* pr_cb = pr + off_cb;   // dconv_b[F]
  for (int j0 = 0; j0 < cols; j0 += 128) {   // uniform trip count: the barriers below are block-wide
    const int j = j0 + tcol;
    const bool act = j < cols;
    if (VEC == 4) {
      float4 kf = act ? ((const float4*)att_w)[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 gaw = make_float4(0.f, 0.f, 0.f, 0.f), gcb = gaw;
      for (int row = r0 + rsub; act && row < r1; row += kDcRsub) {
        if (ids && ids[row] == 0) continue;
        const float wt = w[row], dst = ds[row];
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (wt != 0.f) {   // masked rows: x * 0 -- their dconv row is exactly 0
          const int n = row / T;
          float4 x = drop4(((const float4*)(c + (size_t)row * F))[j], key, (uint32_t)row * (uint32_t)F + j * 4, thresh,
                           scale);
          float4 dp = ((const float4*)(dpooled + (size_t)n * F))[j];
          gaw.x += dst * x.x;
          gaw.y += dst * x.y;
          gaw.z += dst * x.z;
          gaw.w += dst * x.w;
          o.x = grad_gate(x.x, relu) ? (dp.x * wt + dst * kf.x) * scale : 0.f;
          o.y = grad_gate(x.y, relu) ? (dp.y * wt + dst * kf.y) * scale : 0.f;
          o.z = grad_gate(x.z, relu) ? (dp.z * wt + dst * kf.z) * scale : 0.f;
          o.w = grad_gate(x.w, relu) ? (dp.w * wt + dst * kf.w) * scale : 0.f;
          gcb.x += o.x;
          gcb.y += o.y;
          gcb.z += o.z;
          gcb.w += o.w;
          gm = fmaxf(fmaxf(gm, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        }
        ((float4*)(dc + (size_t)row * F))[j] = o;
      }
      if (act && rsub == 1) {
        ((float4*)red)[j] = gaw;
        ((float4*)(red + F))[j] = gcb;
      }
      __syncthreads();
      if (act && rsub == 0) {
        float4 a1 = ((float4*)red)[j], b1 = ((float4*)(red + F))[j];
        pr_aw[j * 4 + 0] = gaw.x + a1.x;
        pr_aw[j * 4 + 1] = gaw.y + a1.y;
        pr_aw[j * 4 + 2] = gaw.z + a1.z;
        pr_aw[j * 4 + 3] = gaw.w + a1.w;
        pr_cb[j * 4 + 0] = gcb.x + b1.x;
        pr_cb[j * 4 + 1] = gcb.y + b1.y;
        pr_cb[j * 4 + 2] = gcb.z + b1.z;
        pr_cb[j * 4 + 3] = gcb.w + b1.w;
      }
      __syncthreads();
    } else {
      float kf = act ? att_w[j] : 0.f, gaw = 0.f, gcb = 0.f;
      for (int row = r0 + rsub; act && row < r1; row += kDcRsub) {
        if (ids && ids[row] == 0) continue;
        const float wt = w[row], dst = ds[row];
        float o = 0.f;
        if (wt != 0.f) {
          const int n = row / T;
          float x = c[(size_t)row * F + j];
          if (thresh) x = keep_elem(key, (uint32_t)row * (uint32_t)F + j, thresh) ? x * scale : 0.f;
          gaw += dst * x;
          o = grad_gate(x, relu) ? (dpooled[(size_t)n * F + j] * wt + dst * kf) * scale : 0.f;
          gcb += o;
          gm = fmaxf(gm, fabsf(o));
        }
        dc[(size_t)row * F + j] = o;
      }
      if (act && rsub == 1) {
        red[j] = gaw;
        red[F + j] = gcb;
      }
      __syncthreads();
      if (act && rsub == 0) {
        pr_aw[j] = gaw + red[j];
        pr_cb[j] = gcb + red[F + j];
      }
      __syncthreads();
    }
  }
  if (threadIdx.x == 0) {
    float gab = 0.f;
    for (int row = r0; row < r1; ++row) gab += ds[row];   // ds is 0 on masked / pad rows
    pr[off_ab] = gab;
  }
  if (gparts) {    // max |dc| over this block's rows
    __shared__ float gms[4];
    gm = wave_max(gm);
    if ((threadIdx.x & 63) == 0) gms[threadIdx.x >> 6] = gm;
    __syncthreads();
    if (threadIdx.x == 0) gparts[blockIdx.x] = fmaxf(fmaxf(gms[0], gms[1]), fmaxf(gms[2], gms[3]));
  }
}

void launch_attpool_bwd(const float* c, const int32_t* ids, const float* att_w, const float* a, const float* w,
                        const float* dpooled, float* dc, float* parts, float* row_scratch, int n_titles, int T, int F,
                        float drop, uint32_t key, hipStream_t s, int ldp, int off_aw, int off_cb, int off_ab, int relu,
                        const uint32_t* live, const uint32_t* kbits, int title0, int title_count, float* gparts, int tpb,
                        int row0) {
  if (ldp <= 0) {   // default layout of the public entry point: datt_w[F] | dconv_b[F] | datt_b
    ldp = 2 * F + 1;
    off_aw = 0;
    off_cb = F;
    off_ab = 2 * F;
  }
  if (!n_titles) return;
  uint32_t thresh = drop > 0.f ? dropout_thresh(drop) : 0u;
  float scale = drop > 0.f ? 1.0f / (1.0f - drop) : 1.0f;
  const int rows = n_titles * T;
  const int nparts = attpool_bwd_parts(n_titles);
  if (title_path(T, F)) {   // (title0, title_count: a sub-range of the titles -- this path only; all arrays stay title-indexed)
    const int cnt = title_count >= 0 ? title_count : n_titles - title0;
    // tpb > 1 (the training step): one partial-sum row / one maximum per tpb titles, rows row0 .. row0 + cdiv(cnt, tpb) - 1;
    // the default keeps one row per title at the title's own index
    const int per = tpb > 1 ? tpb : 1, r0 = tpb > 1 ? row0 : title0;
    if (cnt > 0)
      hipLaunchKernelGGL(att_bwd_title_kernel, dim3(cdiv(cnt, per)), dim3(256), (64 + 8 * F) * sizeof(float), s, c, ids, live,
                         att_w, a, w, dpooled, dc, parts, T, F, thresh, key, scale, ldp, off_aw, off_cb, off_ab, relu,
                         thresh ? kbits : nullptr, title0, gparts, per, title0 + cnt, r0);
    return;
  }
  if (F % 4 == 0) {
    hipLaunchKernelGGL(att_bwd_dw_kernel<4>, dim3(cdiv(rows, 4)), dim3(256), 0, s, c, w, dpooled, row_scratch, rows, T, F,
                       thresh, key, scale);
    hipLaunchKernelGGL(att_bwd_ds_kernel, dim3(cdiv(n_titles, 128)), dim3(128), 0, s, row_scratch, w, a, n_titles, T);
    hipLaunchKernelGGL(att_bwd_dc_kernel<4>, dim3(nparts), dim3(128 * kDcRsub), 2 * F * sizeof(float), s, c, ids, att_w, w, row_scratch, dpooled, dc,
                       parts, rows, T, F, thresh, key, scale, ldp, off_aw, off_cb, off_ab, relu, gparts);
  } else {
    hipLaunchKernelGGL(att_bwd_dw_kernel<1>, dim3(cdiv(rows, 4)), dim3(256), 0, s, c, w, dpooled, row_scratch, rows, T, F,
                       thresh, key, scale);
    hipLaunchKernelGGL(att_bwd_ds_kernel, dim3(cdiv(n_titles, 128)), dim3(128), 0, s, row_scratch, w, a, n_titles, T);
    hipLaunchKernelGGL(att_bwd_dc_kernel<1>, dim3(nparts), dim3(128 * kDcRsub), 2 * F * sizeof(float), s, c, ids, att_w, w, row_scratch, dpooled, dc,
                       parts, rows, T, F, thresh, key, scale, ldp, off_aw, off_cb, off_ab, relu, gparts);
  }
}

}  // namespace lstur
