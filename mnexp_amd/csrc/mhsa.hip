// Multi-head self-attention over the tokens of one title (gfx950), forward and backward: the token stage of the
// build-defined `mhsa` news encoder (BASELINE.json configs[4]).  The reference has NO such encoder (SURVEY.md
// section 8 row a17: task/paper.py:119-130 lists cnnatt/gruatt/lstmatt/bigruatt/bilstmatt only), so the
// specification is this repo's (oracle/lstur_oracle.py: mhsa_fwd / mhsa_bwd) and parity is against that
// restatement alone.
//
//   [Q|K|V] = E . qkv_w + qkv_b                 (projection GEMMs, caller)
//   per head: S = Q K^T / sqrt(dh), pad keys masked, P = softmax_keys(S), O = P V; rows of pad tokens stay untouched
//
// Titles are short (T <= 32), so a whole (title, head) problem is ONE 32x32 MFMA tile:
//   * S^T = K Q^T on v_mfma_f32_32x32x16_bf16: the accumulator then has the query on the lane and the 16 keys of
//     a lane half in its registers, so the softmax over keys is a register reduction plus one half-swap;
//   * P^T stays in registers and feeds the next MFMA directly as its A operand (accumulator-as-operand: registers
//     8s..8s+7 are k-step s; the key order inside a k-step is permuted, the B operand reads V in the same order);
//   * backward recomputes S^T / P^T (cheaper than a [N_t, h, T, T] tape), dP^T = V dO^T lands in the same layout,
//     dQ = dS K again takes dS^T from registers; dV = P^T dO and dK = dS^T Q contract over the query (the lane
//     axis), so P^T and dS^T go through one bf16 LDS tile each.
// All MFMA operands are bf16 (round-to-nearest-even, f2bf), accumulation and softmax are fp32.
#include "lstur_device.h"
#include "lstur_internal.h"

namespace lstur {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// round-to-nearest-even to bf16 (v_cvt_pk_bf16_f32); restated in oracle/lstur_oracle.py: bf16_round
__device__ __forceinline__ unsigned short f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ f32x16 mfma32(u16x8 a, u16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// 8 consecutive fp32 -> bf16 fragment (zeros when !ok).  The load itself is unconditional -- callers pass an
// in-bounds address either way -- and the predicate is a select: a guarded load compiles to a branch plus a
// s_waitcnt per load, which serialises the loads of a tile.
__device__ __forceinline__ u16x8 ld8(const float* __restrict__ p, bool ok) {
  const float4 a = ((const float4*)p)[0], b = ((const float4*)p)[1];
  u16x8 r;
  r[0] = f2bf(a.x); r[1] = f2bf(a.y); r[2] = f2bf(a.z); r[3] = f2bf(a.w);
  r[4] = f2bf(b.x); r[5] = f2bf(b.y); r[6] = f2bf(b.z); r[7] = f2bf(b.w);
  const u16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  return ok ? r : z;
}
__device__ __forceinline__ u16x8 ld8(const unsigned short* __restrict__ p, bool ok) {   // already bf16
  const u16x8 r = *(const u16x8*)p;
  const u16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  return ok ? r : z;
}
// one element through a buffer descriptor (offset in elements; kBufOob drops the store: no branch, no s_waitcnt)
__device__ __forceinline__ void bst1(__amdgpu_buffer_rsrc_t r, uint32_t off, float v, const float*) {
  buf_store1(v, r, off == kBufOob ? kBufOob : off * 4u, 0);
}
__device__ __forceinline__ void bst1(__amdgpu_buffer_rsrc_t r, uint32_t off, float v, const unsigned short*) {
  __builtin_amdgcn_raw_buffer_store_b16((short)f2bf(v), r, (int)(off == kBufOob ? kBufOob : off * 2u), 0, 0);
}
// accumulator register r of lane half hh <-> tile row
__device__ __forceinline__ int acc_row(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }
// tile row held in element j of k-step s when an accumulator is reused as an MFMA operand
__device__ __forceinline__ int perm_row(int s, int hh, int j) { return 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3); }

constexpr int kMhsaWaves = 4;

// S^T = K Q^T (scaled, pad keys masked) -> p[16] = softmax over keys, fp32.  kmask: bit t = token t is not a pad.
template <int DH, typename QT>
__device__ __forceinline__ void mhsa_probs(const QT* __restrict__ rowp, bool rowok, int F, int hc, int hh,
                                           uint32_t kmask, float scale, float (&p)[16]) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int s = 0; s < DH / 16; ++s) {
    const u16x8 a = ld8(rowp + F + hc * DH + s * 16 + 8 * hh, rowok);   // K: tile row = key
    const u16x8 b = ld8(rowp + hc * DH + s * 16 + 8 * hh, rowok);       // Q: tile column = query
    acc = mfma32(a, b, acc);
  }
  float m = -INFINITY;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const bool ok = (kmask >> acc_row(r, hh)) & 1u;
    p[r] = ok ? acc[r] * scale : -INFINITY;
    m = fmaxf(m, p[r]);
  }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float l = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    p[r] = (p[r] == -INFINITY) ? 0.f : expf(p[r] - m);
    l += p[r];
  }
  l += __shfl_xor(l, 32, 64);
  const float inv = l > 0.f ? 1.0f / l : 0.f;     // a title of pads only: P = 0
#pragma unroll
  for (int r = 0; r < 16; ++r) p[r] *= inv;
}

// rows [0,32) x DH columns of one operand (fp32 in HBM) -> bf16 LDS tile [32][DH + 2]
template <int DH, typename QT>
__device__ __forceinline__ void stage_tile(const QT* __restrict__ base, int ld, int T, uint32_t rowmask, int lane,
                                           unsigned short (*tile)[DH + 2]) {
  constexpr int CH = DH / 8;
  for (int i = lane; i < 32 * CH; i += 64) {
    const int row = i / CH, ch = i - row * CH;
    const u16x8 v = ld8(base + (size_t)min(row, T - 1) * ld + ch * 8, row < T && ((rowmask >> row) & 1u));
#pragma unroll
    for (int j = 0; j < 8; ++j) tile[row][ch * 8 + j] = v[j];
  }
}

template <int DH, typename QT>
__global__ __launch_bounds__(kMhsaWaves * 64) void mhsa_fwd_kernel(const QT* __restrict__ qkv,
                                                                  const int32_t* __restrict__ ids,
                                                                  float* __restrict__ out, int T, int F, int heads,
                                                                  float scale) {
  constexpr int NC = (DH + 31) / 32;
  __shared__ unsigned short vt_all[kMhsaWaves][32][DH + 2];
  const int n = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r32 = lane & 31, hh = lane >> 5, F3 = 3 * F;
  const bool rowok = r32 < T;
  const uint32_t kmask = (uint32_t)__ballot(rowok && ids[(size_t)n * T + min(r32, T - 1)] != 0);
  const QT* tile0 = qkv + (size_t)n * T * F3;
  const QT* rowp = tile0 + (size_t)min(r32, T - 1) * F3;
  unsigned short (*vt)[DH + 2] = vt_all[wv];
  const __amdgpu_buffer_rsrc_t rs_out = make_rsrc(out + (size_t)n * T * F, (uint32_t)T * (uint32_t)F * 4u);
  for (int h0 = 0; h0 < heads; h0 += kMhsaWaves) {
    const int h = h0 + wv;
    const bool act = h < heads;
    const int hc = act ? h : 0;
    float p[16];
    mhsa_probs<DH, QT>(rowp, rowok, F, hc, hh, kmask, scale, p);
    __syncthreads();                                   // the previous head's reads of vt are done
    stage_tile<DH, QT>(tile0 + 2 * F + hc * DH, F3, T, 0xFFFFFFFFu, lane, vt);
    __syncthreads();
    u16x8 a[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) a[s][j] = f2bf(p[8 * s + j]);
#pragma unroll
    for (int cb = 0; cb < NC; ++cb) {
      const int c = cb * 32 + r32;
      f32x16 o;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        u16x8 b;
#pragma unroll
        for (int j = 0; j < 8; ++j) b[j] = c < DH ? vt[perm_row(s, hh, j)][c] : (unsigned short)0;
        o = mfma32(a[s], b, o);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = acc_row(r, hh);
        const bool ok = act && c < DH && ((kmask >> q) & 1u);                // pad rows: neither read nor written
        bst1(rs_out, ok ? (uint32_t)q * (uint32_t)F + (uint32_t)(h * DH + c) : kBufOob, o[r], (const float*)nullptr);
      }
    }
  }
}

template <int DH, typename QT>
__global__ __launch_bounds__(kMhsaWaves * 64) void mhsa_bwd_kernel(const QT* __restrict__ qkv,
                                                                  const float* __restrict__ dout,
                                                                  const int32_t* __restrict__ ids,
                                                                  QT* __restrict__ dqkv, int T, int F, int heads,
                                                                  float scale) {
  constexpr int NC = (DH + 31) / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  constexpr int TILE = 32 * (DH + 2), XT = 32 * 40, PER = 3 * TILE + 2 * XT;
  const int n = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned short* mine = lds + (size_t)wv * PER;
  unsigned short (*xp)[40] = (unsigned short (*)[40])mine;                 // P^T   [key][query], 80-B rows
  unsigned short (*xs)[40] = (unsigned short (*)[40])(mine + XT);          // dS^T  [key][query]
  unsigned short (*qt)[DH + 2] = (unsigned short (*)[DH + 2])(mine + 2 * XT);
  unsigned short (*kt)[DH + 2] = (unsigned short (*)[DH + 2])(mine + 2 * XT + TILE);
  unsigned short (*gt)[DH + 2] = (unsigned short (*)[DH + 2])(mine + 2 * XT + 2 * TILE);   // dO
  const int r32 = lane & 31, hh = lane >> 5, F3 = 3 * F;
  const bool rowok = r32 < T;
  const uint32_t kmask = (uint32_t)__ballot(rowok && ids[(size_t)n * T + min(r32, T - 1)] != 0);
  const QT* tile0 = qkv + (size_t)n * T * F3;
  const QT* rowp = tile0 + (size_t)min(r32, T - 1) * F3;
  const float* dtile = dout + (size_t)n * T * F;
  const float* drow = dtile + (size_t)min(r32, T - 1) * F;
  const bool qok = (kmask >> r32) & 1u;               // rows of pad tokens in `dout` were never written
  const __amdgpu_buffer_rsrc_t rs_dq =
      make_rsrc(dqkv + (size_t)n * T * F3, (uint32_t)T * (uint32_t)F3 * (uint32_t)sizeof(QT));
  for (int h0 = 0; h0 < heads; h0 += kMhsaWaves) {
    const int h = h0 + wv;
    const bool act = h < heads;
    const int hc = act ? h : 0;
    float p[16];
    mhsa_probs<DH, QT>(rowp, rowok, F, hc, hh, kmask, scale, p);
    // dP^T[key][query] = sum_c V[key][c] dO[query][c]
    f32x16 dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) dp[r] = 0.f;
#pragma unroll
    for (int s = 0; s < DH / 16; ++s) {
      const u16x8 a = ld8(rowp + 2 * F + hc * DH + s * 16 + 8 * hh, rowok);
      const u16x8 b = ld8(drow + hc * DH + s * 16 + 8 * hh, qok);
      dp = mfma32(a, b, dp);
    }
    float delta = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) delta += p[r] * dp[r];
    delta += __shfl_xor(delta, 32, 64);
    float ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) ds[r] = p[r] * (dp[r] - delta) * scale;
    __syncthreads();                                   // the previous head's LDS reads are done
    stage_tile<DH, QT>(tile0 + hc * DH, F3, T, 0xFFFFFFFFu, lane, qt);
    stage_tile<DH, QT>(tile0 + F + hc * DH, F3, T, 0xFFFFFFFFu, lane, kt);
    stage_tile<DH, float>(dtile + hc * DH, F, T, kmask, lane, gt);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      xp[acc_row(r, hh)][r32] = f2bf(p[r]);
      xs[acc_row(r, hh)][r32] = f2bf(ds[r]);
    }
    __syncthreads();
    u16x8 ad[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) ad[s][j] = f2bf(ds[8 * s + j]);
#pragma unroll
    for (int cb = 0; cb < NC; ++cb) {
      const int c = cb * 32 + r32;
      f32x16 gv, gk, gq;
#pragma unroll
      for (int r = 0; r < 16; ++r) gv[r] = gk[r] = gq[r] = 0.f;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const u16x8 ap = *(const u16x8*)&xp[r32][16 * s + 8 * hh];    // row = key, k = 8 consecutive queries
        const u16x8 as = *(const u16x8*)&xs[r32][16 * s + 8 * hh];
        u16x8 bo, bq, bk;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int q = 16 * s + 8 * hh + j;
          bo[j] = c < DH ? gt[q][c] : (unsigned short)0;
          bq[j] = c < DH ? qt[q][c] : (unsigned short)0;
          bk[j] = c < DH ? kt[perm_row(s, hh, j)][c] : (unsigned short)0;
        }
        gv = mfma32(ap, bo, gv);       // dV[key][c] = sum_q P^T[key][q]  dO[q][c]
        gk = mfma32(as, bq, gk);       // dK[key][c] = sum_q dS^T[key][q] Q[q][c]
        gq = mfma32(ad[s], bk, gq);    // dQ[q][c]   = sum_key dS^T[key][q] K[key][c]
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = acc_row(r, hh);
        // rows of pad tokens come out as exact zeros (the weight-gradient GEMM reads them); rows >= T do not exist
        const uint32_t off = (act && c < DH && row < T) ? (uint32_t)row * (uint32_t)F3 + (uint32_t)(h * DH + c) : kBufOob;
        bst1(rs_dq, off, gq[r], (const QT*)nullptr);
        bst1(rs_dq, off == kBufOob ? kBufOob : off + F, gk[r], (const QT*)nullptr);
        bst1(rs_dq, off == kBufOob ? kBufOob : off + 2 * F, gv[r], (const QT*)nullptr);
      }
    }
  }
}

bool mhsa_supported(int T, int F, int heads) {
  if (heads <= 0 || F % heads) return false;
  const int dh = F / heads;
  return T >= 1 && T <= 32 && (dh == 16 || dh == 32 || dh == 64) && F <= 512;
}

static size_t mhsa_bwd_lds(int dh) { return (size_t)kMhsaWaves * (3 * 32 * (dh + 2) + 2 * 32 * 40) * sizeof(unsigned short); }

template <typename QT>
static void mhsa_fwd_t(const QT* qkv, const int32_t* ids, float* out, int n_titles, int T, int F, int heads, hipStream_t s) {
  if (!n_titles) return;
  const int dh = F / heads;
  const float scale = 1.0f / sqrtf((float)dh);
#define LSTUR_MHSA_F(DH)                                                                                         \
  hipLaunchKernelGGL((mhsa_fwd_kernel<DH, QT>), dim3(n_titles), dim3(kMhsaWaves * 64), 0, s, qkv, ids, out, T, F, heads, \
                     scale)
  if (dh == 16) LSTUR_MHSA_F(16);
  else if (dh == 32) LSTUR_MHSA_F(32);
  else LSTUR_MHSA_F(64);
#undef LSTUR_MHSA_F
}
void launch_mhsa_fwd(const float* qkv, const int32_t* ids, float* out, int n_titles, int T, int F, int heads,
                     hipStream_t s) {
  mhsa_fwd_t<float>(qkv, ids, out, n_titles, T, F, heads, s);
}
void launch_mhsa_fwd_bf16(const unsigned short* qkv, const int32_t* ids, float* out, int n_titles, int T, int F,
                          int heads, hipStream_t s) {
  mhsa_fwd_t<unsigned short>(qkv, ids, out, n_titles, T, F, heads, s);
}

template <typename QT>
static void mhsa_bwd_t(const QT* qkv, const float* dout, const int32_t* ids, QT* dqkv, int n_titles, int T, int F,
                       int heads, hipStream_t s) {
  if (!n_titles) return;
  const int dh = F / heads;
  const float scale = 1.0f / sqrtf((float)dh);
#define LSTUR_MHSA_B(DH)                                                                                               \
  do {                                                                                                                 \
    static bool attr_set = false;                                                                                      \
    if (!attr_set) {   /* dh = 64 needs 70 KB of LDS (4 waves x 5 tiles) */                                            \
      (void)hipFuncSetAttribute((const void*)mhsa_bwd_kernel<DH, QT>, hipFuncAttributeMaxDynamicSharedMemorySize,      \
                                (int)mhsa_bwd_lds(DH));                                                                \
      attr_set = true;                                                                                                 \
    }                                                                                                                  \
    hipLaunchKernelGGL((mhsa_bwd_kernel<DH, QT>), dim3(n_titles), dim3(kMhsaWaves * 64), mhsa_bwd_lds(DH), s, qkv, dout, \
                       ids, dqkv, T, F, heads, scale);                                                                 \
  } while (0)
  if (dh == 16) LSTUR_MHSA_B(16);
  else if (dh == 32) LSTUR_MHSA_B(32);
  else LSTUR_MHSA_B(64);
#undef LSTUR_MHSA_B
}
void launch_mhsa_bwd(const float* qkv, const float* dout, const int32_t* ids, float* dqkv, int n_titles, int T, int F,
                     int heads, hipStream_t s) {
  mhsa_bwd_t<float>(qkv, dout, ids, dqkv, n_titles, T, F, heads, s);
}
void launch_mhsa_bwd_bf16(const unsigned short* qkv, const float* dout, const int32_t* ids, unsigned short* dqkv,
                          int n_titles, int T, int F, int heads, hipStream_t s) {
  mhsa_bwd_t<unsigned short>(qkv, dout, ids, dqkv, n_titles, T, F, heads, s);
}

}  // namespace lstur
