// Masked Keras-2.2.4 GRU over the clicked-news sequence (gfx950): recurrence and BPTT.
//
// Reference: keras.layers.Masking() + keras.layers.GRU(H)(x, initial_state=user_vec) as the reference
// instantiates it (task/paper.py:592, 597-613): gate order z|r|h, recurrent_activation hard_sigmoid,
// activation tanh, reset_after=False, K.rnn mask semantics (a masked step carries state AND output).
//   z = hs(xw_z + h U_z); r = hs(xw_r + h U_r); hh = tanh(xw_h + (r*h) U_h); h' = z h + (1-z) hh
//
// The W steps are strictly sequential per sample, samples are independent: one workgroup owns 16 samples
// (one MFMA row tile) and runs all W steps with the state in LDS.  Each step is two dependent
// [16 x H] x [H x N] products on v_mfma_f32_16x16x4_f32; the recurrent matrix is read as pre-packed
// MFMA operand fragments (misc.hip: pack_b) straight from L2 into registers -- every fragment is used by
// exactly one wave, so LDS staging would be pure overhead.
#include "lstur_device.h"
#include "lstur_internal.h"

namespace lstur {

constexpr int GW = 8;          // waves per workgroup
constexpr int kMaxTiles = 8;   // column tiles per wave and phase (N <= 8*8*16 = 1024 columns)

// acc[i] += A_lds[16 x K] . Bpacked[:, tile(i)]   for the tiles wave wv owns (wv, wv+GW, ...)
__device__ __forceinline__ void wave_gemm16(const float* __restrict__ a_lds, int lda, const float* __restrict__ bp,
                                            int np, int nchunk, int ntiles, int wv, int lane, f32x4* acc) {
  const int c = lane & 15, kq = lane >> 4;
  const float* arow = a_lds + c * lda + kq * 4;
  const float* bbase = bp + (size_t)c * 16 + kq * 4;
#pragma unroll
  for (int i = 0; i < kMaxTiles; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
  for (int ch = 0; ch < nchunk; ++ch) {
    float4 a = *(const float4*)(arow + ch * 16);
    const float* bch = bbase + (size_t)ch * np * 16;
#pragma unroll
    for (int i = 0; i < kMaxTiles; ++i) {
      int tile = wv + i * GW;
      if (tile < ntiles) {
        float4 b = *(const float4*)(bch + (size_t)tile * 256);
        acc[i] = mfma16(a.x, b.x, acc[i]);
        acc[i] = mfma16(a.y, b.y, acc[i]);
        acc[i] = mfma16(a.z, b.z, acc[i]);
        acc[i] = mfma16(a.w, b.w, acc[i]);
      }
    }
  }
}

__global__ __launch_bounds__(GW * 64) void gru_fwd_kernel(const float* __restrict__ xw,
                                                          const uint8_t* __restrict__ gmask,
                                                          const float* __restrict__ h0,
                                                          const float* __restrict__ u_zr,  // packed [H x 2H]
                                                          const float* __restrict__ u_h,   // packed [H x H]
                                                          float* __restrict__ hs, float* __restrict__ zs,
                                                          float* __restrict__ rs, float* __restrict__ hhs, int n,
                                                          int W, int H) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int nch = cdiv(H, 16), Hp = nch * 16, LDH = Hp + 4;
  const int N1 = 2 * H, NP1 = ceil16(N1), NT1 = NP1 / 16, NT2 = Hp / 16;
  float* hL = smem;
  float* rhL = hL + 16 * LDH;
  float* zL = rhL + 16 * LDH;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int c = lane & 15, rq = lane >> 4;
  const int b0 = blockIdx.x * 16;

  for (int i = tid; i < 3 * 16 * LDH; i += GW * 64) smem[i] = 0.f;
  __syncthreads();
  for (int i = tid; i < 16 * H; i += GW * 64) {
    int row = i / H, col = i - row * H, b = b0 + row;
    float v = (h0 && b < n) ? h0[(size_t)b * H + col] : 0.f;
    hL[row * LDH + col] = v;
    if (b < n) hs[(size_t)b * (W + 1) * H + col] = v;
  }
  __syncthreads();

  f32x4 acc[kMaxTiles];
  for (int t = 0; t < W; ++t) {
    // ---- phase 1: z | r ----
    wave_gemm16(hL, LDH, u_zr, NP1, nch, NT1, wv, lane, acc);
#pragma unroll
    for (int i = 0; i < kMaxTiles; ++i) {
      int tile = wv + i * GW;
      if (tile >= NT1) continue;
      int col = tile * 16 + c;
      if (col >= N1) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int row = rq * 4 + j, b = b0 + row;
        if (b >= n) continue;
        size_t bt = (size_t)b * W + t;
        float g = hard_sigmoid(acc[i][j] + xw[bt * 3 * H + col]);
        if (col < H) {
          zL[row * LDH + col] = g;
          zs[bt * H + col] = g;
        } else {
          int k = col - H;
          rs[bt * H + k] = g;
          rhL[row * LDH + k] = g * hL[row * LDH + k];
        }
      }
    }
    __syncthreads();
    // ---- phase 2: candidate + state update ----
    wave_gemm16(rhL, LDH, u_h, Hp, nch, NT2, wv, lane, acc);
#pragma unroll
    for (int i = 0; i < kMaxTiles; ++i) {
      int tile = wv + i * GW;
      if (tile >= NT2) continue;
      int col = tile * 16 + c;
      if (col >= H) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int row = rq * 4 + j, b = b0 + row;
        if (b >= n) continue;
        size_t bt = (size_t)b * W + t;
        float hh = tanhf(acc[i][j] + xw[bt * 3 * H + 2 * H + col]);
        float z = zL[row * LDH + col], hp = hL[row * LDH + col];
        float hn = z * hp + (1.0f - z) * hh;
        float ho = gmask[bt] ? hn : hp;  // K.rnn: masked step carries the previous state/output
        hhs[bt * H + col] = hh;
        hs[((size_t)b * (W + 1) + t + 1) * H + col] = ho;
        hL[row * LDH + col] = ho;
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// BPTT.  Per step (t = W-1 .. 0), with dh the gradient wrt the step's output state:
//   dhn = m ? dh : 0 ; carry = m ? 0 : dh
//   dpre_h = dhn (1-z)(1-hh^2) ; dpre_z = dhn (hp-hh) hs'(z)
//   drh = dpre_h U_h^T ; dpre_r = drh hp hs'(r)
//   dh_prev = carry + dhn z + drh r + [dpre_z | dpre_r] U_zr^T
// dgates = [dpre_z | dpre_r | dpre_h], rh = r*hp and hprev = hp are stored (time-major per sample,
// [n,W,*]) for the weight-gradient GEMMs.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float hs_grad(float g) { return (g > 0.f && g < 1.f) ? 0.2f : 0.f; }

__global__ __launch_bounds__(GW * 64) void gru_bwd_kernel(const float* __restrict__ dh_last,
                                                          const uint8_t* __restrict__ gmask,
                                                          const float* __restrict__ hs, const float* __restrict__ zs,
                                                          const float* __restrict__ rs, const float* __restrict__ hhs,
                                                          const float* __restrict__ ut_zr,  // packed [2H x H]
                                                          const float* __restrict__ ut_h,   // packed [H x H]
                                                          float* __restrict__ dgates, float* __restrict__ rh,
                                                          float* __restrict__ hprev, float* __restrict__ dh0, int n,
                                                          int W, int H) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int nch = cdiv(H, 16), Hp = nch * 16, LDH = Hp + 4;
  const int N1 = 2 * H, nch2 = cdiv(N1, 16), LD2 = nch2 * 16 + 4, NT = Hp / 16;
  float* dhL = smem;               // [16][LDH] gradient wrt current state
  float* nL = dhL + 16 * LDH;      // [16][LDH] partial dh_prev
  float* g2L = nL + 16 * LDH;      // [16][LDH] dpre_h
  float* g1L = g2L + 16 * LDH;     // [16][LD2] dpre_z | dpre_r
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int c = lane & 15, rq = lane >> 4;
  const int b0 = blockIdx.x * 16;
  const int total_lds = 3 * 16 * LDH + 16 * LD2;

  for (int i = tid; i < total_lds; i += GW * 64) smem[i] = 0.f;
  __syncthreads();
  for (int i = tid; i < 16 * H; i += GW * 64) {
    int row = i / H, col = i - row * H, b = b0 + row;
    dhL[row * LDH + col] = b < n ? dh_last[(size_t)b * H + col] : 0.f;
  }
  __syncthreads();

  f32x4 acc[kMaxTiles];
  for (int t = W - 1; t >= 0; --t) {
    // ---- (a) elementwise part that does not need drh ----
    for (int i = tid; i < 16 * H; i += GW * 64) {
      int row = i / H, col = i - row * H, b = b0 + row;
      float dpz = 0.f, dph = 0.f, part = 0.f;
      if (b < n) {
        size_t bt = (size_t)b * W + t;
        float dh = dhL[row * LDH + col];
        bool m = gmask[bt] != 0;
        float hp = hs[((size_t)b * (W + 1) + t) * H + col];
        float z = zs[bt * H + col], hh = hhs[bt * H + col];
        float dhn = m ? dh : 0.f;
        dph = dhn * (1.0f - z) * (1.0f - hh * hh);
        dpz = dhn * (hp - hh) * hs_grad(z);
        part = (m ? 0.f : dh) + dhn * z;
        dgates[bt * 3 * H + col] = dpz;
        dgates[bt * 3 * H + 2 * H + col] = dph;
        hprev[bt * H + col] = hp;
      }
      g2L[row * LDH + col] = dph;
      g1L[row * LD2 + col] = dpz;
      nL[row * LDH + col] = part;
    }
    __syncthreads();
    // ---- (b) drh = dpre_h . U_h^T ----
    wave_gemm16(g2L, LDH, ut_h, Hp, nch, NT, wv, lane, acc);
#pragma unroll
    for (int i = 0; i < kMaxTiles; ++i) {
      int tile = wv + i * GW;
      if (tile >= NT) continue;
      int col = tile * 16 + c;
      if (col >= H) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int row = rq * 4 + j, b = b0 + row;
        float dpr = 0.f;
        if (b < n) {
          size_t bt = (size_t)b * W + t;
          float drh = acc[i][j];
          float r = rs[bt * H + col];
          float hp = hs[((size_t)b * (W + 1) + t) * H + col];
          dpr = drh * hp * hs_grad(r);
          dgates[bt * 3 * H + H + col] = dpr;
          rh[bt * H + col] = r * hp;
          nL[row * LDH + col] += drh * r;
        }
        g1L[row * LD2 + H + col] = dpr;
      }
    }
    __syncthreads();
    // ---- (c) dh_prev = part + [dpre_z | dpre_r] . U_zr^T ----
    wave_gemm16(g1L, LD2, ut_zr, Hp, nch2, NT, wv, lane, acc);
#pragma unroll
    for (int i = 0; i < kMaxTiles; ++i) {
      int tile = wv + i * GW;
      if (tile >= NT) continue;
      int col = tile * 16 + c;
      if (col >= H) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int row = rq * 4 + j;
        dhL[row * LDH + col] = nL[row * LDH + col] + acc[i][j];
      }
    }
    __syncthreads();
  }
  for (int i = tid; i < 16 * H; i += GW * 64) {
    int row = i / H, col = i - row * H, b = b0 + row;
    if (b < n) dh0[(size_t)b * H + col] = dhL[row * LDH + col];
  }
}

static void set_lds_attr(const void* fn, size_t lds, size_t* cur) {
  if (lds > 65536 && lds > *cur) {
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    *cur = lds;
  }
}

void launch_gru_fwd(const float* xw, const uint8_t* gmask, const float* h0, const float* u_zr_packed,
                    const float* u_h_packed, float* hs, float* zs, float* rs, float* hhs, int n, int W, int H,
                    hipStream_t s) {
  if (!n) return;
  int Hp = ceil16(H), LDH = Hp + 4;
  size_t lds = (size_t)3 * 16 * LDH * sizeof(float);
  static size_t cur = 0;
  set_lds_attr((const void*)gru_fwd_kernel, lds, &cur);
  hipLaunchKernelGGL(gru_fwd_kernel, dim3(cdiv(n, 16)), dim3(GW * 64), lds, s, xw, gmask, h0, u_zr_packed, u_h_packed,
                     hs, zs, rs, hhs, n, W, H);
}

void launch_gru_bwd(const float* dh_last, const uint8_t* gmask, const float* hs, const float* zs, const float* rs,
                    const float* hhs, const float* ut_zr_packed, const float* ut_h_packed, float* dgates, float* rh,
                    float* hprev, float* dh0, int n, int W, int H, hipStream_t s) {
  if (!n) return;
  int Hp = ceil16(H), LDH = Hp + 4, LD2 = ceil16(2 * H) + 4;
  size_t lds = (size_t)(3 * 16 * LDH + 16 * LD2) * sizeof(float);
  static size_t cur = 0;
  set_lds_attr((const void*)gru_bwd_kernel, lds, &cur);
  hipLaunchKernelGGL(gru_bwd_kernel, dim3(cdiv(n, 16)), dim3(GW * 64), lds, s, dh_last, gmask, hs, zs, rs, hhs,
                     ut_zr_packed, ut_h_packed, dgates, rh, hprev, dh0, n, W, H);
}

}  // namespace lstur
