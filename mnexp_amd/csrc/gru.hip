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
// exactly one wave, so LDS staging would be pure overhead.  The step is latency-bound, so fragment loads
// run a few chunks ahead of the MFMAs in a register ring and each step's gate inputs are fetched
// before its products start.
#include "lstur_device.h"
#include "lstur_internal.h"

namespace lstur {

constexpr int GW = 8;      // waves per workgroup
constexpr int GLOADS = 8;  // float4 B-fragment loads in flight per wave (ring depth = GLOADS / tiles)

// One pass of C (static) column tiles first, first+GW, ...: acc[j] = A_lds[16 x K] . Bpacked[:, tile j].
// Branch-free inside: the tile count is a template parameter, so the MFMA stream has no predication.
template <int C>
__device__ __forceinline__ void gemm_pass(const float* __restrict__ arow, __amdgpu_buffer_rsrc_t brs, uint32_t voff,
                                          int chstride, int nchunk, int first, f32x4* acc) {
  constexpr int GDEPTH = (GLOADS + C - 1) / C;
  float4 ring[GDEPTH][C];
#pragma unroll
  for (int j = 0; j < C; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // scalar byte offset of (chunk ch, tile first + j*GW); past-the-end reads return 0 and are never used
  auto soff = [&](int ch, int j) { return (uint32_t)(ch * chstride + (first + j * GW) * 256) * 4u; };
#pragma unroll
  for (int d = 0; d < GDEPTH; ++d) {
#pragma unroll
    for (int j = 0; j < C; ++j) ring[d][j] = buf_load4(brs, voff, soff(d, j));
  }
  for (int ch0 = 0; ch0 < nchunk; ch0 += GDEPTH) {
#pragma unroll
    for (int d = 0; d < GDEPTH; ++d) {
      const int ch = ch0 + d;
      if (ch < nchunk) {
        float4 a = *(const float4*)(arow + ch * 16);
        float4 b[C];
#pragma unroll
        for (int j = 0; j < C; ++j) b[j] = ring[d][j];
#pragma unroll
        for (int j = 0; j < C; ++j) ring[d][j] = buf_load4(brs, voff, soff(ch + GDEPTH, j));
        // consecutive MFMAs go to different accumulators (dependent-issue latency 40 > issue 32 cycles)
#pragma unroll
        for (int j = 0; j < C; ++j) acc[j] = mfma16(a.x, b[j].x, acc[j]);
#pragma unroll
        for (int j = 0; j < C; ++j) acc[j] = mfma16(a.y, b[j].y, acc[j]);
#pragma unroll
        for (int j = 0; j < C; ++j) acc[j] = mfma16(a.z, b[j].z, acc[j]);
#pragma unroll
        for (int j = 0; j < C; ++j) acc[j] = mfma16(a.w, b[j].w, acc[j]);
      }
    }
  }
}

// acc[i] = A_lds[16 x K] . Bpacked[:, tile(i)]   for the tiles wave wv owns (wv, wv+GW, ...), MAXT in {2, 4, 8}
template <int MAXT>
__device__ __forceinline__ void wave_gemm16(const float* __restrict__ a_lds, int lda, const float* __restrict__ bp,
                                            int np, int nchunk, int ntiles, int wv, int lane, f32x4* acc) {
  const int c = lane & 15, kq = lane >> 4;
  const float* arow = a_lds + c * lda + kq * 4;
  const int chstride = np * 16;
  const __amdgpu_buffer_rsrc_t brs = make_rsrc(bp, (uint32_t)nchunk * chstride * 4u);
  const uint32_t voff = (c * 16 + kq * 4) * 4;   // per-lane part; the rest of every address is wave-uniform
#pragma unroll
  for (int p = 0; p < (MAXT + 3) / 4; ++p) {
    constexpr int kPassMax = MAXT < 4 ? MAXT : 4;
    const int first = wv + p * 4 * GW;
    int cnt = first < ntiles ? (ntiles - first + GW - 1) / GW : 0;   // wave-uniform
    cnt = cnt > kPassMax ? kPassMax : cnt;
    if (kPassMax >= 4 && cnt == 4)
      gemm_pass<4>(arow, brs, voff, chstride, nchunk, first, acc + p * 4);
    else if (kPassMax >= 3 && cnt == 3)
      gemm_pass<3>(arow, brs, voff, chstride, nchunk, first, acc + p * 4);
    else if (cnt == 2)
      gemm_pass<2>(arow, brs, voff, chstride, nchunk, first, acc + p * 4);
    else if (cnt == 1)
      gemm_pass<1>(arow, brs, voff, chstride, nchunk, first, acc + p * 4);
  }
}

template <int MAXT>
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
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform -> scalar addressing of fragments
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

  f32x4 acc[MAXT];
  float xin[MAXT][4];
  // unsigned (wrapping) index arithmetic: keeps every address a 32-bit offset until its use, so no
  // 64-bit loop-invariant pointer per element is hoisted out of the time loop (VGPR blow-up otherwise)
  const uint32_t uW = W, uH = H, uH3 = 3 * H;
  for (int t = 0; t < W; ++t) {
    const uint32_t ut = t;
    // ---- phase 1: z | r ----
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
      int col = (wv + i * GW) * 16 + c;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int b = b0 + rq * 4 + j;
        xin[i][j] = (col < N1 && b < n) ? xw[((uint32_t)b * uW + ut) * uH3 + (uint32_t)col] : 0.f;
      }
    }
    wave_gemm16<MAXT>(hL, LDH, u_zr, NP1, nch, NT1, wv, lane, acc);
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
      int tile = wv + i * GW;
      int col = tile * 16 + c;
      if (tile >= NT1 || col >= N1) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int row = rq * 4 + j, b = b0 + row;
        if (b >= n) continue;
        uint32_t bt = (uint32_t)b * uW + ut;
        float g = hard_sigmoid(acc[i][j] + xin[i][j]);
        if (col < H) {
          zL[row * LDH + col] = g;
          zs[bt * uH + (uint32_t)col] = g;
        } else {
          int k = col - H;
          rs[bt * uH + (uint32_t)k] = g;
          rhL[row * LDH + k] = g * hL[row * LDH + k];
        }
      }
    }
    // candidate inputs + mask for phase 2 (issued before the barrier so they fly during it)
    float mk[4];
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
      int col = (wv + i * GW) * 16 + c;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int b = b0 + rq * 4 + j;
        xin[i][j] = (col < H && b < n) ? xw[((uint32_t)b * uW + ut) * uH3 + 2u * uH + (uint32_t)col] : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int b = b0 + rq * 4 + j;
      mk[j] = (b < n && gmask[(uint32_t)b * uW + ut]) ? 1.f : 0.f;
    }
    __syncthreads();
    // ---- phase 2: candidate + state update ----
    wave_gemm16<MAXT>(rhL, LDH, u_h, Hp, nch, NT2, wv, lane, acc);
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
      int tile = wv + i * GW;
      int col = tile * 16 + c;
      if (tile >= NT2 || col >= H) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int row = rq * 4 + j, b = b0 + row;
        if (b >= n) continue;
        uint32_t bt = (uint32_t)b * uW + ut;
        float hh = tanhf(acc[i][j] + xin[i][j]);
        float z = zL[row * LDH + col], hp = hL[row * LDH + col];
        float hn = z * hp + (1.0f - z) * hh;
        float ho = mk[j] != 0.f ? hn : hp;  // K.rnn: masked step carries the previous state/output
        hhs[bt * uH + (uint32_t)col] = hh;
        hs[((uint32_t)b * (uW + 1u) + ut + 1u) * uH + (uint32_t)col] = ho;
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

template <int MAXT>
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
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform -> scalar addressing of fragments
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

  // Every wave owns the same (row, col) elements in all three phases: lane (c, rq) of tile `tile`
  // holds rows rq*4+j, col tile*16+c -- so the element-wise work needs no extra LDS round trip.
  f32x4 acc[MAXT];
  float zv[MAXT][4], rv[MAXT][4], hv[MAXT][4], pv[MAXT][4], mk[4];
  const uint32_t uW = W, uH = H, uH3 = 3 * H;  // wrapping 32-bit index math (see gru_fwd_kernel)
  for (int t = W - 1; t >= 0; --t) {
    const uint32_t ut = t;
    // ---- fetch this step's saved activations for the elements this lane owns ----
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int b = b0 + rq * 4 + j;
      mk[j] = (b < n && gmask[(uint32_t)b * uW + ut]) ? 1.f : 0.f;
    }
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
      int col = (wv + i * GW) * 16 + c;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int b = b0 + rq * 4 + j;
        bool ok = col < H && b < n;
        uint32_t bt = ((uint32_t)b * uW + ut) * uH + (uint32_t)col;
        zv[i][j] = ok ? zs[bt] : 0.f;
        rv[i][j] = ok ? rs[bt] : 0.f;
        hv[i][j] = ok ? hhs[bt] : 0.f;
        pv[i][j] = ok ? hs[((uint32_t)b * (uW + 1u) + ut) * uH + (uint32_t)col] : 0.f;
      }
    }
    // ---- (a) element-wise part that does not need drh ----
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
      int tile = wv + i * GW;
      int col = tile * 16 + c;
      if (tile >= NT || col >= H) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int row = rq * 4 + j, b = b0 + row;
        float dh = dhL[row * LDH + col];
        float dhn = mk[j] != 0.f ? dh : 0.f;
        float z = zv[i][j], hh = hv[i][j], hp = pv[i][j];
        float dph = dhn * (1.0f - z) * (1.0f - hh * hh);
        float dpz = dhn * (hp - hh) * hs_grad(z);
        g2L[row * LDH + col] = dph;
        g1L[row * LD2 + col] = dpz;
        nL[row * LDH + col] = (mk[j] != 0.f ? 0.f : dh) + dhn * z;
        if (b < n) {
          uint32_t bt = (uint32_t)b * uW + ut;
          dgates[bt * uH3 + (uint32_t)col] = dpz;
          dgates[bt * uH3 + 2u * uH + (uint32_t)col] = dph;
          hprev[bt * uH + (uint32_t)col] = hp;
          rh[bt * uH + (uint32_t)col] = rv[i][j] * hp;
        }
      }
    }
    __syncthreads();
    // ---- (b) drh = dpre_h . U_h^T ----
    wave_gemm16<MAXT>(g2L, LDH, ut_h, Hp, nch, NT, wv, lane, acc);
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
      int tile = wv + i * GW;
      int col = tile * 16 + c;
      if (tile >= NT || col >= H) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int row = rq * 4 + j, b = b0 + row;
        float drh = acc[i][j];
        float r = rv[i][j], hp = pv[i][j];
        float dpr = drh * hp * hs_grad(r);
        g1L[row * LD2 + H + col] = dpr;
        nL[row * LDH + col] += drh * r;
        if (b < n) dgates[((uint32_t)b * uW + ut) * uH3 + uH + (uint32_t)col] = dpr;
      }
    }
    __syncthreads();
    // ---- (c) dh_prev = part + [dpre_z | dpre_r] . U_zr^T ----
    wave_gemm16<MAXT>(g1L, LD2, ut_zr, Hp, nch2, NT, wv, lane, acc);
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
      int tile = wv + i * GW;
      int col = tile * 16 + c;
      if (tile >= NT || col >= H) continue;
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
  int tiles_per_wave = cdiv(ceil16(2 * H) / 16, GW);
  if (tiles_per_wave <= 4) {
    static size_t cur = 0;
    set_lds_attr((const void*)gru_fwd_kernel<4>, lds, &cur);
    hipLaunchKernelGGL(gru_fwd_kernel<4>, dim3(cdiv(n, 16)), dim3(GW * 64), lds, s, xw, gmask, h0, u_zr_packed,
                       u_h_packed, hs, zs, rs, hhs, n, W, H);
  } else {
    static size_t cur = 0;
    set_lds_attr((const void*)gru_fwd_kernel<8>, lds, &cur);
    hipLaunchKernelGGL(gru_fwd_kernel<8>, dim3(cdiv(n, 16)), dim3(GW * 64), lds, s, xw, gmask, h0, u_zr_packed,
                       u_h_packed, hs, zs, rs, hhs, n, W, H);
  }
}

void launch_gru_bwd(const float* dh_last, const uint8_t* gmask, const float* hs, const float* zs, const float* rs,
                    const float* hhs, const float* ut_zr_packed, const float* ut_h_packed, float* dgates, float* rh,
                    float* hprev, float* dh0, int n, int W, int H, hipStream_t s) {
  if (!n) return;
  int Hp = ceil16(H), LDH = Hp + 4, LD2 = ceil16(2 * H) + 4;
  size_t lds = (size_t)(3 * 16 * LDH + 16 * LD2) * sizeof(float);
  int tiles_per_wave = cdiv(Hp / 16, GW);
  if (tiles_per_wave <= 2) {
    static size_t cur = 0;
    set_lds_attr((const void*)gru_bwd_kernel<2>, lds, &cur);
    hipLaunchKernelGGL(gru_bwd_kernel<2>, dim3(cdiv(n, 16)), dim3(GW * 64), lds, s, dh_last, gmask, hs, zs, rs, hhs,
                       ut_zr_packed, ut_h_packed, dgates, rh, hprev, dh0, n, W, H);
  } else {
    static size_t cur = 0;
    set_lds_attr((const void*)gru_bwd_kernel<4>, lds, &cur);
    hipLaunchKernelGGL(gru_bwd_kernel<4>, dim3(cdiv(n, 16)), dim3(GW * 64), lds, s, dh_last, gmask, hs, zs, rs, hhs,
                       ut_zr_packed, ut_h_packed, dgates, rh, hprev, dh0, n, W, H);
  }
}

}  // namespace lstur
