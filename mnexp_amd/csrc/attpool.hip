// Masking -> Dropout -> additive attention pool over title tokens (gfx950), forward and backward.
//
// Reference: keras.layers.Masking() + Dropout + models.SimpleAttentionMaskSupport
// (task/paper.py:157-158, models.py:474-489):
//   mask_t = any_f(c[t,f] != 0);  x = dropout(c);  a_t = tanh(x_t . k + b);
//   e_t = exp(a_t) * mask_t;  w_t = e_t / (sum_t e_t + 1e-7);  pooled_f = sum_t x[t,f] w_t
// No max-subtraction (tanh-bounded), an all-masked title pools to 0.  One title tile [T][F] lives in
// LDS; row reductions are wavefront-shuffle reductions, the pool is a column pass.
#include "lstur_device.h"
#include "lstur_internal.h"

namespace lstur {

// Title tile [T][F] -> LDS.  Rows of pad tokens (id == 0) are zero by construction (the conv output is
// multiplied by (id != 0)); with ids given they are not read at all (the conv kernels skip them too).
__device__ __forceinline__ void load_title_tile(const float* __restrict__ src, const int32_t* __restrict__ ids,
                                                float* __restrict__ xs, int T, int F) {
  if ((F & 3) == 0) {
    const int F4 = F / 4;
    for (int i = threadIdx.x; i < T * F4; i += blockDim.x) {
      int t = i / F4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!ids || ids[t] != 0) v = ((const float4*)src)[i];
      ((float4*)xs)[i] = v;
    }
  } else {
    for (int i = threadIdx.x; i < T * F; i += blockDim.x) xs[i] = (!ids || ids[i / F] != 0) ? src[i] : 0.f;
  }
}

__global__ __launch_bounds__(256) void attpool_fwd_kernel(const float* __restrict__ c,
                                                          const int32_t* __restrict__ ids,
                                                          const float* __restrict__ att_w,
                                                          const float* __restrict__ att_b,
                                                          float* __restrict__ pooled, float* __restrict__ a_out,
                                                          float* __restrict__ w_out, int T, int F, uint32_t thresh,
                                                          uint32_t key, float scale) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;           // [T][F]
  float* es = smem + T * F;   // [T] masked exp
  float* as = es + T;         // [T] tanh
  const int n = blockIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  load_title_tile(c + (size_t)n * T * F, ids ? ids + (size_t)n * T : nullptr, xs, T, F);
  __syncthreads();
  const float b = att_b[0];
  for (int t = wv; t < T; t += 4) {
    float* row = xs + t * F;
    bool nz = false;
    float s = 0.f;
    uint32_t e0 = ((uint32_t)n * T + t) * (uint32_t)F;
    for (int f = lane; f < F; f += 64) {
      float v = row[f];
      nz |= v != 0.f;
      if (thresh) {
        v = keep_elem(key, e0 + f, thresh) ? v * scale : 0.f;
        row[f] = v;
      }
      s += v * att_w[f];
    }
    s = wave_sum(s);
    bool m = __any(nz);
    if (lane == 0) {
      float a = tanhf(s + b);
      as[t] = a;
      es[t] = m ? expf(a) : 0.f;
    }
  }
  __syncthreads();
  float S = 0.f;
  for (int t = 0; t < T; ++t) S += es[t];
  const float inv = 1.0f / (S + LSTUR_EPS);
  for (int f = threadIdx.x; f < F; f += 256) {
    float acc = 0.f;
    for (int t = 0; t < T; ++t) acc += xs[t * F + f] * (es[t] * inv);
    pooled[(size_t)n * F + f] = acc;
  }
  for (int t = threadIdx.x; t < T; t += 256) {
    a_out[(size_t)n * T + t] = as[t];
    w_out[(size_t)n * T + t] = es[t] * inv;
  }
}

void launch_attpool_fwd(const float* c, const int32_t* ids, const float* att_w, const float* att_b, float* pooled,
                        float* a, float* w, int n_titles, int T, int F, float drop, uint32_t key, hipStream_t s) {
  if (!n_titles) return;
  size_t lds = ((size_t)T * F + 2 * T) * sizeof(float);
  static size_t attr = 0;
  if (lds > 65536 && lds > attr) {
    (void)hipFuncSetAttribute((const void*)attpool_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = lds;
  }
  uint32_t thresh = drop > 0.f ? dropout_thresh(drop) : 0u;
  float scale = drop > 0.f ? 1.0f / (1.0f - drop) : 1.0f;
  hipLaunchKernelGGL(attpool_fwd_kernel, dim3(n_titles), dim3(256), lds, s, c, ids, att_w, att_b, pooled, a, w, T, F,
                     thresh, key, scale);
}

// ---------------------------------------------------------------------------------------------
// Backward through pool, attention, dropout and the ReLU of the conv output.
//   dw_t = sum_f dp_f x_tf ; ds_t = (dw_t - sum_t' dw_t' w_t') w_t (1 - a_t^2)
//   dx_tf = dp_f w_t + ds_t k_f ; dc_tf = dx_tf * keep*scale * [c_tf > 0]
// Parameter-gradient partial sums are kept in registers across the titles a workgroup visits and
// written once: parts[wg] = datt_w[F] | dconv_b[F] | datt_b.
// ---------------------------------------------------------------------------------------------
constexpr int kFPT = 4;  // columns per thread (F <= 1024)

int attpool_bwd_parts(int n_titles) { return n_titles < 1024 ? (n_titles > 0 ? n_titles : 1) : 1024; }

__global__ __launch_bounds__(256) void attpool_bwd_kernel(const float* __restrict__ c,
                                                          const int32_t* __restrict__ ids,
                                                          const float* __restrict__ att_w,
                                                          const float* __restrict__ a_in,
                                                          const float* __restrict__ w_in,
                                                          const float* __restrict__ dpooled, float* __restrict__ dc,
                                                          float* __restrict__ parts, int n_titles, int T, int F,
                                                          uint32_t thresh, uint32_t key, float scale) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;            // [T][F]
  float* dp = xs + T * F;      // [F]
  float* ws = dp + F;          // [T]
  float* av = ws + T;          // [T]
  float* dwv = av + T;         // [T] -> ds
  float* vr = dwv + T;         // [T] 1 = row is a real token (dconv row is written)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float gaw[kFPT], gcb[kFPT], kf[kFPT];
#pragma unroll
  for (int i = 0; i < kFPT; ++i) {
    gaw[i] = 0.f;
    gcb[i] = 0.f;
    int f = threadIdx.x + i * 256;
    kf[i] = f < F ? att_w[f] : 0.f;
  }
  float gab = 0.f;

  for (int n = blockIdx.x; n < n_titles; n += gridDim.x) {
    __syncthreads();  // previous title's LDS reads done
    const int32_t* idn = ids ? ids + (size_t)n * T : nullptr;
    load_title_tile(c + (size_t)n * T * F, idn, xs, T, F);
    for (int f = threadIdx.x; f < F; f += 256) dp[f] = dpooled[(size_t)n * F + f];
    for (int t = threadIdx.x; t < T; t += 256) {
      ws[t] = w_in[(size_t)n * T + t];
      av[t] = a_in[(size_t)n * T + t];
      vr[t] = (!idn || idn[t] != 0) ? 1.f : 0.f;
    }
    __syncthreads();
    for (int t = wv; t < T; t += 4) {
      float* row = xs + t * F;
      float s = 0.f;
      uint32_t e0 = ((uint32_t)n * T + t) * (uint32_t)F;
      for (int f = lane; f < F; f += 64) {
        float v = row[f];
        if (thresh) {
          v = keep_elem(key, e0 + f, thresh) ? v * scale : 0.f;
          row[f] = v;
        }
        s += dp[f] * v;
      }
      s = wave_sum(s);
      if (lane == 0) dwv[t] = s;
    }
    __syncthreads();
    float sumdw = 0.f;
    for (int t = 0; t < T; ++t) sumdw += dwv[t] * ws[t];
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += 256) {
      float a = av[t];
      dwv[t] = (dwv[t] - sumdw) * ws[t] * (1.0f - a * a);  // ds_t
    }
    __syncthreads();
    float* dcn = dc + (size_t)n * T * F;
#pragma unroll
    for (int i = 0; i < kFPT; ++i) {
      int f = threadIdx.x + i * 256;
      if (f < F) {
        float dpf = dp[f];
        for (int t = 0; t < T; ++t) {
          float xv = xs[t * F + f];
          float ds = dwv[t];
          float dx = dpf * ws[t] + ds * kf[i];
          gaw[i] += ds * xv;
          float d = xv > 0.f ? dx * scale : 0.f;
          gcb[i] += d;
          if (vr[t] != 0.f) dcn[t * F + f] = d;   // pad rows are never read by the kernel gradient
        }
      }
    }
    if (threadIdx.x == 0)
      for (int t = 0; t < T; ++t) gab += dwv[t];
  }
  float* pr = parts + (size_t)blockIdx.x * (2 * F + 1);
#pragma unroll
  for (int i = 0; i < kFPT; ++i) {
    int f = threadIdx.x + i * 256;
    if (f < F) {
      pr[f] = gaw[i];
      pr[F + f] = gcb[i];
    }
  }
  if (threadIdx.x == 0) pr[2 * F] = gab;
}

void launch_attpool_bwd(const float* c, const int32_t* ids, const float* att_w, const float* a, const float* w,
                        const float* dpooled, float* dc, float* parts, int n_titles, int T, int F, float drop,
                        uint32_t key, hipStream_t s) {
  if (!n_titles) return;
  size_t lds = ((size_t)T * F + F + 4 * T) * sizeof(float);
  static size_t attr = 0;
  if (lds > 65536 && lds > attr) {
    (void)hipFuncSetAttribute((const void*)attpool_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = lds;
  }
  uint32_t thresh = drop > 0.f ? dropout_thresh(drop) : 0u;
  float scale = drop > 0.f ? 1.0f / (1.0f - drop) : 1.0f;
  hipLaunchKernelGGL(attpool_bwd_kernel, dim3(attpool_bwd_parts(n_titles)), dim3(256), lds, s, c, ids, att_w, a, w,
                     dpooled, dc, parts, n_titles, T, F, thresh, key, scale);
}

}  // namespace lstur
