// Device-side helpers shared by the gfx950 kernels of liblstur_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LSTUR_EPS 1e-7f  // keras.backend.epsilon()

namespace lstur {

// ---- counter-based dropout hash (restated in oracle/lstur_oracle.py: dropout_keep) ---------
__host__ __device__ __forceinline__ uint32_t fmix32(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}
__host__ __device__ __forceinline__ uint32_t dropout_key(uint32_t seed, uint32_t step, uint32_t stream) {
  uint32_t k = fmix32(seed + 0x9E3779B9u * (stream + 1u));
  return fmix32(k ^ (step * 0x7FEB352Du + 0x165667B1u));
}
// One hash decides a PAIR of elements (its two 16-bit halves against a 16-bit threshold): the masks are pure VALU work
// on the gather / attention kernels' critical path, and a hash per element made those kernels ALU-bound.
__host__ __device__ __forceinline__ uint32_t dropout_thresh(float p) { return (uint32_t)(p * 65536.0f); }
__device__ __forceinline__ uint32_t pair_hash(uint32_t key, uint32_t pair) { return fmix32(pair * 0x9E3779B1u + key); }
// keep element idx?  (thresh == 0 keeps everything)
__device__ __forceinline__ bool keep_elem(uint32_t key, uint32_t idx, uint32_t thresh) {
  const uint32_t h = pair_hash(key, idx >> 1);
  return ((idx & 1u) ? (h >> 16) : (h & 0xFFFFu)) >= thresh;
}
// the four decisions of elements e .. e+3 (e a multiple of 4) from two hashes
__device__ __forceinline__ void keep_quad(uint32_t key, uint32_t e, uint32_t thresh, bool (&k)[4]) {
  const uint32_t h0 = pair_hash(key, e >> 1), h1 = pair_hash(key, (e >> 1) + 1u);
  k[0] = (h0 & 0xFFFFu) >= thresh;
  k[1] = (h0 >> 16) >= thresh;
  k[2] = (h1 & 0xFFFFu) >= thresh;
  k[3] = (h1 >> 16) >= thresh;
}

// ---- wave (64 lanes) reductions ---------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float hard_sigmoid(float x) { return fminf(fmaxf(0.2f * x + 0.5f, 0.0f), 1.0f); }

// ---- buffer (SRSRC) loads: wave-uniform base in SGPRs + ONE 32-bit per-lane offset + a scalar offset.
// Keeps streamed-operand addressing out of the VGPR file (a 64-bit flat pointer per stream otherwise);
// out-of-range reads return 0.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t r, uint32_t voff_bytes, uint32_t soff_bytes) {
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff_bytes, (int)soff_bytes, 0);
  return *(float4*)&v;
}

__device__ __forceinline__ void buf_store4(float4 v, __amdgpu_buffer_rsrc_t r, uint32_t voff_bytes, uint32_t soff_bytes) {
  __builtin_amdgcn_raw_buffer_store_b128(*(u32x4*)&v, r, (int)voff_bytes, (int)soff_bytes, 0);
}

// one dword per lane through a buffer descriptor: lanes whose offset is past the descriptor's range (e.g.
// kBufOob) load 0 and store nothing, so column guards need no exec-mask branches and no address VGPRs
constexpr uint32_t kBufOob = 0x80000000u;
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, uint32_t voff_bytes, uint32_t soff_bytes) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff_bytes, (int)soff_bytes, 0));
}
__device__ __forceinline__ void buf_store1(float v, __amdgpu_buffer_rsrc_t r, uint32_t voff_bytes, uint32_t soff_bytes) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), r, (int)voff_bytes, (int)soff_bytes, 0);
}

// position (0..3) of k-quad kq inside the 16-float row of column n in a packed MFMA operand image (misc.hip: pack_b)
__host__ __device__ __forceinline__ int pack_b_slot(int n, int kq) { return ((n >> 2) & 3) ^ ((0x9C >> (2 * kq)) & 3); }

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}


// ---- listwise scorer of one sample (dot products with its K1 candidate doc vectors, softmax, Keras' categorical
// cross-entropy with its renormalise / clip semantics, gradients): ONE wave, the user vector in registers (uv[c] = element
// lane + 64 c, zero past U).  Shared by misc.hip's score_fast_kernel and the tail of the forward recurrence (gru4.hip),
// which scores its own samples when its last state IS the user vector.   Reference: task/paper.py:446-464, 658.
template <int KM, int UC>
__device__ __forceinline__ void score_row(const float (&uv)[UC], const float* __restrict__ d, float* __restrict__ logits,
                                          float* __restrict__ probs, float* __restrict__ loss_i,
                                          float* __restrict__ correct_i, float* __restrict__ du, float* __restrict__ dd,
                                          int b, int lane, int K1, int U, float inv_n) {
  const __amdgpu_buffer_rsrc_t rsD = make_rsrc(d + (size_t)b * K1 * U, (uint32_t)K1 * (uint32_t)U * 4u);
  float dv[KM][UC];
#pragma unroll
  for (int j = 0; j < KM; ++j)
#pragma unroll
    for (int c = 0; c < UC; ++c)
      dv[j][c] = buf_load1(rsD, (j < K1 && lane + 64 * c < U) ? ((uint32_t)j * (uint32_t)U + (uint32_t)(lane + 64 * c)) * 4u : kBufOob, 0);
  float z[KM], e[KM];
#pragma unroll
  for (int j = 0; j < KM; ++j) {
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < UC; ++c) acc += uv[c] * dv[j][c];
    z[j] = wave_sum(acc);
  }
  float zmax = -INFINITY;
#pragma unroll
  for (int j = 0; j < KM; ++j) zmax = j < K1 ? fmaxf(zmax, z[j]) : zmax;
  float esum = 0.f;
#pragma unroll
  for (int j = 0; j < KM; ++j) {
    e[j] = j < K1 ? expf(z[j] - zmax) : 0.f;
    if (j < K1) esum += e[j];
  }
  float S = 0.f, p0 = 0.f, pmax = 0.f, pj[KM];
#pragma unroll
  for (int j = 0; j < KM; ++j) {
    pj[j] = e[j] / esum;
    if (j < K1) {
      S += pj[j];
      if (j == 0) p0 = pj[j];
      pmax = fmaxf(pmax, pj[j]);
    }
  }
  const float q0 = p0 / S;
  const float cq0 = fminf(fmaxf(q0, LSTUR_EPS), 1.0f - LSTUR_EPS);
  const bool inside = (q0 >= LSTUR_EPS) && (q0 <= 1.0f - LSTUR_EPS);
  const float gq0 = inside ? -1.0f / cq0 : 0.f;
  const float common = (gq0 * p0) / (S * S);
  float dot = 0.f;
#pragma unroll
  for (int j = 0; j < KM; ++j)
    if (j < K1) dot += ((j == 0 ? gq0 / S : 0.f) - common) * pj[j];
  if (lane == 0) {
    loss_i[b] = -logf(cq0);
    correct_i[b] = (p0 >= pmax) ? 1.f : 0.f;
  }
#pragma unroll
  for (int j = 0; j < KM; ++j)
    if (j < K1 && lane == j) {
      logits[(size_t)b * K1 + j] = z[j];
      probs[(size_t)b * K1 + j] = pj[j];
    }
  if (du == nullptr) return;
  float dz[KM];
#pragma unroll
  for (int j = 0; j < KM; ++j) dz[j] = pj[j] * (((j == 0 ? gq0 / S : 0.f) - common) - dot) * inv_n;
#pragma unroll
  for (int c = 0; c < UC; ++c) {
    const int col = lane + 64 * c;
    if (col >= U) continue;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < KM; ++j)
      if (j < K1) {
        acc += dz[j] * dv[j][c];
        dd[((size_t)b * K1 + j) * U + col] = dz[j] * uv[c];
      }
    du[(size_t)b * U + col] = acc;
  }
}

}  // namespace lstur
