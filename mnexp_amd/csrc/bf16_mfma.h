// bf16 MFMA helpers shared by gemm_bf16.hip and conv3.hip (gfx950).
#pragma once
#include "lstur_device.h"

namespace lstur {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

// 4 x fp32 -> 4 x bf16, round-to-nearest-even (2 x v_cvt_pk_bf16_f32; = oracle bf16_round)
__device__ __forceinline__ u16x4 cvt4(float4 v) {
  const f32x4v f = {v.x, v.y, v.z, v.w};
  return __builtin_bit_cast(u16x4, __builtin_convertvector(f, bf16x4));
}
__device__ __forceinline__ float bf2f(unsigned short b) { return __uint_as_float((uint32_t)b << 16); }
__device__ __forceinline__ f32x16 mfma32b(u16x8 a, u16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4v mfma16b(u16x8 a, u16x8 b, f32x4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// MFMA fragment (index x0 + (lane & 31), 8 consecutive k from k0 + 8 * (lane >> 5)) out of a row-major [k][x] bf16
// tile: two transposed reads of a 4 x 16 block per 16-lane group.  EXEC must be full.
__device__ __forceinline__ u16x8 tr_frag(const unsigned short* tile, int stride, int k0, int x0, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const unsigned short* a = tile + (k0 + 8 * (g >> 1) + q) * stride + x0 + 16 * (g & 1) + 4 * p;
  typedef __attribute__((address_space(3))) s16x4 lds4;
  const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4*)a);
  const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4*)(a + 4 * stride));
  u16x8 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    r[j] = (unsigned short)v0[j];
    r[4 + j] = (unsigned short)v1[j];
  }
  return r;
}
// row stride (elements) of a [k][X] tile: bytes = 64 (mod 128) -> the 4 k-rows of a transposed read start 16 banks
// apart (64 banks of 4 bytes): conflict-free
__host__ __device__ constexpr int tr_stride(int X) { return ((X * 2 - 64 + 127) / 128 * 128 + 64) / 2; }

// x = h + m + l exactly to ~2^-25 relative, each part a bf16 (8 significant bits): the six products
// hh + hm + mh + hl + lh + mm of two such splits reproduce the fp32 product to ~2^-23 relative (the dropped terms
// ml, lm, ll are below 2^-24), i.e. an fp32-grade contraction on the bf16 matrix cores (16x the fp32 MFMA rate / 6).
__device__ __forceinline__ void split3(float4 v, u16x4& h, u16x4& m, u16x4& l) {
  h = cvt4(v);
  const float4 r1 = make_float4(v.x - bf2f(h[0]), v.y - bf2f(h[1]), v.z - bf2f(h[2]), v.w - bf2f(h[3]));
  m = cvt4(r1);
  const float4 r2 = make_float4(r1.x - bf2f(m[0]), r1.y - bf2f(m[1]), r1.z - bf2f(m[2]), r1.w - bf2f(m[3]));
  l = cvt4(r2);
}

}  // namespace lstur
