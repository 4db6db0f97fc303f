// 16-bit MFMA helpers (bf16 and fp16 operands, fp32 accumulation) shared by the conv / GEMM / projection kernels (gfx950).
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


// ---- fp16 pairs ("f16x2"): x * s = h + l with h = fp16(x s), l = fp16(x s - h), s a power of two that puts the tensor's
// largest magnitude at 2^12 .. 2^15.  h carries 11 significant bits, l the next 11 (l is a normal fp16 down to |x s| = 2^-3;
// below that its absolute error is 2^-25, i.e. < 2^-37 of the tensor's maximum), so h + l = x s to 2^-22 relative, and the
// three products hh + hl + lh of two such pairs reproduce the fp32 product to ~2^-21 relative (the dropped ll term is
// below 2^-22).  A dot product of K such products carries a random-sign error of ~sqrt(K) 2^-23 |a||b| -- below what the
// fp32 accumulation of the same K terms rounds away -- at HALF the matrix-core work of the bf16 triple (3 instead of 6
// products) and 4 instead of 6 bytes per staged element.  Products of fp16 values are exact in the MFMA's fp32 accumulate;
// the scale is undone once per accumulator (exact: a power of two).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x16 mfma32h(u16x8 a, u16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// v must already carry the tensor's scale and lie inside +-65504 (fp16 range)
__device__ __forceinline__ void split2(float4 v, u16x4& h, u16x4& l) {
  const f32x4v f = {v.x, v.y, v.z, v.w};
  const f16x4 hh = __builtin_convertvector(f, f16x4);                       // round-to-nearest-even
  const f32x4v r = f - __builtin_convertvector(hh, f32x4v);                 // exact in fp32
  h = __builtin_bit_cast(u16x4, hh);
  l = __builtin_bit_cast(u16x4, __builtin_convertvector(r, f16x4));
}
// power of two s with max * s in [2^target, 2^(target + 1)), from the bits of max = |largest element| (1 for max = 0 or a
// subnormal maximum: such a tensor is zero at fp16-pair resolution).  s and 1 / s are normal floats.
__host__ __device__ __forceinline__ uint32_t pow2_scale_bits(uint32_t max_bits, int target) {
  const int ef = (int)((max_bits >> 23) & 0xFFu);
  if (ef == 0) return 0x3F800000u;
  int f = 127 + target - (ef - 127);
  f = f < 1 ? 1 : (f > 253 ? 253 : f);
  return (uint32_t)f << 23;
}
__device__ __forceinline__ float pow2_scale(uint32_t max_bits, int target) { return __uint_as_float(pow2_scale_bits(max_bits, target)); }

}  // namespace lstur
