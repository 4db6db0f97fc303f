// Where does the embedding gather's time go, hot and COLD?  (30 us back to back, 45 us behind a step's traffic.)
// Variants of gatherx_kernel<2, true, true> (mnexp_amd/csrc/convx.hip) over a synthetic batch of config 2's shape, each timed
// back to back (word table in the Infinity Cache) and behind 1.5 GB of unrelated traffic (everything from HBM):
//   hipcc --offload-arch=gfx950 -O3 -I mnexp_amd/csrc tools/probe_gather.hip -o tools/probe_gather && tools/probe_gather
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <random>
#include <vector>
#include "lstur_device.h"
#include "bf16_mfma.h"
using namespace lstur;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Args {
  const float4* table; const int32_t* ids; const uint32_t* live; uint4* out;
  uint32_t total8; int T, Tp, D4, C8, pad_l, pad_r; uint32_t thresh, key; float scale, sE;
};

__device__ __forceinline__ uint4 pk(u16x4 a, u16x4 b) {
  return make_uint4((uint32_t)a[0] | ((uint32_t)a[1] << 16), (uint32_t)a[2] | ((uint32_t)a[3] << 16),
                    (uint32_t)b[0] | ((uint32_t)b[1] << 16), (uint32_t)b[2] | ((uint32_t)b[3] << 16));
}
__device__ __forceinline__ void nt_store(uint4 v, uint4* p, bool nt) {
  typedef unsigned int u32x4nt __attribute__((ext_vector_type(4)));
  if (nt) __builtin_nontemporal_store(*(const u32x4nt*)&v, (u32x4nt*)p); else *p = v;
}

// MODE 0 full, 1 no table reads (values from the index), 2 no stores (one lane in 2^20 keeps the compiler honest)
// LIVE: the title's non-pad bit mask instead of the neighbours' ids;  MAP 0: grid-stride (the product kernel), 1: a block's U
// groups are adjacent
template <int MODE, bool LIVE, int U, bool NT, int MAP>
__global__ __launch_bounds__(256) void gather_chunk(Args a) {
  __shared__ uint4 xbuf[4][64 * 2];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t stride = MAP == 0 ? gridDim.x * blockDim.x : 256u;
  const uint32_t start = MAP == 0 ? blockIdx.x * blockDim.x : blockIdx.x * 256u * U;
  const uint32_t step = MAP == 0 ? U * stride : gridDim.x * 256u * U;
  for (uint32_t b0 = start; b0 < a.total8; b0 += step) {
    const uint32_t i0 = b0 + threadIdx.x;
    uint32_t c8[U], tok[U];
    bool inr[U], real[U];
    int id[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t i = min(i0 + u * stride, a.total8 - 1u);
      inr[u] = i0 + u * stride < a.total8;
      const uint32_t row = i / (uint32_t)a.C8;
      c8[u] = i - row * (uint32_t)a.C8;
      const uint32_t n = row / (uint32_t)a.Tp;
      const int t = (int)(row - n * (uint32_t)a.Tp) - a.pad_l;
      real[u] = t >= 0 && t < a.T;
      tok[u] = n * (uint32_t)a.T + (uint32_t)min(max(t, 0), a.T - 1);
      id[u] = a.ids[tok[u]];
      if (LIVE) {
        const uint32_t m = a.live[n];
        const int lo = max(t - a.pad_r, 0), hi = min(t + a.pad_l, a.T - 1);
        const uint32_t w = hi >= lo ? ((hi - lo + 1 >= 32 ? 0xFFFFFFFFu : ((1u << (hi - lo + 1)) - 1u)) << lo) : 0u;
        inr[u] = inr[u] && (m & w) != 0u;
      } else {
        bool need = false;
        for (int tt = max(t - a.pad_r, 0); tt <= min(t + a.pad_l, a.T - 1); ++tt) need |= a.ids[n * (uint32_t)a.T + (uint32_t)tt] != 0;
        inr[u] = inr[u] && need;
      }
    }
    float4 va[U], vb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (MODE == 1) {
        va[u] = make_float4((float)id[u], 1.f, 2.f, (float)c8[u]);
        vb[u] = va[u];
      } else {
        va[u] = a.table[(size_t)id[u] * a.D4 + min((int)(2u * c8[u]), a.D4 - 1)];
        vb[u] = a.table[(size_t)id[u] * a.D4 + min((int)(2u * c8[u] + 1u), a.D4 - 1)];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned long long live = __ballot(inr[u]);
      if (live == 0ull) continue;
      float4 w[2] = {va[u], vb[u]};
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const uint32_t c4 = 2u * c8[u] + hf;
        const uint32_t e = (tok[u] * (uint32_t)a.D4 + c4) * 4u;
        bool k[4] = {true, true, true, true};
        if (MODE < 3) keep_quad(a.key, e, a.thresh, k);
        w[hf].x = k[0] ? w[hf].x * a.scale : 0.f;
        w[hf].y = k[1] ? w[hf].y * a.scale : 0.f;
        w[hf].z = k[2] ? w[hf].z * a.scale : 0.f;
        w[hf].w = k[3] ? w[hf].w * a.scale : 0.f;
        if (!real[u] || (int)c4 >= a.D4) w[hf] = make_float4(0.f, 0.f, 0.f, 0.f);
        w[hf] = make_float4(w[hf].x * a.sE, w[hf].y * a.sE, w[hf].z * a.sE, w[hf].w * a.sE);
      }
      if (MODE == 4) {
        xbuf[wv][lane * 2 + 0] = *(uint4*)&w[0];
        xbuf[wv][lane * 2 + 1] = *(uint4*)&w[1];
      } else {
        u16x4 h0, l0, h1, l1;
        split2(w[0], h0, l0);
        split2(w[1], h1, l1);
        xbuf[wv][lane * 2 + 0] = pk(h0, h1);
        xbuf[wv][lane * 2 + 1] = pk(l0, l1);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      uint4* o = a.out + (size_t)(b0 + wv * 64 + u * stride) * 2;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int pc = q * 64 + lane;
        const uint4 v = xbuf[wv][pc];
        bool st = (live >> (pc / 2)) & 1ull;
        if (MODE == 2) st = st && (v.x == 0x12345678u && v.y == 0x9abcdef0u);
        if (st) nt_store(v, o + pc, NT);
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// one thread = one 16-byte QUAD of a table row: every load instruction of a wave reads 1 KB of consecutive bytes (inside a
// row); 76 quad slots per padded row (the last one is the pad of D = 300 -> 304).  A block's U groups are adjacent.
template <int U, bool NT>
__global__ __launch_bounds__(256) void gather_quad(Args a) {
  __shared__ uint2 xbuf[4][64 * 2];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t Q = 2u * a.C8, total4 = a.total8 * 2u;
  for (uint32_t b0 = blockIdx.x * 256u * U; b0 < total4; b0 += gridDim.x * 256u * U) {
    const uint32_t i0 = b0 + threadIdx.x;
    uint32_t c4[U], tok[U];
    bool inr[U], real[U];
    int id[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t i = min(i0 + u * 256u, total4 - 1u);
      inr[u] = i0 + u * 256u < total4;
      const uint32_t row = i / Q;
      c4[u] = i - row * Q;
      const uint32_t n = row / (uint32_t)a.Tp;
      const int t = (int)(row - n * (uint32_t)a.Tp) - a.pad_l;
      real[u] = t >= 0 && t < a.T && (int)c4[u] < a.D4;
      tok[u] = n * (uint32_t)a.T + (uint32_t)min(max(t, 0), a.T - 1);
      const uint32_t m = a.live[n];
      const int lo = max(t - a.pad_r, 0), hi = min(t + a.pad_l, a.T - 1);
      const uint32_t w = hi >= lo ? ((hi - lo + 1 >= 32 ? 0xFFFFFFFFu : ((1u << (hi - lo + 1)) - 1u)) << lo) : 0u;
      inr[u] = inr[u] && (m & w) != 0u;
      id[u] = real[u] && inr[u] ? a.ids[tok[u]] : 0;
    }
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = a.table[(size_t)id[u] * a.D4 + min((int)c4[u], a.D4 - 1)];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned long long live = __ballot(inr[u]);
      if (live == 0ull) continue;
      float4 w = v[u];
      const uint32_t e = (tok[u] * (uint32_t)a.D4 + c4[u]) * 4u;
      bool k[4];
      keep_quad(a.key, e, a.thresh, k);
      w.x = k[0] ? w.x * a.scale : 0.f;
      w.y = k[1] ? w.y * a.scale : 0.f;
      w.z = k[2] ? w.z * a.scale : 0.f;
      w.w = k[3] ? w.w * a.scale : 0.f;
      if (!real[u]) w = make_float4(0.f, 0.f, 0.f, 0.f);
      w = make_float4(w.x * a.sE, w.y * a.sE, w.z * a.sE, w.w * a.sE);
      u16x4 h, l;
      split2(w, h, l);
      // the wave's 64 quads = 32 chunks of [h 16 B][l 16 B]; lane's quad is half (lane & 1) of chunk lane / 2
      const int ch = lane >> 1, hf = lane & 1;
      xbuf[wv][ch * 4 + hf] = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
      xbuf[wv][ch * 4 + 2 + hf] = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const uint4 o = ((const uint4*)xbuf[wv])[lane];            // 16-byte piece `lane` of the wave's 1 KB
      // piece p belongs to chunk p / 2 = quads 2 (p / 2), 2 (p / 2) + 1 -> lanes with the same live bit (a row is 76 quads: even)
      if ((live >> ((lane >> 1) * 2)) & 1ull) nt_store(o, a.out + (size_t)(b0 + u * 256u + wv * 64) / 2u * 2u + lane, NT);
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// the COVERED rows only, from a list {padded row, word id (-1: a pad row -> zeros), token index}: every wave has work and the
// table load hangs on ONE coalesced load
template <int U, bool NT>
__global__ __launch_bounds__(256) void gather_rows(Args a, const int4* __restrict__ rows, uint32_t n_chunks) {
  __shared__ uint4 xbuf[4][64 * 2];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (uint32_t b0 = blockIdx.x * 256u * U; b0 < n_chunks; b0 += gridDim.x * 256u * U) {
    const uint32_t i0 = b0 + threadIdx.x;
    uint32_t c8[U];
    bool inr[U];
    int4 r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t i = min(i0 + u * 256u, n_chunks - 1u);
      inr[u] = i0 + u * 256u < n_chunks;
      const uint32_t row = i / (uint32_t)a.C8;
      c8[u] = i - row * (uint32_t)a.C8;
      r[u] = rows[row];
    }
    float4 va[U], vb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int id = max(r[u].y, 0);
      va[u] = a.table[(size_t)id * a.D4 + min((int)(2u * c8[u]), a.D4 - 1)];
      vb[u] = a.table[(size_t)id * a.D4 + min((int)(2u * c8[u] + 1u), a.D4 - 1)];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float4 w[2] = {va[u], vb[u]};
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const uint32_t c4 = 2u * c8[u] + hf;
        const uint32_t e = ((uint32_t)r[u].z * (uint32_t)a.D4 + c4) * 4u;
        bool k[4];
        keep_quad(a.key, e, a.thresh, k);
        w[hf].x = k[0] ? w[hf].x * a.scale : 0.f;
        w[hf].y = k[1] ? w[hf].y * a.scale : 0.f;
        w[hf].z = k[2] ? w[hf].z * a.scale : 0.f;
        w[hf].w = k[3] ? w[hf].w * a.scale : 0.f;
        if (r[u].y < 0 || (int)c4 >= a.D4) w[hf] = make_float4(0.f, 0.f, 0.f, 0.f);
        w[hf] = make_float4(w[hf].x * a.sE, w[hf].y * a.sE, w[hf].z * a.sE, w[hf].w * a.sE);
      }
      u16x4 h0, l0, h1, l1;
      split2(w[0], h0, l0);
      split2(w[1], h1, l1);
      // lane's chunk goes to out[(row * C8 + c8) * 2 ..]: rows of the list are not adjacent in Ex, a row's 38 chunks are
      xbuf[wv][lane * 2 + 0] = pk(h0, h1);
      xbuf[wv][lane * 2 + 1] = pk(l0, l1);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int pc = q * 64 + lane, src = pc >> 1;                 // piece pc belongs to lane src's chunk
        const uint4 v = xbuf[wv][pc];
        const uint32_t orow = (uint32_t)__shfl(r[u].x, src, 64), oc = (uint32_t)__shfl((int)c8[u], src, 64);
        const bool ok = __shfl((int)inr[u], src, 64);
        if (ok) nt_store(v, a.out + ((size_t)orow * a.C8 + oc) * 2 + (pc & 1), NT);
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// v2: the product candidate.  LIVE mask, U chunks per thread, divisions by multiply-high, one multiplier, validity folded into
// the keep decisions, 32-bit table index.
struct FastDiv { uint32_t m, sh, one; };
static FastDiv fast_div(uint32_t d) {     // exact for n < 2^31
  FastDiv f{0, 0, d == 1};
  if (d == 1) return f;
  uint32_t L = 0;
  while ((1u << L) < d) ++L;
  f.m = (uint32_t)(((1ull << (31 + L)) + d - 1) / d);
  f.sh = L - 1;
  return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, FastDiv f) { return f.one ? n : __umulhi(n, f.m) >> f.sh; }
struct Args2 { Args a; FastDiv dc8, dtp; float mul; };
template <int U, bool NT>
__global__ __launch_bounds__(256) void gather_v2(Args2 g) {
  const Args& a = g.a;
  __shared__ uint4 xbuf[4][64 * 2];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t b0 = blockIdx.x * blockDim.x; b0 < a.total8; b0 += U * stride) {
    const uint32_t i0 = b0 + threadIdx.x;
    uint32_t c8[U], tok[U], idx[U];
    bool inr[U], real[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t i = min(i0 + u * stride, a.total8 - 1u);
      const uint32_t row = fdiv(i, g.dc8);
      c8[u] = i - row * (uint32_t)a.C8;
      const uint32_t n = fdiv(row, g.dtp);
      const int t = (int)(row - n * (uint32_t)a.Tp) - a.pad_l;
      real[u] = t >= 0 && t < a.T;
      tok[u] = n * (uint32_t)a.T + (uint32_t)min(max(t, 0), a.T - 1);
      const uint32_t m = a.live[n];
      const int lo = max(t - a.pad_r, 0), hi = min(t + a.pad_l, a.T - 1);
      const uint32_t w = hi >= lo ? ((0xFFFFFFFFu >> (31 - (hi - lo))) << lo) : 0u;
      inr[u] = i0 + u * stride < a.total8 && (m & w) != 0u;
      const int id = (real[u] && inr[u]) ? a.ids[tok[u]] : 0;
      idx[u] = (uint32_t)id * (uint32_t)a.D4 + 2u * c8[u];
    }
    float4 va[U], vb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool second = (int)(2u * c8[u] + 1u) < a.D4;
      va[u] = a.table[idx[u]];
      vb[u] = a.table[idx[u] + (second ? 1u : 0u)];
      real[u] = real[u];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned long long live = __ballot(inr[u]);
      if (live == 0ull) continue;
      float4 w[2] = {va[u], vb[u]};
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const uint32_t c4 = 2u * c8[u] + hf;
        const uint32_t e = (tok[u] * (uint32_t)a.D4 + c4) * 4u;
        bool k[4];
        keep_quad(a.key, e, a.thresh, k);
        const bool ok = real[u] && (int)c4 < a.D4;
        w[hf].x = (k[0] && ok) ? w[hf].x * g.mul : 0.f;
        w[hf].y = (k[1] && ok) ? w[hf].y * g.mul : 0.f;
        w[hf].z = (k[2] && ok) ? w[hf].z * g.mul : 0.f;
        w[hf].w = (k[3] && ok) ? w[hf].w * g.mul : 0.f;
      }
      u16x4 h0, l0, h1, l1;
      split2(w[0], h0, l0);
      split2(w[1], h1, l1);
      xbuf[wv][lane * 2 + 0] = pk(h0, h1);
      xbuf[wv][lane * 2 + 1] = pk(l0, l1);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      uint4* o = a.out + (size_t)(b0 + wv * 64 + u * stride) * 2;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int pc = q * 64 + lane;
        const uint4 v = xbuf[wv][pc];
        if ((live >> (pc / 2)) & 1ull) nt_store(v, o + pc, NT);
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// issue rate of v_mul_lo_u32 against v_add_u32 / v_mul_u32_u24: 8 independent chains per lane, one wave per SIMD
template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(uint32_t* out, uint32_t k, int iters) {
  uint32_t x[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * 8 + j + k;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (OP == 0) x[j] = x[j] * k;
      else if (OP == 1) x[j] = (x[j] + k) ^ 0x5bd1e995u;
      else if (OP == 2) x[j] = __umul24(x[j], k) + 1u;
      else x[j] = __umulhi(x[j], k) + 1u;
    }
  }
  uint32_t r = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) r ^= x[j];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

// v3: persistent workgroups, three chunks per thread in a software pipeline -- ids of chunk j + 2 and table rows of chunk
// j + 1 are in flight while chunk j is hashed, split and stored: reads and writes interleave instead of coming in bursts
struct Stg { uint32_t c8, tok, win, lv; int id; bool in, real; float4 va, vb; };
template <bool NT>
__global__ __launch_bounds__(256) void gather_v3(Args a) {
  __shared__ uint4 xbuf[4][64 * 2];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t step = gridDim.x * 256u;
  auto A = [&](uint32_t b0, Stg& s) {
    const uint32_t i = min(b0 + threadIdx.x, a.total8 - 1u);
    s.in = b0 + threadIdx.x < a.total8;
    const uint32_t row = i / (uint32_t)a.C8;
    s.c8 = i - row * (uint32_t)a.C8;
    const uint32_t n = row / (uint32_t)a.Tp;
    const int t = (int)(row - n * (uint32_t)a.Tp) - a.pad_l;
    s.real = t >= 0 && t < a.T;
    s.tok = n * (uint32_t)a.T + (uint32_t)min(max(t, 0), a.T - 1);
    const int lo = max(t - a.pad_r, 0), hi = min(t + a.pad_l, a.T - 1);
    s.win = hi >= lo ? ((0xFFFFFFFFu >> (31 - (hi - lo))) << lo) : 0u;
    s.id = a.ids[s.tok];
    s.lv = a.live[n];
  };
  auto B = [&](Stg& s) {
    s.in = s.in && (s.lv & s.win) != 0u;
    const int id = s.in ? s.id : 0;
    s.va = a.table[(size_t)id * a.D4 + min((int)(2u * s.c8), a.D4 - 1)];
    s.vb = a.table[(size_t)id * a.D4 + min((int)(2u * s.c8 + 1u), a.D4 - 1)];
  };
  auto P = [&](uint32_t b0, const Stg& s) {
    const unsigned long long live = __ballot(s.in);
    if (live == 0ull) return;
    float4 w[2] = {s.va, s.vb};
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const uint32_t c4 = 2u * s.c8 + hf;
      const uint32_t e = (s.tok * (uint32_t)a.D4 + c4) * 4u;
      bool k[4];
      keep_quad(a.key, e, a.thresh, k);
      w[hf].x = k[0] ? w[hf].x * a.scale : 0.f;
      w[hf].y = k[1] ? w[hf].y * a.scale : 0.f;
      w[hf].z = k[2] ? w[hf].z * a.scale : 0.f;
      w[hf].w = k[3] ? w[hf].w * a.scale : 0.f;
      if (!s.real || (int)c4 >= a.D4) w[hf] = make_float4(0.f, 0.f, 0.f, 0.f);
      w[hf] = make_float4(w[hf].x * a.sE, w[hf].y * a.sE, w[hf].z * a.sE, w[hf].w * a.sE);
    }
    u16x4 h0, l0, h1, l1;
    split2(w[0], h0, l0);
    split2(w[1], h1, l1);
    xbuf[wv][lane * 2 + 0] = pk(h0, h1);
    xbuf[wv][lane * 2 + 1] = pk(l0, l1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint4* o = a.out + (size_t)(b0 + wv * 64) * 2;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int pc = q * 64 + lane;
      const uint4 v = xbuf[wv][pc];
      if ((live >> (pc / 2)) & 1ull) nt_store(v, o + pc, NT);
    }
    __builtin_amdgcn_wave_barrier();
  };
  uint32_t b0 = blockIdx.x * 256u;
  if (b0 >= a.total8) return;
  Stg s0, s1, s2;
  A(b0, s0);
  A(b0 + step, s1);
  B(s0);
  for (; b0 < a.total8; b0 += step) {
    B(s1);                      // (past the end: clamped loads of lanes that are not `in` -> row 0)
    A(b0 + 2u * step, s2);
    P(b0, s0);
    s0 = s1;
    s1 = s2;
  }
}

__global__ void trash_kernel(float4* __restrict__ p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = p[i];
    v.x += 1.f;
    p[i] = v;
  }
}

int main() {
  const int T = 30, D = 300, k = 3, V = 50000, Nd = 20000, B = 64, W = 50, K1 = 5;
  const int Nt = B * (W + K1), pad_l = 1, pad_r = 1, Tp = T + 2, D4 = D / 4, C8 = (D + 7) / 8;
  std::mt19937 rng(1);
  std::vector<int32_t> titles((size_t)Nd * T), ids((size_t)Nt * T);
  for (int d = 0; d < Nd; ++d) {
    int len = 5 + (int)(rng() % (T + 1));
    for (int t = 0; t < T; ++t) titles[(size_t)d * T + t] = (d && t < len) ? 1 + (int)(rng() % (V - 1)) : 0;
  }
  for (int n = 0; n < Nt; ++n) {
    int d = 1 + (int)(rng() % (Nd - 1));
    if (rng() % 20 == 0) d = 0;
    for (int t = 0; t < T; ++t) ids[(size_t)n * T + t] = titles[(size_t)d * T + t];
  }
  std::vector<int32_t> rowlist;
  std::vector<uint32_t> live(Nt);
  size_t covered = 0, covered_real = 0;
  for (int n = 0; n < Nt; ++n) {
    uint32_t m = 0;
    for (int t = 0; t < T; ++t) if (ids[(size_t)n * T + t]) m |= 1u << t;
    live[n] = m;
    for (int tp = 0; tp < Tp; ++tp) {
      int t = tp - pad_l;
      bool need = false;
      for (int tt = std::max(t - pad_r, 0); tt <= std::min(t + pad_l, T - 1); ++tt) need |= (m >> tt) & 1u;
      covered += need;
      covered_real += need && t >= 0 && t < T;
      if (need) {
        const bool real = t >= 0 && t < T;
        rowlist.push_back(n * Tp + tp);
        rowlist.push_back(real ? ids[(size_t)n * T + t] : -1);
        rowlist.push_back(n * T + std::min(std::max(t, 0), T - 1));
        rowlist.push_back(0);
      }
    }
  }
  const double moved = (double)Nt * T * 4 + (double)covered_real * D * 4 + (double)covered * C8 * 32;
  printf("titles %d, covered rows %zu of %d (real %zu), moved %.1f MB\n", Nt, covered, Nt * Tp, covered_real, moved / 1e6);
  std::vector<float> table((size_t)V * D);
  for (auto& x : table) x = (float)((int)(rng() % 2001) - 1000) * 1e-4f;
  float *d_table; int32_t* d_ids; uint32_t* d_live; uint4* d_out; float4* d_trash;
  const size_t total8 = (size_t)Nt * Tp * C8, trash_n = (size_t)1536 << 20 >> 4;
  CK(hipMalloc(&d_table, table.size() * 4));
  CK(hipMalloc(&d_ids, ids.size() * 4));
  CK(hipMalloc(&d_live, live.size() * 4));
  CK(hipMalloc(&d_out, total8 * 32));
  CK(hipMalloc(&d_trash, trash_n * 16));
  CK(hipMemcpy(d_table, table.data(), table.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_ids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_live, live.data(), live.size() * 4, hipMemcpyHostToDevice));
  int4* d_rows;
  CK(hipMalloc(&d_rows, rowlist.size() * 4));
  CK(hipMemcpy(d_rows, rowlist.data(), rowlist.size() * 4, hipMemcpyHostToDevice));
  const uint32_t n_chunks = (uint32_t)(covered * C8);
  CK(hipMemset(d_out, 0, total8 * 32));
  CK(hipMemset(d_trash, 0, trash_n * 16));
  Args a{(const float4*)d_table, d_ids, d_live, d_out, (uint32_t)total8, T, Tp, D4, C8, pad_l, pad_r,
         dropout_thresh(0.2f), dropout_key(7, 3, 0), 1.25f, 32768.f};
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<uint4> ref(total8 * 2), got(total8 * 2);
  bool have_ref = false;
  auto run_l = [&](const char* name, auto launch, int blocks, bool check) {
    float hot = 1e9f, cold[7];
    for (int i = 0; i < 6; ++i) {
      launch();
      CK(hipStreamSynchronize(s));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (i) hot = std::min(hot, ms);
    }
    for (int i = 0; i < 7; ++i) {
      hipLaunchKernelGGL(trash_kernel, dim3(2048), dim3(256), 0, s, d_trash, trash_n);
      launch();
      CK(hipStreamSynchronize(s));
      CK(hipEventElapsedTime(&cold[i], e0, e1));
    }
    std::sort(cold, cold + 7);
    const char* ok = "";
    if (check) {
      CK(hipMemcpy(got.data(), d_out, total8 * 32, hipMemcpyDeviceToHost));
      if (!have_ref) { ref = got; have_ref = true; ok = " (reference)"; }
      else ok = memcmp(ref.data(), got.data(), total8 * 32) == 0 ? " same bytes" : " DIFFERENT BYTES";
      CK(hipMemset(d_out, 0, total8 * 32));
    }
    printf("%-44s blocks %5d  hot %6.1f us (%.2f TB/s)  cold median %6.1f us (%.2f TB/s) min %6.1f%s\n", name, blocks, hot * 1e3,
           moved / (hot * 1e-3) / 1e12, cold[3] * 1e3, moved / (cold[3] * 1e-3) / 1e12, cold[0] * 1e3, ok);
    fflush(stdout);
  };
  auto run = [&](const char* name, auto kern, int blocks, bool check, auto... extra) {
    run_l(name, [&] { hipExtLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, s, e0, e1, 0, a, extra...); }, blocks, check);
  };
  {
    uint32_t* d_r;
    CK(hipMalloc(&d_r, 256 * 256 * 4));
    auto rate = [&](const char* name, auto kern, int per) {
      float best = 1e9f;
      for (int i = 0; i < 3; ++i) {
        hipExtLaunchKernelGGL(kern, dim3(256), dim3(256), 0, s, e0, e1, 0, d_r, 0x9E3779B1u, 4096);
        CK(hipStreamSynchronize(s));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
      }
      printf("%-28s %7.1f us for 4096 x 8 x %d instructions per wave -> %.2f ns per instruction per SIMD\n", name, best * 1e3, per,
             best * 1e6 / (4096.0 * 8 * per));
    };
    rate("v_mul_lo_u32", rate_kernel<0>, 1);
    rate("v_add_u32 + v_xor_b32", rate_kernel<1>, 2);
    rate("v_mul_u32_u24 + v_add_u32", rate_kernel<2>, 2);
    rate("v_mul_hi_u32 + v_add_u32", rate_kernel<3>, 2);
  }
  const int full = (int)std::min<size_t>((total8 + 1023) / 1024, 4096);
  run("product: chunk, ids, U4, nt, grid-stride", gather_chunk<0, false, 4, true, 0>, full, true);
  run("  no table reads", gather_chunk<1, false, 4, true, 0>, full, false);
  run("  no stores", gather_chunk<2, false, 4, true, 0>, full, false);
  CK(hipMemset(d_out, 0, total8 * 32));
  run("chunk, LIVE mask", gather_chunk<0, true, 4, true, 0>, full, true);
  run("chunk, LIVE, plain stores", gather_chunk<0, true, 4, false, 0>, full, true);
  run("chunk, LIVE, U2", gather_chunk<0, true, 2, true, 0>, (int)std::min<size_t>((total8 + 511) / 512, 8192), true);
  run("chunk, LIVE, U8", gather_chunk<0, true, 8, true, 0>, (int)std::min<size_t>((total8 + 2047) / 2048, 4096), true);
  run("chunk, LIVE, U4, block-adjacent", gather_chunk<0, true, 4, true, 1>, (int)((total8 + 1023) / 1024), true);
  run("chunk, LIVE, U4, block-adjacent, 2048 blocks", gather_chunk<0, true, 4, true, 1>, 2048, true);
  run("chunk, LIVE, U4, block-adjacent, 1024 blocks", gather_chunk<0, true, 4, true, 1>, 1024, true);
  run("chunk, LIVE, U8, block-adjacent", gather_chunk<0, true, 8, true, 1>, (int)((total8 + 2047) / 2048), true);
  run("  no table reads (LIVE, adjacent)", gather_chunk<1, true, 4, true, 1>, (int)((total8 + 1023) / 1024), false);
  run("  no stores (LIVE, adjacent)", gather_chunk<2, true, 4, true, 1>, (int)((total8 + 1023) / 1024), false);
  CK(hipMemset(d_out, 0, total8 * 32));
  run("quad, U8", gather_quad<8, true>, (int)((total8 * 2 + 2047) / 2048), true);
  run("quad, U4", gather_quad<4, true>, (int)((total8 * 2 + 1023) / 1024), true);
  run("quad, U16", gather_quad<16, true>, (int)((total8 * 2 + 4095) / 4096), true);
  run("quad, U8, plain stores", gather_quad<8, false>, (int)((total8 * 2 + 2047) / 2048), true);
  run("quad, U8, 2048 blocks", gather_quad<8, true>, 2048, true);
  run("chunk, LIVE, U2 (again)", gather_chunk<0, true, 2, true, 0>, (int)std::min<size_t>((total8 + 511) / 512, 8192), false);
  run("  ... without the dropout hash", gather_chunk<3, true, 2, true, 0>, (int)std::min<size_t>((total8 + 511) / 512, 8192), false);
  run("  ... without hash and split", gather_chunk<4, true, 2, true, 0>, (int)std::min<size_t>((total8 + 511) / 512, 8192), false);
  run("  ... without hash and split, U4", gather_chunk<4, true, 4, true, 0>, (int)std::min<size_t>((total8 + 1023) / 1024, 8192), false);
  CK(hipMemset(d_out, 0, total8 * 32));
  run("v3 pipelined, 2048 blocks", gather_v3<true>, 2048, true);
  run("v3 pipelined, 1536 blocks", gather_v3<true>, 1536, true);
  run("v3 pipelined, 1024 blocks", gather_v3<true>, 1024, true);
  run("v3 pipelined, 4096 blocks", gather_v3<true>, 4096, true);
  run("v3 pipelined, 2048 blocks, plain stores", gather_v3<false>, 2048, true);
  Args2 g2{a, fast_div(C8), fast_div(Tp), 1.25f * 32768.f};
  auto run2 = [&](const char* name, auto kern, int blocks) {
    run_l(name, [&] { hipExtLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, s, e0, e1, 0, g2); }, blocks, true);
  };
  run2("v2, U2", gather_v2<2, true>, (int)std::min<size_t>((total8 + 511) / 512, 8192));
  run2("v2, U1", gather_v2<1, true>, (int)((total8 + 255) / 256));
  run2("v2, U2, 4096 blocks", gather_v2<2, true>, 4096);
  run2("v2, U3", gather_v2<3, true>, (int)((total8 + 767) / 768));
  run2("v2, U4", gather_v2<4, true>, (int)std::min<size_t>((total8 + 1023) / 1024, 4096));
  run2("v2, U2, plain stores", gather_v2<2, false>, (int)std::min<size_t>((total8 + 511) / 512, 8192));
  run("chunk, LIVE, U1", gather_chunk<0, true, 1, true, 0>, (int)((total8 + 255) / 256), true);
  run("chunk, LIVE, U2, block-adjacent", gather_chunk<0, true, 2, true, 1>, (int)((total8 + 511) / 512), true);
  run("rows list, U1", gather_rows<1, true>, (int)((n_chunks + 255) / 256), true, d_rows, n_chunks);
  run("rows list, U2", gather_rows<2, true>, (int)((n_chunks + 511) / 512), true, d_rows, n_chunks);
  run("rows list, U4", gather_rows<4, true>, (int)((n_chunks + 1023) / 1024), true, d_rows, n_chunks);
  run("rows list, U2, plain stores", gather_rows<2, false>, (int)((n_chunks + 511) / 512), true, d_rows, n_chunks);
  run("rows list, U2, 2048 blocks", gather_rows<2, true>, 2048, true, d_rows, n_chunks);
  run("rows list, U4, 1024 blocks", gather_rows<4, true>, 1024, true, d_rows, n_chunks);
  return 0;
}
