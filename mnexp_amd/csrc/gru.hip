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
//
// Element-wise traffic of the recurrence (gate inputs in, saved activations out, gate gradients out) moves
// in the MFMA ACCUMULATOR ("fragment") layout: tape[blk][t][tile][lane][4] holds, for lane (c, rq), the 4
// sample rows rq*4..rq*4+3 of hidden column tile*16+c -- exactly one accumulator register quad.  Every
// such access is ONE 16-byte, fully coalesced load/store per tile (1 KiB per wave instruction) instead of 4
// scattered dwords with their own address arithmetic; the epilogues were VALU-issue bound on that arithmetic.
// Two fully parallel layout kernels convert between the row-major tensors of the rest of the step and the
// tape (xw -> fragments before the forward recurrence; gate gradients / r*h / h_prev -> row-major after BPTT).
#include <stdlib.h>

#include "lstur_device.h"
#include "lstur_internal.h"

namespace lstur {

constexpr int GW = 16;  // waves per workgroup

// ring depth for a wave that owns C column tiles: ~8 float4 fragment loads in flight per wave
__host__ __device__ constexpr int ring_depth(int C) { return C >= 4 ? 2 : (C == 3 ? 3 : (C == 2 ? 4 : 6)); }

// Register ring of packed-B fragments for C tiles (first, first+GW, ...) of one operand image.
template <int C>
struct FragRing {
  static constexpr int D = ring_depth(C);
  float4 r[D][C];
  __amdgpu_buffer_rsrc_t rs;
  uint32_t voff;      // per-lane byte offset inside a 16x16 fragment block
  int chstride;       // floats per K chunk of the image
  int first;          // first tile of this wave
  __device__ __forceinline__ void bind(const float* img, int np, int nchunk, int first_tile, int lane,
                                       bool nostream = false) {
    chstride = np * 16;
    // nostream: timing-only build with a zero-record descriptor (every fragment load returns 0 at once)
    rs = make_rsrc(img, nostream ? 0u : (uint32_t)nchunk * (uint32_t)chstride * 4u);
    voff = ((lane & 15) * 16 + (lane >> 4) * 4) * 4;
    first = first_tile;
  }
  // scalar byte offset of (chunk, tile j); reads past the image return 0 (buffer range check)
  __device__ __forceinline__ uint32_t soff(int ch, int j) const {
    return (uint32_t)(ch * chstride + (first + j * GW) * 256) * 4u;
  }
  __device__ __forceinline__ void prefetch() {   // chunks 0..D-1
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
      for (int j = 0; j < C; ++j) r[d][j] = buf_load4(rs, voff, soff(d, j));
  }
};

// acc[j] = A_lds[16 x K] . image[:, tile j]; the ring must hold chunks 0..D-1 on entry (prefetch()).
// Branch-free MFMA stream: tile count and ring depth are template parameters.  NCH > 0 fixes the number
// of K chunks at compile time: the chunk loop is then straight-line code (static ring slots, loads and LDS
// reads scheduled ahead of the MFMAs with counted waits); NCH == 0 keeps a runtime loop for other widths.
template <int C, int NCH>
__device__ __forceinline__ void gemm_run(FragRing<C>& ring, const float* __restrict__ arow, int nchunk, f32x4* acc) {
  constexpr int D = FragRing<C>::D;
#pragma unroll
  for (int j = 0; j < C; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if constexpr (NCH > 0) {
    float4 a = *(const float4*)(arow);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int d = ch % D;
      float4 an = a;
      if (ch + 1 < NCH) an = *(const float4*)(arow + (ch + 1) * 16);
      float4 b[C];
#pragma unroll
      for (int j = 0; j < C; ++j) b[j] = ring.r[d][j];
      if (ch + D < NCH) {
#pragma unroll
        for (int j = 0; j < C; ++j) ring.r[d][j] = buf_load4(ring.rs, ring.voff, ring.soff(ch + D, j));
      }
#pragma unroll
      for (int j = 0; j < C; ++j) acc[j] = mfma16(a.x, b[j].x, acc[j]);
#pragma unroll
      for (int j = 0; j < C; ++j) acc[j] = mfma16(a.y, b[j].y, acc[j]);
#pragma unroll
      for (int j = 0; j < C; ++j) acc[j] = mfma16(a.z, b[j].z, acc[j]);
#pragma unroll
      for (int j = 0; j < C; ++j) acc[j] = mfma16(a.w, b[j].w, acc[j]);
      a = an;
    }
  } else {
    for (int ch0 = 0; ch0 < nchunk; ch0 += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int ch = ch0 + d;
        if (ch < nchunk) {
          float4 a = *(const float4*)(arow + ch * 16);
          float4 b[C];
#pragma unroll
          for (int j = 0; j < C; ++j) b[j] = ring.r[d][j];
#pragma unroll
          for (int j = 0; j < C; ++j) ring.r[d][j] = buf_load4(ring.rs, ring.voff, ring.soff(ch + D, j));
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
}

// ---- fragment tape ------------------------------------------------------------------------------------
// sections of S = nb * W * NT * 256 floats (nb = ceil(n/16) sample blocks, NT = Hp/16 hidden tiles)
enum { TP_XZ = 0, TP_XR, TP_XH, TP_Z, TP_R, TP_HH, TP_HP, TP_DZ, TP_DR, TP_DH, TP_RH, TP_SECTIONS };
static size_t generic_tape_floats(int n, int W, int H) {
  return (size_t)TP_SECTIONS * cdiv(n, 16) * W * (ceil16(H) / 16) * 256;
}
__host__ __device__ __forceinline__ size_t tape_section(int n, int W, int H) {
  return (size_t)cdiv(n, 16) * W * (ceil16(H) / 16) * 256;
}

// xw [n*W, 3H] (row-major, gates z|r|h) -> tape sections XZ, XR, XH.  One thread per fragment float4.
__global__ void gru_xw_to_tape_kernel(const float* __restrict__ xw, float* __restrict__ tape, int n, int W, int H,
                                      size_t S) {
  const int NT = ceil16(H) / 16;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over 3 * S / 4
  if (i >= 3 * S / 4) return;
  const int gate = (int)(i / (S / 4));
  size_t r = i - (size_t)gate * (S / 4);
  const int lane = (int)(r & 63);
  r >>= 6;
  const int tile = (int)(r % NT);
  r /= NT;
  const int t = (int)(r % W), blk = (int)(r / W);
  const int col = tile * 16 + (lane & 15);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col < H) {
    const int b = blk * 16 + (lane >> 4) * 4;
    const size_t base = ((size_t)b * W + t) * 3 * H + (size_t)gate * H + col, rs = (size_t)W * 3 * H;
    if (b + 0 < n) v.x = xw[base];
    if (b + 1 < n) v.y = xw[base + rs];
    if (b + 2 < n) v.z = xw[base + 2 * rs];
    if (b + 3 < n) v.w = xw[base + 3 * rs];
  }
  ((float4*)tape)[(size_t)(TP_XZ + gate) * (S / 4) + (i - (size_t)gate * (S / 4))] = v;
}

// tape sections DZ, DR, DH, RH, HP -> dgates [n*W, 3H], rh [n*W, H], hprev [n*W, H] (row-major)
__global__ void gru_tape_to_rows_kernel(const float* __restrict__ tape, float* __restrict__ dgates,
                                        float* __restrict__ rh, float* __restrict__ hprev, int n, int W, int H,
                                        size_t S) {
  const int NT = ceil16(H) / 16;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over S / 4 fragment quads
  if (i >= S / 4) return;
  size_t r = i;
  const int lane = (int)(r & 63);
  r >>= 6;
  const int tile = (int)(r % NT);
  r /= NT;
  const int t = (int)(r % W), blk = (int)(r / W);
  const int col = tile * 16 + (lane & 15);
  if (col >= H) return;
  const float4* tp = (const float4*)tape;
  const float4 dz = tp[(size_t)TP_DZ * (S / 4) + i], dr = tp[(size_t)TP_DR * (S / 4) + i],
               dh = tp[(size_t)TP_DH * (S / 4) + i], rhv = tp[(size_t)TP_RH * (S / 4) + i],
               hp = tp[(size_t)TP_HP * (S / 4) + i];
  const int b = blk * 16 + (lane >> 4) * 4;
  const float dzv[4] = {dz.x, dz.y, dz.z, dz.w}, drv[4] = {dr.x, dr.y, dr.z, dr.w}, dhv[4] = {dh.x, dh.y, dh.z, dh.w},
              rv[4] = {rhv.x, rhv.y, rhv.z, rhv.w}, pv[4] = {hp.x, hp.y, hp.z, hp.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (b + j < n) {
      const size_t row = (size_t)(b + j) * W + t;
      dgates[row * 3 * H + col] = dzv[j];
      dgates[row * 3 * H + H + col] = drv[j];
      dgates[row * 3 * H + 2 * H + col] = dhv[j];
      rh[row * H + col] = rv[j];
      hprev[row * H + col] = pv[j];
    }
  }
}

struct GruFwdArgs {
  const uint8_t* gmask; const float* h0; const float* u_zr; const float* u_h; float* tape; float* hlast;
  int n, W, H, dbg;
};

// Gate columns of the z|r operand image are padded per gate to Hp = ceil16(H) (N = 2*Hp: z at [0,H), r at
// [Hp, Hp+H)), so no 16-column tile straddles two gates and every epilogue is uniform per tile.
//
// Whole time loop for a wave that owns C1 tiles of the z|r product and C2 tiles of the candidate product.
template <int C1, int C2, int NCH>
__device__ __forceinline__ void gru_fwd_body(const GruFwdArgs& g, float* smem, int wv, int lane) {
  const int H = g.H, W = g.W, n = g.n;
  const int nch = cdiv(H, 16), Hp = nch * 16, LDH = Hp + 4, NT = nch;
  float* hL = smem;
  float* rhL = hL + 16 * LDH;
  float* zL = rhL + 16 * LDH;
  const int c = lane & 15, rq = lane >> 4, kq = lane >> 4;
  const bool nostream = (g.dbg & 1) != 0;
  const size_t S = tape_section(n, W, H);

  constexpr int C2R = C2 > 0 ? C2 : 1;      // a wave without candidate tiles keeps a dummy ring (never run)
  FragRing<C1> ring1;
  FragRing<C2R> ring2;
  ring1.bind(g.u_zr, 2 * Hp, nch, wv, lane, nostream);
  ring2.bind(g.u_h, Hp, nch, wv, lane, nostream);
  ring1.prefetch();

  // tiles of this wave: z|r product tile T1 = wv + i*GW in [0, 2*NT): gate = T1 >= NT, hidden tile = T1 - gate*NT
  bool isr[C1], ok1[C1], ok2[C2R];
  int ht1[C1], l1[C1], l2[C2R];
#pragma unroll
  for (int i = 0; i < C1; ++i) {
    const int T1 = wv + i * GW;
    isr[i] = T1 >= NT;                                        // wave-uniform
    ht1[i] = T1 - (isr[i] ? NT : 0);
    const int hc = ht1[i] * 16 + c;
    ok1[i] = hc < H;                                          // lanes past H: LDS padding columns stay 0
    l1[i] = rq * 4 * LDH + min(hc, Hp - 1);
  }
#pragma unroll
  for (int i = 0; i < C2R; ++i) {
    const int hc = (wv + i * GW) * 16 + c;
    ok2[i] = C2 > 0 && hc < H;
    l2[i] = rq * 4 * LDH + min(hc, Hp - 1);
  }
  // mask rows of this lane (clamped: rows past n are never read back)
  uint32_t mo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) mo[j] = (uint32_t)min((int)blockIdx.x * 16 + rq * 4 + j, n - 1) * (uint32_t)W;

  float4* tp = (float4*)g.tape;
  const size_t S4 = S / 4;
  const float* a1 = hL + c * LDH + kq * 4;
  const float* a2 = rhL + c * LDH + kq * 4;
  f32x4 acc1[C1], acc2[C2R];
  float4 x1[C1], x2[C2R];
  float mk[4];
  unsigned long long c0 = 0, r0 = 0;    // diagnostic build only (dbg bit 4): in-kernel clock
  if (g.dbg & 16) {
    c0 = __builtin_amdgcn_s_memtime();
    r0 = __builtin_amdgcn_s_memrealtime();
  }
  unsigned long long st[6] = {0, 0, 0, 0, 0, 0};   // diagnostic build only (dbg bit 5): per-segment shader cycles
  const bool stamp = (g.dbg & 32) != 0;
#define LSTUR_STAMP(k)                                     \
  if (stamp) {                                             \
    unsigned long long now = __builtin_amdgcn_s_memtime(); \
    st[k] += now - last;                                   \
    last = now;                                            \
  }
  for (int t = 0; t < W; ++t) {
    unsigned long long last = stamp ? __builtin_amdgcn_s_memtime() : 0;
    // fragment quad index of (this block, step t, hidden tile 0) + this lane
    const size_t f0 = ((size_t)blockIdx.x * W + t) * NT * 64 + lane;
    // ---- phase 1: z | r ----
#pragma unroll
    for (int i = 0; i < C1; ++i) x1[i] = tp[(size_t)(isr[i] ? TP_XR : TP_XZ) * S4 + f0 + (size_t)ht1[i] * 64];
    gemm_run<C1, NCH>(ring1, a1, nch, acc1);
    LSTUR_STAMP(0)
    if (C2 > 0) ring2.prefetch();     // candidate-product fragments fly during the gate epilogue + barrier
#pragma unroll
    for (int i = 0; i < C1; ++i) {
      float4 gt;
      gt.x = hard_sigmoid(acc1[i][0] + x1[i].x);
      gt.y = hard_sigmoid(acc1[i][1] + x1[i].y);
      gt.z = hard_sigmoid(acc1[i][2] + x1[i].z);
      gt.w = hard_sigmoid(acc1[i][3] + x1[i].w);
      tp[(size_t)(isr[i] ? TP_R : TP_Z) * S4 + f0 + (size_t)ht1[i] * 64] = gt;
      if (ok1[i]) {
        if (!isr[i]) {
          zL[l1[i]] = gt.x;
          zL[l1[i] + LDH] = gt.y;
          zL[l1[i] + 2 * LDH] = gt.z;
          zL[l1[i] + 3 * LDH] = gt.w;
        } else {
          rhL[l1[i]] = gt.x * hL[l1[i]];
          rhL[l1[i] + LDH] = gt.y * hL[l1[i] + LDH];
          rhL[l1[i] + 2 * LDH] = gt.z * hL[l1[i] + 2 * LDH];
          rhL[l1[i] + 3 * LDH] = gt.w * hL[l1[i] + 3 * LDH];
        }
      }
    }
    // candidate inputs + mask for phase 2 (issued before the barrier so they fly during it)
    if (C2 > 0) {
#pragma unroll
      for (int i = 0; i < C2R; ++i) x2[i] = tp[(size_t)TP_XH * S4 + f0 + (size_t)(wv + i * GW) * 64];
#pragma unroll
      for (int j = 0; j < 4; ++j) mk[j] = g.gmask[mo[j] + t] ? 1.f : 0.f;
    }
    LSTUR_STAMP(1)
    __syncthreads();
    LSTUR_STAMP(2)
    // ---- phase 2: candidate + state update ----
    if (C2 > 0) gemm_run<C2R, NCH>(ring2, a2, nch, acc2);
    LSTUR_STAMP(3)
    if (t + 1 < W) ring1.prefetch();  // next step's z|r fragments fly during the update + barrier
    if (C2 > 0) {
#pragma unroll
      for (int i = 0; i < C2R; ++i) {
        const float xv[4] = {x2[i].x, x2[i].y, x2[i].z, x2[i].w};
        float hh[4], hp[4], ho[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          hh[j] = tanhf(acc2[i][j] + xv[j]);
          float z = zL[l2[i] + j * LDH];
          hp[j] = hL[l2[i] + j * LDH];
          float hn = z * hp[j] + (1.0f - z) * hh[j];
          ho[j] = mk[j] != 0.f ? hn : hp[j];  // K.rnn: masked step carries the previous state/output
        }
        const size_t fi = f0 + (size_t)(wv + i * GW) * 64;
        tp[(size_t)TP_HH * S4 + fi] = make_float4(hh[0], hh[1], hh[2], hh[3]);
        tp[(size_t)TP_HP * S4 + fi] = make_float4(hp[0], hp[1], hp[2], hp[3]);
        if (ok2[i]) {
#pragma unroll
          for (int j = 0; j < 4; ++j) hL[l2[i] + j * LDH] = ho[j];
        }
      }
    }
    LSTUR_STAMP(4)
    __syncthreads();
    LSTUR_STAMP(5)
  }
#undef LSTUR_STAMP
  if (stamp && lane == 0 && blockIdx.x == 0)
    for (int k = 0; k < 6; ++k) g.tape[(size_t)TP_DZ * S + wv * 8 + k] = (float)st[k];
  if ((g.dbg & 16) && wv == 0 && lane == 0 && blockIdx.x == 0) {
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    g.tape[(size_t)TP_DR * S] = (float)(c1 - c0);       // shader cycles
    g.tape[(size_t)TP_DR * S + 1] = (float)(r1 - r0);   // 100 MHz ticks
  }
}

// number of tiles wave wv owns out of ntiles dealt round-robin over GW waves
__device__ __forceinline__ int tiles_of(int wv, int ntiles) { return wv < ntiles ? (ntiles - wv + GW - 1) / GW : 0; }

__global__ __launch_bounds__(GW * 64) void gru_fwd_kernel(GruFwdArgs g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = g.H, W = g.W, n = g.n;
  const int nch = cdiv(H, 16), Hp = nch * 16, LDH = Hp + 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform -> scalar addressing of fragments
  const int b0 = blockIdx.x * 16;
  float* hL = smem;
  for (int i = tid; i < 3 * 16 * LDH; i += GW * 64) smem[i] = 0.f;
  __syncthreads();
  for (int i = tid; i < 16 * H; i += GW * 64) {
    int row = i / H, col = i - row * H, b = b0 + row;
    hL[row * LDH + col] = (g.h0 && b < n) ? g.h0[(size_t)b * H + col] : 0.f;
  }
  __syncthreads();
  // static (tile-count) specialisation per wave: every wave runs the same number of barriers
  const int c1 = tiles_of(wv, 2 * Hp / 16), c2 = tiles_of(wv, Hp / 16);
  bool done = false;
#define LSTUR_FWD_CASE(A, B)                                      \
  if (!done && c1 == A && c2 == B) {                              \
    if (nch == 13)                                                \
      gru_fwd_body<A, B, 13>(g, smem, wv, lane);                  \
    else                                                          \
      gru_fwd_body<A, B, 0>(g, smem, wv, lane);                   \
    done = true;                                                  \
  }
  if constexpr (GW == 8) {   // H <= 256: up to 4 z|r tiles and 2 candidate tiles per wave
    LSTUR_FWD_CASE(4, 2) LSTUR_FWD_CASE(3, 2) LSTUR_FWD_CASE(3, 1) LSTUR_FWD_CASE(4, 1) LSTUR_FWD_CASE(3, 0)
    LSTUR_FWD_CASE(4, 0) LSTUR_FWD_CASE(2, 2) LSTUR_FWD_CASE(1, 2)
  }
  LSTUR_FWD_CASE(2, 1) LSTUR_FWD_CASE(1, 1) LSTUR_FWD_CASE(1, 0) LSTUR_FWD_CASE(2, 0)
#undef LSTUR_FWD_CASE
  if (!done)   // a wave without tiles (tiny H) still has to meet the barriers
    for (int t = 0; t < W; ++t) {
      __syncthreads();
      __syncthreads();
    }
  // last state (the loop ends on a barrier, so hL is complete)
  for (int i = tid; i < 16 * H; i += GW * 64) {
    int row = i / H, col = i - row * H, b = b0 + row;
    if (b < n) g.hlast[(size_t)b * H + col] = hL[row * LDH + col];
  }
}

// ---------------------------------------------------------------------------------------------
// BPTT.  Per step (t = W-1 .. 0), with dh the gradient wrt the step's output state:
//   dhn = m ? dh : 0 ; carry = m ? 0 : dh
//   dpre_h = dhn (1-z)(1-hh^2) ; dpre_z = dhn (hp-hh) hs'(z)
//   drh = dpre_h U_h^T ; dpre_r = drh hp hs'(r)
//   dh_prev = carry + dhn z + drh r + [dpre_z | dpre_r] U_zr^T
// [dpre_z | dpre_r | dpre_h], r*hp (and hp, saved by the forward) go to the tape; gru_tape_to_rows_kernel
// turns them into the row-major operands of the weight-gradient GEMMs.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float hs_grad(float g) { return (g > 0.f && g < 1.f) ? 0.2f : 0.f; }

struct GruBwdArgs {
  const float* dh_last; const uint8_t* gmask; float* tape; const float* ut_zr; const float* ut_h; float* dh0;
  int n, W, H;
};

// Every wave owns the same C column tiles (of H hidden units) in all three phases: lane (c, rq) of tile
// `tile` holds rows rq*4+j, col tile*16+c -- so the element-wise work needs no extra LDS round trip.
template <int C, int NCH, int NCH2>
__device__ __forceinline__ void gru_bwd_body(const GruBwdArgs& g, float* smem, int wv, int lane) {
  const int H = g.H, W = g.W, n = g.n;
  const int nch = cdiv(H, 16), Hp = nch * 16, LDH = Hp + 4, NT = nch;
  const int nch2 = 2 * nch, LD2 = 2 * Hp + 4;      // K of the last product is gate-padded: z at [0,H), r at [Hp,Hp+H)
  float* dhL = smem;               // [16][LDH] gradient wrt current state
  float* nL = dhL + 16 * LDH;      // [16][LDH] partial dh_prev
  float* g2L = nL + 16 * LDH;      // [16][LDH] dpre_h
  float* g1L = g2L + 16 * LDH;     // [16][LD2] dpre_z | dpre_r
  const int c = lane & 15, rq = lane >> 4, kq = lane >> 4;
  const size_t S4 = tape_section(n, W, H) / 4;

  FragRing<C> ringh, ringzr;
  ringh.bind(g.ut_h, Hp, nch, wv, lane);
  ringzr.bind(g.ut_zr, Hp, nch2, wv, lane);
  ringh.prefetch();
  const float* a2 = g2L + c * LDH + kq * 4;
  const float* a1 = g1L + c * LD2 + kq * 4;

  bool ok[C];
  int l[C], lz[C];
#pragma unroll
  for (int i = 0; i < C; ++i) {
    const int hc = (wv + i * GW) * 16 + c;
    ok[i] = hc < H;
    l[i] = rq * 4 * LDH + min(hc, Hp - 1);
    lz[i] = rq * 4 * LD2 + min(hc, Hp - 1);
  }
  uint32_t mo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) mo[j] = (uint32_t)min((int)blockIdx.x * 16 + rq * 4 + j, n - 1) * (uint32_t)W;

  float4* tp = (float4*)g.tape;
  f32x4 acc[C];
  float4 zv[C], rv[C], hv[C], pv[C];
  float mk[4];
  for (int t = W - 1; t >= 0; --t) {
    const size_t f0 = ((size_t)blockIdx.x * W + t) * NT * 64 + lane;
    // ---- this step's saved activations for the elements this lane owns ----
#pragma unroll
    for (int j = 0; j < 4; ++j) mk[j] = g.gmask[mo[j] + t] ? 1.f : 0.f;
#pragma unroll
    for (int i = 0; i < C; ++i) {
      const size_t fi = f0 + (size_t)(wv + i * GW) * 64;
      zv[i] = tp[(size_t)TP_Z * S4 + fi];
      rv[i] = tp[(size_t)TP_R * S4 + fi];
      hv[i] = tp[(size_t)TP_HH * S4 + fi];
      pv[i] = tp[(size_t)TP_HP * S4 + fi];
    }
    // ---- (a) element-wise part that does not need drh ----
#pragma unroll
    for (int i = 0; i < C; ++i) {
      const float z[4] = {zv[i].x, zv[i].y, zv[i].z, zv[i].w}, hh[4] = {hv[i].x, hv[i].y, hv[i].z, hv[i].w},
                  hp[4] = {pv[i].x, pv[i].y, pv[i].z, pv[i].w}, r[4] = {rv[i].x, rv[i].y, rv[i].z, rv[i].w};
      float dpz[4], dph[4], rhp[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float dh = dhL[l[i] + j * LDH];
        float dhn = mk[j] != 0.f ? dh : 0.f;
        dph[j] = dhn * (1.0f - z[j]) * (1.0f - hh[j] * hh[j]);
        dpz[j] = dhn * (hp[j] - hh[j]) * hs_grad(z[j]);
        rhp[j] = r[j] * hp[j];
        if (ok[i]) {
          g2L[l[i] + j * LDH] = dph[j];
          g1L[lz[i] + j * LD2] = dpz[j];
          nL[l[i] + j * LDH] = (mk[j] != 0.f ? 0.f : dh) + dhn * z[j];
        }
      }
      const size_t fi = f0 + (size_t)(wv + i * GW) * 64;
      tp[(size_t)TP_DZ * S4 + fi] = make_float4(dpz[0], dpz[1], dpz[2], dpz[3]);
      tp[(size_t)TP_DH * S4 + fi] = make_float4(dph[0], dph[1], dph[2], dph[3]);
      tp[(size_t)TP_RH * S4 + fi] = make_float4(rhp[0], rhp[1], rhp[2], rhp[3]);
    }
    __syncthreads();
    // ---- (b) drh = dpre_h . U_h^T ----
    gemm_run<C, NCH>(ringh, a2, nch, acc);
    ringzr.prefetch();
#pragma unroll
    for (int i = 0; i < C; ++i) {
      const float hp[4] = {pv[i].x, pv[i].y, pv[i].z, pv[i].w}, r[4] = {rv[i].x, rv[i].y, rv[i].z, rv[i].w};
      float dpr[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float drh = acc[i][j];
        dpr[j] = drh * hp[j] * hs_grad(r[j]);
        if (ok[i]) {
          g1L[lz[i] + j * LD2 + Hp] = dpr[j];
          nL[l[i] + j * LDH] += drh * r[j];
        }
      }
      tp[(size_t)TP_DR * S4 + f0 + (size_t)(wv + i * GW) * 64] = make_float4(dpr[0], dpr[1], dpr[2], dpr[3]);
    }
    __syncthreads();
    // ---- (c) dh_prev = part + [dpre_z | dpre_r] . U_zr^T ----
    gemm_run<C, NCH2>(ringzr, a1, nch2, acc);
    if (t > 0) ringh.prefetch();
#pragma unroll
    for (int i = 0; i < C; ++i) {
      if (ok[i]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) dhL[l[i] + j * LDH] = nL[l[i] + j * LDH] + acc[i][j];
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(GW * 64) void gru_bwd_kernel(GruBwdArgs g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = g.H, W = g.W, n = g.n;
  const int nch = cdiv(H, 16), Hp = nch * 16, LDH = Hp + 4;
  const int LD2 = 2 * Hp + 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b0 = blockIdx.x * 16;
  float* dhL = smem;
  for (int i = tid; i < 3 * 16 * LDH + 16 * LD2; i += GW * 64) smem[i] = 0.f;
  __syncthreads();
  for (int i = tid; i < 16 * H; i += GW * 64) {
    int row = i / H, col = i - row * H, b = b0 + row;
    dhL[row * LDH + col] = b < n ? g.dh_last[(size_t)b * H + col] : 0.f;
  }
  __syncthreads();
  const int cnt = tiles_of(wv, Hp / 16);
  if (GW == 8 && cnt == 2 && nch == 13)
    gru_bwd_body<(GW == 8 ? 2 : 1), 13, 26>(g, smem, wv, lane);
  else if (cnt == 1 && nch == 13)
    gru_bwd_body<1, 13, 26>(g, smem, wv, lane);
  else if (GW == 8 && cnt == 2)
    gru_bwd_body<(GW == 8 ? 2 : 1), 0, 0>(g, smem, wv, lane);
  else if (cnt == 1)
    gru_bwd_body<1, 0, 0>(g, smem, wv, lane);
  else
    for (int t = 0; t < W; ++t) {
      __syncthreads();
      __syncthreads();
      __syncthreads();
    }
  __syncthreads();
  for (int i = tid; i < 16 * H; i += GW * 64) {
    int row = i / H, col = i - row * H, b = b0 + row;
    if (b < n) g.dh0[(size_t)b * H + col] = dhL[row * LDH + col];
  }
}

// ---------------------------------------------------------------------------------------------
// All four operand images of the recurrent matrix U [H, 3H] (z|r|h) in one launch:
//   u_zr  [nch][2Hp][16]  B = [U_z | 0 | U_r | 0]            (forward z|r product, gates padded to Hp)
//   u_h   [nch][Hp][16]   B = U_h                            (forward candidate product)
//   ut_h  [nch][Hp][16]   B = U_h^T                          (BPTT: drh)
//   ut_zr [2nch][Hp][16]  B = [U_z^T ; 0 ; U_r^T ; 0]        (BPTT: dh_prev, K gate-padded)
// ---------------------------------------------------------------------------------------------
static size_t generic_pack_floats(int H) {
  int nch = cdiv(H, 16), Hp = nch * 16;
  return (size_t)nch * 16 * (2 * Hp + Hp + Hp + 2 * Hp);
}
__global__ void gru_pack_kernel(const float* __restrict__ U, int H, float* __restrict__ u_zr, float* __restrict__ u_h,
                                float* __restrict__ ut_h, float* __restrict__ ut_zr) {
  const int nch = cdiv(H, 16), Hp = nch * 16, H3 = 3 * H;
  const int e1 = nch * 2 * Hp * 16, e2 = nch * Hp * 16;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < e1) {   // u_zr: P[ch][n][kk], n in [0, 2Hp)
    int kk = i & 15, nn = (i >> 4) % (2 * Hp), ch = (i >> 4) / (2 * Hp), k = ch * 16 + kk;
    int gate = nn >= Hp, hc = nn - gate * Hp;
    u_zr[i] = (k < H && hc < H) ? U[(size_t)k * H3 + gate * H + hc] : 0.f;
    return;
  }
  i -= e1;
  if (i < e2) {   // u_h
    int kk = i & 15, nn = (i >> 4) % Hp, ch = (i >> 4) / Hp, k = ch * 16 + kk;
    u_h[i] = (k < H && nn < H) ? U[(size_t)k * H3 + 2 * H + nn] : 0.f;
    return;
  }
  i -= e2;
  if (i < e2) {   // ut_h: B[k][n] = U_h[n][k]
    int kk = i & 15, nn = (i >> 4) % Hp, ch = (i >> 4) / Hp, k = ch * 16 + kk;
    ut_h[i] = (k < H && nn < H) ? U[(size_t)nn * H3 + 2 * H + k] : 0.f;
    return;
  }
  i -= e2;
  if (i < 2 * e2) {   // ut_zr: K = 2Hp gate-padded: B[k][n] = U[n][gate*H + (k - gate*Hp)]
    int kk = i & 15, nn = (i >> 4) % Hp, ch = (i >> 4) / Hp, k = ch * 16 + kk;
    int gate = k >= Hp, hc = k - gate * Hp;
    ut_zr[i] = (hc < H && nn < H) ? U[(size_t)nn * H3 + gate * H + hc] : 0.f;
  }
}
static void generic_pack(const float* U, int H, float* u_zr, float* u_h, float* ut_h, float* ut_zr, hipStream_t s) {
  size_t total = generic_pack_floats(H);
  hipLaunchKernelGGL(gru_pack_kernel, dim3((unsigned)cdivz(total, 256)), dim3(256), 0, s, U, H, u_zr, u_h, ut_h, ut_zr);
}

static void set_lds_attr(const void* fn, size_t lds, size_t* cur) {
  if (lds > 65536 && lds > *cur) {
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    *cur = lds;
  }
}

static void generic_fwd(const float* xw, const uint8_t* gmask, const float* h0, const float* u_zr_packed,
                        const float* u_h_packed, float* tape, float* hlast, int n, int W, int H, hipStream_t s) {
  if (!n) return;
  int Hp = ceil16(H), LDH = Hp + 4;
  size_t lds = (size_t)3 * 16 * LDH * sizeof(float);
  static size_t cur = 0;
  set_lds_attr((const void*)gru_fwd_kernel, lds, &cur);
#ifdef LSTUR_ABLATE
  static const int dbg = getenv("LSTUR_GRU_DEBUG") ? atoi(getenv("LSTUR_GRU_DEBUG")) : 0;  // measurement knobs (bench-only library)
#else
  constexpr int dbg = 0;
#endif
  const size_t S = tape_section(n, W, H);
  hipLaunchKernelGGL(gru_xw_to_tape_kernel, dim3((unsigned)cdivz(3 * S / 4, 256)), dim3(256), 0, s, xw, tape, n, W, H, S);
  GruFwdArgs a{gmask, h0, u_zr_packed, u_h_packed, tape, hlast, n, W, H, dbg};
  hipLaunchKernelGGL(gru_fwd_kernel, dim3(cdiv(n, 16)), dim3(GW * 64), lds, s, a);
}

static void generic_bwd(const float* dh_last, const uint8_t* gmask, float* tape, const float* ut_zr_packed,
                        const float* ut_h_packed, float* dgates, float* rh, float* hprev, float* dh0, int n, int W, int H,
                        hipStream_t s) {
  if (!n) return;
  int Hp = ceil16(H), LDH = Hp + 4, LD2 = 2 * Hp + 4;
  size_t lds = (size_t)(3 * 16 * LDH + 16 * LD2) * sizeof(float);
  static size_t cur = 0;
  set_lds_attr((const void*)gru_bwd_kernel, lds, &cur);
  GruBwdArgs a{dh_last, gmask, tape, ut_zr_packed, ut_h_packed, dh0, n, W, H};
  hipLaunchKernelGGL(gru_bwd_kernel, dim3(cdiv(n, 16)), dim3(GW * 64), lds, s, a);
  const size_t S = tape_section(n, W, H);
  hipLaunchKernelGGL(gru_tape_to_rows_kernel, dim3((unsigned)cdivz(S / 4, 256)), dim3(256), 0, s, tape, dgates, rh, hprev,
                     n, W, H, S);
}

// ---- public launchers: fast path (gru4.hip, U resident on chip, H <= 200) or the general kernels above ----------
// `packed` is one opaque buffer of gru_pack_floats(H) floats written by launch_gru_pack.
size_t gru_pack_floats(int H) { return gru4_supported(H) ? gru4_pack_floats(H) : generic_pack_floats(H); }
size_t gru_tape_floats(int n, int W, int H) {
  return gru4_supported(H) ? gru4_tape_floats(n, W, H) : generic_tape_floats(n, W, H);
}
void launch_gru_pack(const float* U, int H, float* packed, hipStream_t s, bool f16, int seq) {
  if (gru4_supported(H)) return launch_gru4_pack(U, H, packed, s, f16, seq);
  const size_t sixth = generic_pack_floats(H) / 6;   // u_zr 2/6 | u_h 1/6 | ut_h 1/6 | ut_zr 2/6
  generic_pack(U, H, packed, packed + 2 * sixth, packed + 3 * sixth, packed + 4 * sixth, s);
}
void launch_gru_fwd(const float* xw, const uint8_t* gmask, const float* h0, const float* packed, float* tape,
                    float* hlast, int n, int W, int H, hipStream_t s, bool f16, const GruScore* sc) {
  if (!n) return;
  if (gru4_supported(H)) return launch_gru4_fwd(xw, gmask, h0, packed, tape, hlast, n, W, H, s, f16, sc);
  const size_t sixth = generic_pack_floats(H) / 6;
  generic_fwd(xw, gmask, h0, packed, packed + 2 * sixth, tape, hlast, n, W, H, s);
}
void launch_gru_bwd(const float* dh_last, const uint8_t* gmask, float* tape, const float* packed, float* dgates,
                    float* rh, float* hprev, float* dh0, int n, int W, int H, hipStream_t s) {
  if (!n) return;
  if (gru4_supported(H))
    return launch_gru4_bwd(dh_last, gmask, tape, packed, dgates, rh, hprev, dh0, n, W, H, s);
  // (the general kernels know the sample-major layout and the whole recurrence only: callers ask gru4_supported first)
  const size_t sixth = generic_pack_floats(H) / 6;
  generic_bwd(dh_last, gmask, tape, packed + 4 * sixth, packed + 3 * sixth, dgates, rh, hprev, dh0, n, W, H, s);
}
bool gru_score_fusable(int H, int K1) { return gru4_supported(H) && H <= 256 && K1 >= 1 && K1 <= 8; }
bool gru_tape_rows(float* tape, int n, int W, int H, float** hprev, float** rh) {
  if (!gru4_supported(H)) return false;
  gru4_tape_rows(tape, n, W, H, hprev, rh);
  return true;
}

}  // namespace lstur
