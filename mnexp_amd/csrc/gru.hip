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
#include <stdlib.h>

#include "lstur_device.h"
#include "lstur_internal.h"

namespace lstur {

constexpr int GW = 8;  // waves per workgroup

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

struct GruFwdArgs {
  const float* xw; const uint8_t* gmask; const float* h0; const float* u_zr; const float* u_h;
  float* hs; float* zs; float* rs; float* hhs; int n, W, H, dbg;
};

// Layout contract of the GRU kernels (see lstur_gru_pack_u / include/lstur.h):
//   * every per-sample buffer holds ceil16(n) sample rows (the last row tile is processed unmasked);
//   * gate columns are padded per gate to Hp = ceil16(H): the z|r image has N = 2*Hp columns (z at [0,H),
//     r at [Hp, Hp+H)), so no 16-column tile straddles two gates and every epilogue is uniform per tile.
//
// Whole time loop for a wave that owns C1 tiles of the z|r product and C2 tiles of the candidate product.
template <int C1, int C2, int NCH>
__device__ __forceinline__ void gru_fwd_body(const GruFwdArgs& g, float* smem, int wv, int lane) {
  const int H = g.H, W = g.W;
  const int nch = cdiv(H, 16), Hp = nch * 16, LDH = Hp + 4;
  float* hL = smem;
  float* rhL = hL + 16 * LDH;
  float* zL = rhL + 16 * LDH;
  const int c = lane & 15, rq = lane >> 4, kq = lane >> 4;
  const uint32_t rowb = blockIdx.x * 16 + rq * 4;            // first of this lane's 4 sample rows
  const uint32_t uW = W, uH = H, uH3 = 3 * H, WH = uW * uH, WH3 = uW * uH3, W1H = (uW + 1u) * uH;
  const bool nostream = (g.dbg & 1) != 0;

  constexpr int C2R = C2 > 0 ? C2 : 1;      // a wave without candidate tiles keeps a dummy ring (never run)
  FragRing<C1> ring1;
  FragRing<C2R> ring2;
  ring1.bind(g.u_zr, 2 * Hp, nch, wv, lane, nostream);
  ring2.bind(g.u_h, Hp, nch, wv, lane, nostream);
  ring1.prefetch();

  // per-lane invariants of the tiles this wave owns (hidden index clamped so every load is in range;
  // stores / LDS writes of lanes past H are predicated off -- padding columns of the state stay 0)
  bool isr[C1], ok1[C1], ok2[C2R];
  uint32_t xo1[C1], go1[C1], xo2[C2R], go2[C2R], ho2[C2R];
  int l1[C1], l2[C2R];
#pragma unroll
  for (int i = 0; i < C1; ++i) {
    const int tcol = (wv + i * GW) * 16;
    isr[i] = tcol >= Hp;                                      // wave-uniform: tile of r gates
    const int hc = tcol - (isr[i] ? Hp : 0) + c;
    ok1[i] = hc < H;
    const int hcc = min(hc, H - 1);
    xo1[i] = rowb * WH3 + (isr[i] ? uH : 0u) + (uint32_t)hcc;
    go1[i] = rowb * WH + (uint32_t)hcc;
    l1[i] = rq * 4 * LDH + hcc;
  }
#pragma unroll
  for (int i = 0; i < C2R; ++i) {
    const int hc = (wv + i * GW) * 16 + c;
    ok2[i] = C2 > 0 && hc < H;
    const int hcc = min(hc, H - 1);
    xo2[i] = rowb * WH3 + 2u * uH + (uint32_t)hcc;
    go2[i] = rowb * WH + (uint32_t)hcc;
    ho2[i] = rowb * W1H + (uint32_t)hcc;
    l2[i] = rq * 4 * LDH + hcc;
  }
  const uint32_t mo = rowb * uW;

  const float* a1 = hL + c * LDH + kq * 4;
  const float* a2 = rhL + c * LDH + kq * 4;
  f32x4 acc1[C1], acc2[C2R];
  float x1[C1][4], x2[C2R][4], mk[4];
  unsigned long long c0 = 0, r0 = 0;    // diagnostic build only (dbg bit 4): in-kernel clock
  if (g.dbg & 16) {
    c0 = __builtin_amdgcn_s_memtime();
    r0 = __builtin_amdgcn_s_memrealtime();
  }
  unsigned long long st[6] = {0, 0, 0, 0, 0, 0};   // diagnostic build only (dbg bit 5): per-segment shader cycles
  const bool stamp = (g.dbg & 32) != 0;
#define LSTUR_STAMP(k)                                   \
  if (stamp) {                                           \
    unsigned long long now = __builtin_amdgcn_s_memtime(); \
    st[k] += now - last;                                 \
    last = now;                                          \
  }
  for (int t = 0; t < W; ++t) {
    const uint32_t ut = t;
    const float* xw_t = g.xw + ut * uH3;
    unsigned long long last = stamp ? __builtin_amdgcn_s_memtime() : 0;
    // ---- phase 1: z | r ----
#pragma unroll
    for (int i = 0; i < C1; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) x1[i][j] = xw_t[xo1[i] + j * WH3];
    gemm_run<C1, NCH>(ring1, a1, nch, acc1);
    LSTUR_STAMP(0)
    if (C2 > 0) ring2.prefetch();     // candidate-product fragments fly during the gate epilogue + barrier
    float* zs_t = g.zs + ut * uH;
    float* rs_t = g.rs + ut * uH;
#pragma unroll
    for (int i = 0; i < C1; ++i) {
      if (ok1[i]) {
        if (!isr[i]) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float gt = hard_sigmoid(acc1[i][j] + x1[i][j]);
            zL[l1[i] + j * LDH] = gt;
            zs_t[go1[i] + j * WH] = gt;
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float gt = hard_sigmoid(acc1[i][j] + x1[i][j]);
            rs_t[go1[i] + j * WH] = gt;
            rhL[l1[i] + j * LDH] = gt * hL[l1[i] + j * LDH];
          }
        }
      }
    }
    // candidate inputs + mask for phase 2 (issued before the barrier so they fly during it)
    if (C2 > 0) {
#pragma unroll
      for (int i = 0; i < C2R; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) x2[i][j] = xw_t[xo2[i] + j * WH3];
#pragma unroll
      for (int j = 0; j < 4; ++j) mk[j] = g.gmask[mo + j * uW + ut] ? 1.f : 0.f;
    }
    LSTUR_STAMP(1)
    __syncthreads();
    LSTUR_STAMP(2)
    // ---- phase 2: candidate + state update ----
    if (C2 > 0) gemm_run<C2R, NCH>(ring2, a2, nch, acc2);
    LSTUR_STAMP(3)
    if (t + 1 < W) ring1.prefetch();  // next step's z|r fragments fly during the update + barrier
    if (C2 > 0) {
      float* hhs_t = g.hhs + ut * uH;
      float* hs_t1 = g.hs + (ut + 1u) * uH;
#pragma unroll
      for (int i = 0; i < C2R; ++i) {
        if (ok2[i]) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float hh = tanhf(acc2[i][j] + x2[i][j]);
            float z = zL[l2[i] + j * LDH], hp = hL[l2[i] + j * LDH];
            float hn = z * hp + (1.0f - z) * hh;
            float ho = mk[j] != 0.f ? hn : hp;  // K.rnn: masked step carries the previous state/output
            hhs_t[go2[i] + j * WH] = hh;
            hs_t1[ho2[i] + j * W1H] = ho;
            hL[l2[i] + j * LDH] = ho;
          }
        }
      }
    }
    LSTUR_STAMP(4)
    __syncthreads();
    LSTUR_STAMP(5)
  }
#undef LSTUR_STAMP
  if (stamp && lane == 0 && blockIdx.x == 0)
    for (int k = 0; k < 6; ++k) g.rs[wv * 8 + k] = (float)st[k];
  if ((g.dbg & 16) && wv == 0 && lane == 0 && blockIdx.x == 0) {
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    g.zs[0] = (float)(c1 - c0);   // shader cycles
    g.zs[1] = (float)(r1 - r0);   // 100 MHz ticks
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
    float v = (g.h0 && b < n) ? g.h0[(size_t)b * H + col] : 0.f;
    hL[row * LDH + col] = v;
    g.hs[(size_t)b * (W + 1) * H + col] = v;
  }
  __syncthreads();
  // static (tile-count) specialisation per wave: every wave runs the same number of barriers
  const int c1 = tiles_of(wv, 2 * Hp / 16), c2 = tiles_of(wv, Hp / 16);
#define LSTUR_FWD_CASE(A, B)                                              \
  if (c1 == A && c2 == B) {                                               \
    if (nch == 13) return gru_fwd_body<A, B, 13>(g, smem, wv, lane);      \
    return gru_fwd_body<A, B, 0>(g, smem, wv, lane);                      \
  }
  LSTUR_FWD_CASE(4, 2) LSTUR_FWD_CASE(3, 2) LSTUR_FWD_CASE(3, 1) LSTUR_FWD_CASE(4, 1) LSTUR_FWD_CASE(2, 1)
  LSTUR_FWD_CASE(1, 1) LSTUR_FWD_CASE(2, 2) LSTUR_FWD_CASE(1, 0) LSTUR_FWD_CASE(2, 0) LSTUR_FWD_CASE(3, 0)
  LSTUR_FWD_CASE(4, 0) LSTUR_FWD_CASE(1, 2)
#undef LSTUR_FWD_CASE
  // a wave without tiles (tiny H) still has to meet the barriers
  for (int t = 0; t < W; ++t) {
    __syncthreads();
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

struct GruBwdArgs {
  const float* dh_last; const uint8_t* gmask; const float* hs; const float* zs; const float* rs; const float* hhs;
  const float* ut_zr; const float* ut_h; float* dgates; float* rh; float* hprev; float* dh0; int n, W, H;
};

// Every wave owns the same C column tiles (of H hidden units) in all three phases: lane (c, rq) of tile
// `tile` holds rows rq*4+j, col tile*16+c -- so the element-wise work needs no extra LDS round trip.
template <int C, int NCH, int NCH2>
__device__ __forceinline__ void gru_bwd_body(const GruBwdArgs& g, float* smem, int wv, int lane) {
  const int H = g.H, W = g.W;
  const int nch = cdiv(H, 16), Hp = nch * 16, LDH = Hp + 4;
  const int nch2 = 2 * nch, LD2 = 2 * Hp + 4;      // K of the last product is gate-padded: z at [0,H), r at [Hp,Hp+H)
  float* dhL = smem;               // [16][LDH] gradient wrt current state
  float* nL = dhL + 16 * LDH;      // [16][LDH] partial dh_prev
  float* g2L = nL + 16 * LDH;      // [16][LDH] dpre_h
  float* g1L = g2L + 16 * LDH;     // [16][LD2] dpre_z | dpre_r
  const int c = lane & 15, rq = lane >> 4, kq = lane >> 4;
  const uint32_t rowb = blockIdx.x * 16 + rq * 4;
  const uint32_t uW = W, uH = H, uH3 = 3 * H, WH = uW * uH, WH3 = uW * uH3, W1H = (uW + 1u) * uH;

  FragRing<C> ringh, ringzr;
  ringh.bind(g.ut_h, Hp, nch, wv, lane);
  ringzr.bind(g.ut_zr, Hp, nch2, wv, lane);
  ringh.prefetch();
  const float* a2 = g2L + c * LDH + kq * 4;
  const float* a1 = g1L + c * LD2 + kq * 4;

  bool ok[C];
  uint32_t go[C], ho[C], dgo[C];
  int l[C], lz[C];
#pragma unroll
  for (int i = 0; i < C; ++i) {
    const int hc = (wv + i * GW) * 16 + c;
    ok[i] = hc < H;
    const int hcc = min(hc, H - 1);
    go[i] = rowb * WH + (uint32_t)hcc;
    ho[i] = rowb * W1H + (uint32_t)hcc;
    dgo[i] = rowb * WH3 + (uint32_t)hcc;
    l[i] = rq * 4 * LDH + hcc;
    lz[i] = rq * 4 * LD2 + hcc;
  }
  const uint32_t mo = rowb * uW;

  f32x4 acc[C];
  float zv[C][4], rv[C][4], hv[C][4], pv[C][4], mk[4];
  for (int t = W - 1; t >= 0; --t) {
    const uint32_t ut = t;
    const uint32_t tH = ut * uH;
    // ---- this step's saved activations for the elements this lane owns ----
#pragma unroll
    for (int j = 0; j < 4; ++j) mk[j] = g.gmask[mo + j * uW + ut] ? 1.f : 0.f;
#pragma unroll
    for (int i = 0; i < C; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        zv[i][j] = g.zs[go[i] + j * WH + tH];
        rv[i][j] = g.rs[go[i] + j * WH + tH];
        hv[i][j] = g.hhs[go[i] + j * WH + tH];
        pv[i][j] = g.hs[ho[i] + j * W1H + tH];
      }
    // ---- (a) element-wise part that does not need drh ----
    float* dg_t = g.dgates + ut * uH3;
#pragma unroll
    for (int i = 0; i < C; ++i) {
      if (ok[i]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float dh = dhL[l[i] + j * LDH];
          float dhn = mk[j] != 0.f ? dh : 0.f;
          float z = zv[i][j], hh = hv[i][j], hp = pv[i][j];
          float dph = dhn * (1.0f - z) * (1.0f - hh * hh);
          float dpz = dhn * (hp - hh) * hs_grad(z);
          g2L[l[i] + j * LDH] = dph;
          g1L[lz[i] + j * LD2] = dpz;
          nL[l[i] + j * LDH] = (mk[j] != 0.f ? 0.f : dh) + dhn * z;
          dg_t[dgo[i] + j * WH3] = dpz;
          dg_t[dgo[i] + j * WH3 + 2u * uH] = dph;
          g.hprev[go[i] + j * WH + tH] = hp;
          g.rh[go[i] + j * WH + tH] = rv[i][j] * hp;
        }
      }
    }
    __syncthreads();
    // ---- (b) drh = dpre_h . U_h^T ----
    gemm_run<C, NCH>(ringh, a2, nch, acc);
    ringzr.prefetch();
#pragma unroll
    for (int i = 0; i < C; ++i) {
      if (ok[i]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float drh = acc[i][j];
          float r = rv[i][j], hp = pv[i][j];
          float dpr = drh * hp * hs_grad(r);
          g1L[lz[i] + j * LD2 + Hp] = dpr;
          nL[l[i] + j * LDH] += drh * r;
          dg_t[dgo[i] + j * WH3 + uH] = dpr;
        }
      }
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
  if (cnt == 2 && nch == 13)
    gru_bwd_body<2, 13, 26>(g, smem, wv, lane);
  else if (cnt == 1 && nch == 13)
    gru_bwd_body<1, 13, 26>(g, smem, wv, lane);
  else if (cnt == 2)
    gru_bwd_body<2, 0, 0>(g, smem, wv, lane);
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
size_t gru_pack_floats(int H) {
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
void launch_gru_pack(const float* U, int H, float* u_zr, float* u_h, float* ut_h, float* ut_zr, hipStream_t s) {
  size_t total = gru_pack_floats(H);
  hipLaunchKernelGGL(gru_pack_kernel, dim3((unsigned)cdivz(total, 256)), dim3(256), 0, s, U, H, u_zr, u_h, ut_h, ut_zr);
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
  static const int dbg = getenv("LSTUR_GRU_DEBUG") ? atoi(getenv("LSTUR_GRU_DEBUG")) : 0;  // measurement knobs
  GruFwdArgs a{xw, gmask, h0, u_zr_packed, u_h_packed, hs, zs, rs, hhs, n, W, H, dbg};
  hipLaunchKernelGGL(gru_fwd_kernel, dim3(cdiv(n, 16)), dim3(GW * 64), lds, s, a);
}

void launch_gru_bwd(const float* dh_last, const uint8_t* gmask, const float* hs, const float* zs, const float* rs,
                    const float* hhs, const float* ut_zr_packed, const float* ut_h_packed, float* dgates, float* rh,
                    float* hprev, float* dh0, int n, int W, int H, hipStream_t s) {
  if (!n) return;
  int Hp = ceil16(H), LDH = Hp + 4, LD2 = 2 * Hp + 4;
  size_t lds = (size_t)(3 * 16 * LDH + 16 * LD2) * sizeof(float);
  static size_t cur = 0;
  set_lds_attr((const void*)gru_bwd_kernel, lds, &cur);
  GruBwdArgs a{dh_last, gmask, hs, zs, rs, hhs, ut_zr_packed, ut_h_packed, dgates, rh, hprev, dh0, n, W, H};
  hipLaunchKernelGGL(gru_bwd_kernel, dim3(cdiv(n, 16)), dim3(GW * 64), lds, s, a);
}

}  // namespace lstur
