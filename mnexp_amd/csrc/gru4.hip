// Masked Keras-2.2.4 GRU recurrence + BPTT with the recurrent matrix RESIDENT ON CHIP (gfx950), H <= 200.
//
// Same mathematics as gru.hip (reference: task/paper.py:592, 597-613 -- Masking() + GRU(H), gate order z|r|h,
// hard_sigmoid / tanh, reset_after=False, K.rnn mask carry); this file is the fast path, gru.hip the general one.
//
// Why another kernel: the recurrence is a chain of W dependent skinny products [rows x H] . [H x 3H].  With 16
// sample rows per workgroup (one 16x16x4 MFMA row tile) a B=64 batch fills 4 of 256 CUs and every step re-streams
// all 480 KB of U from L2 (>= 3 us at 64 B/clk/CU) behind ~6 us of MFMA issue.  Here a workgroup owns FOUR sample
// rows and computes with v_mfma_f32_4x4x1_16b_f32 in A-broadcast mode (cbsz = 4): lanes 0..3 carry the 4 rows'
// state for one k, lane l carries U[k][col0 + l], so ONE instruction updates a [4 rows x 64 columns] strip
// (register j / lane l = row j, column col0 + l; layout probed on hardware by tools/probe_mfma4.hip).  4x more
// workgroups, and U never moves after the prologue: a wave keeps its 64-column strip of one operand in 200 VGPRs
// (KQ float4 = 4 k each; 8 waves x 256 VGPRs), what does not fit in registers sits in LDS.  A step issues no
// global load on its critical path except the prefetched gate inputs.
//
// Forward, per step:   phase 1  [z|r] = hs(xw_zr + h U_zr)     2H columns = G1 strips, waves 0..G1-1 (resident)
//                      phase 2  hh = tanh(xw_h + (r*h) U_h)    H  columns = G2 strips, waves 4..4+G2-1
//                               (strip j is resident when wave 4+j has no phase-1 strip, else it is read from LDS)
// BPTT, per step:      stage 1  drh = dpre_h U_h^T  and  part_z = dpre_z U_z^T      (independent: 2*G2 strips)
//                      stage 2  dh_prev = ... + dpre_r U_r^T                         (G2 strips)
//                      8 strips resident (one per wave), the rest (the 8-column tail strips and one full strip) in LDS.
// State and gate vectors live in LDS in the A-operand image [k/4][row][k%4] (one ds_read_b128 feeds 4 MFMAs).
// Saved activations / gradients are plain row-major tensors (lane = column => coalesced), so the fragment-layout
// conversion kernels of the general path are not needed.
#include <utility>

#include "lstur_device.h"
#include "lstur_internal.h"

namespace lstur {

constexpr int G4_THREADS = 512;   // 8 waves, 2 per SIMD -> 256 VGPRs per wave
constexpr size_t G4_LDS_MAX = 160 * 1024;

// cbsz = 4: ONE 4-lane block of A feeds all 16 blocks; ABID picks which block (lanes 4*ABID .. 4*ABID+3)
template <int ABID>
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, ABID, 0);
}
// position of element (k, row) in an A-operand image [k/4][row][k%4]
__host__ __device__ __forceinline__ int aidx(int k, int row) { return (k >> 2) * 16 + row * 4 + (k & 3); }

// The whole A vector of a strip product in registers: lane l of load i holds k-quad 16i + l/4 of row l&3, i.e. the
// image is read as plain consecutive float4 (1 KiB per wave instruction) and quad q is MFMA block q%16 of load q/16.
// 4 ds_read_b128 feed all 200 MFMAs of an H = 200 strip; nothing waits on LDS inside the MFMA stream.
// (Loads past quad KQ-1 run into the LDS region behind the image; those registers are never used.)
template <int KQ>
struct AVec {
  static constexpr int NA = (KQ + 15) / 16;
  float4 v[NA];
  __device__ __forceinline__ void load(const float* __restrict__ img, int lane) {
#pragma unroll
    for (int i = 0; i < NA; ++i) v[i] = *(const float4*)(img + i * 256 + lane * 4);
  }
};

template <int Q, typename B4>
__device__ __forceinline__ void quad_mfma(const float4& a, const B4& b, f32x4& c0, f32x4& c1) {
  c0 = mfma4<Q % 16>(a.x, b.x, c0);
  c1 = mfma4<Q % 16>(a.y, b.y, c1);
  c0 = mfma4<Q % 16>(a.z, b.z, c0);
  c1 = mfma4<Q % 16>(a.w, b.w, c1);
}

// out[row j][strip column of this lane] = sum_k A[j][k] * B[k][col]; B resident in registers
template <int KQ, size_t... Q>
__device__ __forceinline__ f32x4 strip_res_impl(const float4 (&ub)[KQ], const AVec<KQ>& a, std::index_sequence<Q...>) {
  f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
  (quad_mfma<(int)Q>(a.v[Q / 16], ub[Q], c0, c1), ...);
  return c0 + c1;
}
template <int KQ>
__device__ __forceinline__ f32x4 strip_res(const float4 (&ub)[KQ], const float* __restrict__ aimg, int lane) {
  AVec<KQ> a;
  a.load(aimg, lane);
  return strip_res_impl<KQ>(ub, a, std::make_index_sequence<KQ>{});
}
// B resident for the first QR k-quads, the last KQ-QR read from a per-wave LDS tail [KQ-QR][64 lanes][4] (BPTT at
// H = 200 needs a few more live registers than the forward: 44 resident quads + 6 in LDS avoids scratch spills)
template <int KQ, int QR, size_t... Q, size_t... T>
__device__ __forceinline__ f32x4 strip_mix_impl(const float4 (&ub)[QR], const float* __restrict__ tailL,
                                                const AVec<KQ>& a, std::index_sequence<Q...>,
                                                std::index_sequence<T...>) {
  f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
  (quad_mfma<(int)Q>(a.v[Q / 16], ub[Q], c0, c1), ...);
  (quad_mfma<(int)(QR + T)>(a.v[(QR + T) / 16], *(const float4*)(tailL + T * 256), c0, c1), ...);
  return c0 + c1;
}
template <int KQ, int QR>
__device__ __forceinline__ f32x4 strip_mix(const float4 (&ub)[QR], const float* __restrict__ tailL,
                                           const float* __restrict__ aimg, int lane) {
  AVec<KQ> a;
  a.load(aimg, lane);
  return strip_mix_impl<KQ, QR>(ub, tailL, a, std::make_index_sequence<QR>{}, std::make_index_sequence<KQ - QR>{});
}
// B read from an LDS image [KQ][ncols][4] (uL already points at this lane's column)
template <int KQ, size_t... Q>
__device__ __forceinline__ f32x4 strip_lds_impl(const float* __restrict__ uL, int qstride, const AVec<KQ>& a,
                                                std::index_sequence<Q...>) {
  f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
  (quad_mfma<(int)Q>(a.v[Q / 16], *(const float4*)(uL + (size_t)Q * qstride), c0, c1), ...);
  return c0 + c1;
}
template <int KQ>
__device__ __forceinline__ f32x4 strip_lds(const float* __restrict__ uL, int qstride, const float* __restrict__ aimg,
                                           int lane) {
  AVec<KQ> a;
  a.load(aimg, lane);
  return strip_lds_impl<KQ>(uL, qstride, a, std::make_index_sequence<KQ>{});
}

// ---- operand images -----------------------------------------------------------------------------
// All images are [KQ][N][4] (float4 = 4 consecutive k of one column), zero for k >= H:
//   izr N = 2H  B[k][c] = U[k][c]          (forward z|r)
//   ih  N = H   B[k][c] = U[k][2H + c]     (forward candidate)
//   itz N = H   B[k][n] = U[n][k]          (BPTT, dpre_z . U_z^T)
//   itr N = H   B[k][n] = U[n][H + k]      (BPTT, dpre_r . U_r^T)
//   ith N = H   B[k][n] = U[n][2H + k]     (BPTT, dpre_h . U_h^T)
int gru4_kq(int H) {
  if (getenv("LSTUR_GRU_GENERIC")) return 0;
  if (H < 1 || H > 200) return 0;
  const int q = cdiv(H, 4);
  return q <= 10 ? 10 : (q <= 32 ? 32 : 50);
}
size_t gru4_pack_floats(int H) {
  const int KQ = gru4_kq(H);
  return KQ ? (size_t)KQ * 4 * 6 * H : 0;
}
__global__ void gru4_pack_kernel(const float* __restrict__ U, int H, int KQ, float* __restrict__ img) {
  const size_t e = (size_t)KQ * 4 * H;   // floats of an N = H image
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 6 * e) return;
  const int H3 = 3 * H;
  float v = 0.f;
  if (i < 2 * e) {   // izr
    const int j = (int)(i & 3), c = (int)((i >> 2) % (2 * H)), q = (int)((i >> 2) / (2 * H)), k = q * 4 + j;
    if (k < H) v = U[(size_t)k * H3 + c];
  } else {
    const size_t r = i - 2 * e;
    const int which = (int)(r / e);   // 0 ih, 1 itz, 2 itr, 3 ith
    const size_t x = r - (size_t)which * e;
    const int j = (int)(x & 3), c = (int)((x >> 2) % H), q = (int)((x >> 2) / H), k = q * 4 + j;
    if (k < H) {
      if (which == 0) v = U[(size_t)k * H3 + 2 * H + c];
      else v = U[(size_t)c * H3 + (which - 1) * H + k];
    }
  }
  img[i] = v;
}
void launch_gru4_pack(const float* U, int H, float* img, hipStream_t s) {
  const int KQ = gru4_kq(H);
  if (!KQ) return;
  const size_t total = gru4_pack_floats(H);
  hipLaunchKernelGGL(gru4_pack_kernel, dim3((unsigned)cdivz(total, 256)), dim3(256), 0, s, U, H, KQ, img);
}

// tape of the fast path: row-major sections ZR [n*W,2H] | HH [n*W,H] | HP [n*W,H] | RH [n*W,H]
size_t gru4_tape_floats(int n, int W, int H) { return gru4_kq(H) ? (size_t)5 * n * W * H : 0; }

// ---- forward -------------------------------------------------------------------------------------
struct Gru4Fwd {
  const float* xw; const uint8_t* gmask; const float* h0; const float* izr; const float* ih;
  float* zr; float* hh; float* hp; float* rh; float* hlast;
  int n, W, H;
};

static size_t gru4_fwd_lds(int H, int KQ) {
  const int G1 = cdiv(2 * H, 64), G2 = cdiv(H, 64);
  const int NL = std::max(0, std::min(G2, G1 - 4));
  const size_t floats = (size_t)2 * KQ * 16 + ((4 * H + 3) & ~3) + (size_t)KQ * NL * 64 * 4;
  const size_t aread = (size_t)KQ * 16 + (size_t)cdiv(KQ, 16) * 256;   // AVec reads whole 1 KiB blocks of the last image
  return std::max(floats, aread) * sizeof(float);
}

// Step masks of the 4 rows of this workgroup as wave-uniform bit words (bit i = step base+i), 64 steps at a time:
// one vector load + ballot per row and chunk instead of a byte load (and its wait) on every step.
struct RowMasks {
  unsigned long long bits[4];
  int chunk;
  const uint8_t* gm;
  uint32_t rbase[4];
  int W, lane;
  __device__ __forceinline__ void init(const uint8_t* gmask, int b0, int n, int W_, int lane_) {
    gm = gmask; W = W_; lane = lane_; chunk = -1;
#pragma unroll
    for (int j = 0; j < 4; ++j) rbase[j] = (uint32_t)min(b0 + j, n - 1) * (uint32_t)W_;
  }
  __device__ __forceinline__ void seek(int t) {   // wave-uniform t
    if ((t >> 6) == chunk) return;
    chunk = t >> 6;
    const int tt = chunk * 64 + lane;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint8_t v = tt < W ? gm[rbase[j] + tt] : (uint8_t)0;
      bits[j] = __ballot(v != 0);
    }
  }
  __device__ __forceinline__ bool at(int j, int t) const { return (bits[j] >> (t & 63)) & 1ull; }
};

// Time loop of one wave role.  P1: the wave owns (resident) strip `wv` of [z|r].  P2: 0 = no candidate strip,
// 1 = candidate strip j2 resident, 2 = candidate strip j2 read from the LDS image.  One straight-line body per role
// (no per-phase `if (role)` inside the loop), so every wait the compiler places is for an access of this role only.
// Global traffic goes through buffer descriptors: scalar row/step offset + ONE per-lane column offset (kBufOob for
// columns past H: loads return 0, stores are dropped) -- no address VGPRs next to the 200 operand registers.
template <int KQ, bool P1, int P2>
__device__ __forceinline__ void gru4_fwd_loop(const Gru4Fwd& g, float* hA, float* rhA, float* zL, const float* uL,
                                              int ncl, int wv, int lane, const float4 (&ub)[KQ]) {
  const int H = g.H, W = g.W, n = g.n, H2 = 2 * H, H3 = 3 * H;
  const int b0 = blockIdx.x * 4;
  const uint32_t e4 = (uint32_t)n * W * H * 4;   // bytes of an [n*W, H] tensor
  const __amdgpu_buffer_rsrc_t rxw = make_rsrc(g.xw, 3 * e4), rzr = make_rsrc(g.zr, 2 * e4), rhh = make_rsrc(g.hh, e4),
                               rhp = make_rsrc(g.hp, e4), rrh = make_rsrc(g.rh, e4);
  const int c1 = wv * 64 + lane;            // column of [z|r]
  const bool v1 = P1 && c1 < H2;
  const bool isz = c1 < H;
  const int hc1 = isz ? c1 : c1 - H;
  const int c2 = (wv - 4) * 64 + lane;      // hidden column of the candidate strip
  const bool v2 = P2 != 0 && c2 < H;
  const uint32_t o1 = v1 ? c1 * 4 : kBufOob;                     // [z|r] column (xw, zr)
  const uint32_t o1r = (v1 && !isz) ? hc1 * 4 : kBufOob;         // r columns -> rh
  const uint32_t o2 = v2 ? c2 * 4 : kBufOob, o2x = v2 ? (H2 + c2) * 4 : kBufOob;
  const float* uLl = uL + (size_t)(P2 == 2 ? min(c2, ncl - 1) : 0) * 4;
    uint32_t rb[4];   // row * W (the launcher keeps n*W*3H below 2^30)
  bool rok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    rok[j] = b0 + j < n;
    rb[j] = (uint32_t)min(b0 + j, n - 1) * (uint32_t)W;
  }
  RowMasks rm;
  if (P2) rm.init(g.gmask, b0, n, W, lane);
  for (int t = 0; t < W; ++t) {
    // ---- phase 1: z | r ----
    if (P1) {
      float x1[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) x1[j] = buf_load1(rxw, o1, (rb[j] + t) * H3 * 4);
      __builtin_amdgcn_sched_barrier(0);   // keep the gate-input loads in front of the MFMA stream
      const f32x4 acc = strip_res<KQ>(ub, hA, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float gt = hard_sigmoid(acc[j] + x1[j]);
        if (rok[j]) buf_store1(gt, rzr, o1, (rb[j] + t) * H2 * 4);
        if (v1) {
          if (isz) {
            zL[j * H + hc1] = gt;
          } else {
            const float v = gt * hA[aidx(hc1, j)];
            rhA[aidx(hc1, j)] = v;
            if (rok[j]) buf_store1(v, rrh, o1r, (rb[j] + t) * H * 4);
          }
        }
      }
    }
    float x2[4];
    if (P2) {   // candidate inputs fly over the barrier and the candidate product
      rm.seek(t);
#pragma unroll
      for (int j = 0; j < 4; ++j) x2[j] = buf_load1(rxw, o2x, (rb[j] + t) * H3 * 4);
    }
    __syncthreads();
    // ---- phase 2: candidate + state update ----
    if (P2) {
      const f32x4 acc = P2 == 1 ? strip_res<KQ>(ub, rhA, lane) : strip_lds<KQ>(uLl, ncl * 4, rhA, lane);
      if (v2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float hhv = tanhf(acc[j] + x2[j]);
          const float z = zL[j * H + c2];
          const float hp = hA[aidx(c2, j)];
          const float hn = z * hp + (1.0f - z) * hhv;
          if (rok[j]) {
            buf_store1(hhv, rhh, o2, (rb[j] + t) * H * 4);
            buf_store1(hp, rhp, o2, (rb[j] + t) * H * 4);
          }
          hA[aidx(c2, j)] = rm.at(j, t) ? hn : hp;   // K.rnn: a masked step carries the previous state/output
        }
      }
    }
    __syncthreads();
  }
}

template <int KQ>
__global__ __launch_bounds__(G4_THREADS) void gru4_fwd_kernel(Gru4Fwd g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = g.H, n = g.n, H2 = 2 * H;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G1 = cdiv(H2, 64), G2 = cdiv(H, 64);
  const int NL = max(0, min(G2, G1 - 4));   // phase-2 strips whose wave also owns a phase-1 strip: fed from LDS
  const int ncl = NL * 64;
  float* hA = smem;                         // state, A image
  float* rhA = hA + KQ * 16;                // r*h, A image
  float* zL = rhA + KQ * 16;                // z [4][H]
  float* uL = zL + ((4 * H + 3) & ~3);      // U_h[:, 0:ncl] image [KQ][ncl][4]
  const int b0 = blockIdx.x * 4;

  for (int i = tid; i < 2 * KQ * 16; i += G4_THREADS) smem[i] = 0.f;
  __syncthreads();
  for (int i = tid; i < 4 * H; i += G4_THREADS) {
    const int row = i / H, col = i - row * H, b = b0 + row;
    if (g.h0 && b < n) hA[aidx(col, row)] = g.h0[(size_t)b * H + col];
  }
  for (int i = tid; i < KQ * ncl; i += G4_THREADS) {
    const int q = i / ncl, col = i - q * ncl;
    ((float4*)uL)[i] = ((const float4*)g.ih)[(size_t)q * H + col];
  }

  // roles (all wave-uniform)
  const bool p1 = wv < G1;
  const int j2 = wv - 4;
  const bool p2 = j2 >= 0 && j2 < G2;
  const bool p2res = p2 && !p1;
  float4 ub[KQ];
  {
    const int c1 = wv * 64 + lane, c2 = j2 * 64 + lane;
    const float4* src = p1 ? (const float4*)g.izr : (const float4*)g.ih;
    const int ncol = p1 ? H2 : H;
    const bool ok = p1 ? c1 < H2 : (p2res && c2 < H);
    const int col = ok ? (p1 ? c1 : c2) : 0;
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      float4 v = src[(size_t)q * ncol + col];
      ub[q] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __syncthreads();
  if (p1 && p2) gru4_fwd_loop<KQ, true, 2>(g, hA, rhA, zL, uL, ncl, wv, lane, ub);
  else if (p1) gru4_fwd_loop<KQ, true, 0>(g, hA, rhA, zL, uL, ncl, wv, lane, ub);
  else if (p2res) gru4_fwd_loop<KQ, false, 1>(g, hA, rhA, zL, uL, ncl, wv, lane, ub);
  else gru4_fwd_loop<KQ, false, 0>(g, hA, rhA, zL, uL, ncl, wv, lane, ub);

  for (int i = tid; i < 4 * H; i += G4_THREADS) {
    const int row = i / H, col = i - row * H, b = b0 + row;
    if (b < n) g.hlast[(size_t)b * H + col] = hA[aidx(col, row)];
  }
}

// ---- BPTT ----------------------------------------------------------------------------------------
//   dhn = m ? dh : 0 ; carry = m ? 0 : dh
//   dpre_h = dhn (1-z)(1-hh^2) ; dpre_z = dhn (hp-hh) hs'(z)
//   drh = dpre_h U_h^T ; dpre_r = drh hp hs'(r)
//   dh_prev = carry + dhn z + drh r + dpre_z U_z^T + dpre_r U_r^T
struct Gru4Bwd {
  const float* dh_last; const uint8_t* gmask; const float* zr; const float* hh; const float* hp;
  const float* itz; const float* itr; const float* ith;
  float* dgates; float* dh0;
  int n, W, H;
};
__device__ __forceinline__ float hs_grad4(float g) { return (g > 0.f && g < 1.f) ? 0.2f : 0.f; }

// strip -> wave tables (strip j of a product covers hidden columns 64j .. 64j+63)
//   stage 1: drh strips on waves {0,1,2,6}, part_z strips on waves {3,4,5,7}; stage 2: strips on waves {6,7,0,1}.
//   resident: drh 0..2, part_z 0..2, stage-2 0..1;  LDS: drh 3, part_z 3, stage-2 2 and 3.
__host__ __device__ constexpr int bwd_resident_quads(int KQ) { return KQ > 44 ? 44 : KQ; }
static size_t gru4_bwd_lds(int H, int KQ) {
  const int G2 = cdiv(H, 64);
  const int w2 = G2 >= 3 ? std::min(64, H - 128) : 0, w3 = G2 >= 4 ? H - 192 : 0;
  const size_t floats = (size_t)3 * KQ * 16 + 2 * ((4 * H + 3) & ~3) + (size_t)KQ * 4 * (w2 + 3 * w3) +
                        (size_t)8 * (KQ - bwd_resident_quads(KQ)) * 256;
  const size_t aread = (size_t)2 * KQ * 16 + (size_t)cdiv(KQ, 16) * 256;
  return std::max(floats, aread) * sizeof(float);
}

struct Gru4BwdLds {
  float *dphA, *dpzA, *dprA, *n2, *n3;
  const float *u1, *u2;     // LDS operand images of this wave's stage-1 / stage-2 strip (lane-adjusted), if LDS-fed
  int q1, q2;               // their k-quad strides (floats)
  const float* tail;        // this lane's LDS tail of the resident operand
};

// Time loop of one wave role.  S1: 0 none | 1 drh resident | 2 part_z resident | 3 drh from LDS | 4 part_z from LDS
// (strip s1grp);  S2: 0 none | 1 stage-2 strip resident | 2 from LDS (strip s2grp).  The resident operand belongs to
// S1 when S1 is 1 or 2, else to S2.
template <int KQ, int S1, int S2>
__device__ __forceinline__ void gru4_bwd_loop(const Gru4Bwd& g, const Gru4BwdLds& L, int s1grp, int s2grp, int lane,
                                              const float4 (&ub)[bwd_resident_quads(KQ)]) {
  constexpr int QR = bwd_resident_quads(KQ);
  const int H = g.H, W = g.W, n = g.n, H2 = 2 * H, H3 = 3 * H;
  const int b0 = blockIdx.x * 4;
  const uint32_t e4 = (uint32_t)n * W * H * 4;   // bytes of an [n*W, H] tensor
  const __amdgpu_buffer_rsrc_t rzr = make_rsrc(g.zr, 2 * e4), rhh = make_rsrc(g.hh, e4), rhp = make_rsrc(g.hp, e4),
                               rdg = make_rsrc(g.dgates, 3 * e4);
  const int cs1 = s1grp * 64 + lane;           // output column of the stage-1 strip
  const bool vs1 = S1 != 0 && cs1 < H;
  const int cs2 = s2grp * 64 + lane;
  const bool vs2 = S2 != 0 && cs2 < H;
  const uint32_t o1 = vs1 ? cs1 * 4 : kBufOob, o1r = vs1 ? (H + cs1) * 4 : kBufOob;
  const uint32_t o2 = vs2 ? cs2 * 4 : kBufOob, o2h = vs2 ? (H2 + cs2) * 4 : kBufOob;
  uint32_t rb[4];   // row * W (the launcher keeps n*W*3H below 2^30)
  bool rok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    rok[j] = b0 + j < n;
    rb[j] = (uint32_t)min(b0 + j, n - 1) * (uint32_t)W;
  }
  RowMasks rm;
  if (S2 != 0) rm.init(g.gmask, b0, n, W, lane);

  // element-wise head of step t for the (row, column) elements this lane owns in stage 2
  float n1[4] = {0.f, 0.f, 0.f, 0.f};
  auto head = [&](int t, const float (&dh)[4], const float (&z)[4], const float (&hh)[4], const float (&hp)[4]) {
    rm.seek(t);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool m = rm.at(j, t);
      const float dhn = m ? dh[j] : 0.f;
      const float dph = dhn * (1.0f - z[j]) * (1.0f - hh[j] * hh[j]);
      const float dpz = dhn * (hp[j] - hh[j]) * hs_grad4(z[j]);
      n1[j] = (m ? 0.f : dh[j]) + dhn * z[j];
      if (vs2) {
        L.dphA[aidx(cs2, j)] = dph;
        L.dpzA[aidx(cs2, j)] = dpz;
      }
      if (rok[j]) {
        buf_store1(dpz, rdg, o2, (rb[j] + t) * H3 * 4);
        buf_store1(dph, rdg, o2h, (rb[j] + t) * H3 * 4);
      }
    }
  };
  if (S2 != 0) {
    float dh[4], z[4], hh[4], hp[4];
    const int t = W - 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      dh[j] = (rok[j] && vs2) ? g.dh_last[(size_t)(b0 + j) * H + cs2] : 0.f;
      z[j] = buf_load1(rzr, o2, (rb[j] + t) * H2 * 4);
      hh[j] = buf_load1(rhh, o2, (rb[j] + t) * H * 4);
      hp[j] = buf_load1(rhp, o2, (rb[j] + t) * H * 4);
    }
    head(t, dh, z, hh, hp);
  }
  __syncthreads();

  for (int t = W - 1; t >= 0; --t) {
    // ---- stage 1 ----
    if (S1 == 1 || S1 == 3) {
      float r[4], hp[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        r[j] = buf_load1(rzr, o1r, (rb[j] + t) * H2 * 4);
        hp[j] = buf_load1(rhp, o1, (rb[j] + t) * H * 4);
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the loads in front of the MFMA stream
      const f32x4 acc = S1 == 1 ? strip_mix<KQ, QR>(ub, L.tail, L.dphA, lane) : strip_lds<KQ>(L.u1, L.q1, L.dphA, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float drh = acc[j];
        const float dpr = drh * hp[j] * hs_grad4(r[j]);
        if (vs1) {
          L.dprA[aidx(cs1, j)] = dpr;
          L.n2[j * H + cs1] = drh * r[j];
        }
        if (rok[j]) buf_store1(dpr, rdg, o1r, (rb[j] + t) * H3 * 4);
      }
    } else if (S1 == 2 || S1 == 4) {
      const f32x4 acc = S1 == 2 ? strip_mix<KQ, QR>(ub, L.tail, L.dpzA, lane) : strip_lds<KQ>(L.u1, L.q1, L.dpzA, lane);
      if (vs1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) L.n3[j * H + cs1] = acc[j];
      }
    }
    __syncthreads();
    // ---- stage 2 (+ the element-wise head of step t-1) ----
    if (S2 != 0) {
      float z[4], hh[4], hp[4];
      const int tp = max(t - 1, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        z[j] = buf_load1(rzr, o2, (rb[j] + tp) * H2 * 4);
        hh[j] = buf_load1(rhh, o2, (rb[j] + tp) * H * 4);
        hp[j] = buf_load1(rhp, o2, (rb[j] + tp) * H * 4);
      }
      __builtin_amdgcn_sched_barrier(0);
      const f32x4 acc = S2 == 1 ? strip_mix<KQ, QR>(ub, L.tail, L.dprA, lane) : strip_lds<KQ>(L.u2, L.q2, L.dprA, lane);
      float dh[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) dh[j] = vs2 ? n1[j] + L.n2[j * H + cs2] + L.n3[j * H + cs2] + acc[j] : 0.f;
      if (t > 0) {
        head(t - 1, dh, z, hh, hp);
      } else if (vs2) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (rok[j]) g.dh0[(size_t)(b0 + j) * H + cs2] = dh[j];
      }
    }
    __syncthreads();
  }
}

template <int KQ>
__global__ __launch_bounds__(G4_THREADS) void gru4_bwd_kernel(Gru4Bwd g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = g.H;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G2 = cdiv(H, 64);
  const int w2 = G2 >= 3 ? min(64, H - 128) : 0, w3 = G2 >= 4 ? H - 192 : 0;
  const int H4 = (4 * H + 3) & ~3;
  constexpr int QR = bwd_resident_quads(KQ);
  Gru4BwdLds L;
  L.dphA = smem;                    // dpre_h, A image
  L.dpzA = L.dphA + KQ * 16;        // dpre_z
  L.dprA = L.dpzA + KQ * 16;        // dpre_r
  L.n2 = L.dprA + KQ * 16;          // drh * r          [4][H]
  L.n3 = L.n2 + H4;                 // dpre_z . U_z^T   [4][H]
  float* Lr2 = L.n3 + H4;                    // itr[:, 128:128+w2]   [KQ][w2][4]
  float* Lh3 = Lr2 + (size_t)KQ * w2 * 4;    // ith[:, 192:H]
  float* Lz3 = Lh3 + (size_t)KQ * w3 * 4;    // itz[:, 192:H]
  float* Lr3 = Lz3 + (size_t)KQ * w3 * 4;    // itr[:, 192:H]
  float* tailL = Lr3 + (size_t)KQ * w3 * 4 + (size_t)wv * (KQ - QR) * 256 + lane * 4;   // this lane's resident tail
  L.tail = tailL;

  for (int i = tid; i < 3 * KQ * 16 + 2 * H4; i += G4_THREADS) smem[i] = 0.f;
  for (int i = tid; i < KQ * w2; i += G4_THREADS) {
    const int q = i / w2, c = i - q * w2;
    ((float4*)Lr2)[i] = ((const float4*)g.itr)[(size_t)q * H + 128 + c];
  }
  for (int i = tid; i < KQ * w3; i += G4_THREADS) {
    const int q = i / w3, c = i - q * w3;
    ((float4*)Lh3)[i] = ((const float4*)g.ith)[(size_t)q * H + 192 + c];
    ((float4*)Lz3)[i] = ((const float4*)g.itz)[(size_t)q * H + 192 + c];
    ((float4*)Lr3)[i] = ((const float4*)g.itr)[(size_t)q * H + 192 + c];
  }

  // ---- roles (wave-uniform): strip j of a product covers hidden columns 64j .. 64j+63 ----
  //   stage 1: drh strips on waves {0,1,2,6}, part_z strips on waves {3,4,5,7}; stage 2: strips on waves {6,7,0,1}.
  //   resident: drh 0..2, part_z 0..2, stage-2 0..1;  LDS: drh 3, part_z 3, stage-2 2 and 3.
  int s1 = 0, s1grp = 0;
  if (wv <= 2) { s1 = 1; s1grp = wv; }
  else if (wv <= 5) { s1 = 2; s1grp = wv - 3; }
  else if (wv == 6) { s1 = 3; s1grp = 3; }
  else { s1 = 4; s1grp = 3; }
  if (s1grp >= G2) s1 = 0;
  const int s2grp = wv >= 6 ? wv - 6 : (wv <= 1 ? wv + 2 : 99);
  const int s2 = s2grp >= G2 ? 0 : (s2grp < 2 ? 1 : 2);
  const bool s1res = s1 == 1 || s1 == 2;
  L.u1 = (s1 == 3 ? Lh3 : Lz3) + (size_t)min(lane, max(w3 - 1, 0)) * 4;
  L.q1 = w3 * 4;
  L.u2 = s2grp == 2 ? Lr2 + (size_t)min(lane, max(w2 - 1, 0)) * 4 : Lr3 + (size_t)min(lane, max(w3 - 1, 0)) * 4;
  L.q2 = (s2grp == 2 ? w2 : w3) * 4;

  float4 ub[QR];
  {
    const int cs1 = s1grp * 64 + lane, cs2 = s2grp * 64 + lane;
    const float4* src = s1res ? (const float4*)(s1 == 1 ? g.ith : g.itz) : (const float4*)g.itr;
    const bool ok = s1res ? cs1 < H : (s2 == 1 && cs2 < H);
    const int col = ok ? (s1res ? cs1 : cs2) : 0;
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      float4 v = src[(size_t)q * H + col];
      v = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      if (q < QR) ub[q] = v;
      else *(float4*)(tailL + (q - QR) * 256) = v;
    }
  }
  __syncthreads();

#define LSTUR_ROLE(A, B) \
  if (s1 == A && s2 == B) gru4_bwd_loop<KQ, A, B>(g, L, s1grp, s2grp, lane, ub); else
  LSTUR_ROLE(1, 2) LSTUR_ROLE(1, 0) LSTUR_ROLE(2, 0) LSTUR_ROLE(3, 1) LSTUR_ROLE(4, 1) LSTUR_ROLE(0, 1)
  gru4_bwd_loop<KQ, 0, 0>(g, L, s1grp, s2grp, lane, ub);
#undef LSTUR_ROLE
}

// ---- launchers -------------------------------------------------------------------------------------
template <typename K>
static void set_lds(K kern, size_t lds) {
  if (lds > 65536) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

bool gru4_supported(int H) {
  const int KQ = gru4_kq(H);
  if (!KQ) return false;
  const int G1 = cdiv(2 * H, 64), G2 = cdiv(H, 64);
  const int NL = std::max(0, std::min(G2, G1 - 4));
  return G1 <= 8 && G2 <= 4 && NL * 64 <= H && gru4_fwd_lds(H, KQ) <= G4_LDS_MAX && gru4_bwd_lds(H, KQ) <= G4_LDS_MAX;
}

void gru4_tape_rows(float* tape, int n, int W, int H, float** hprev, float** rh) {
  const size_t e = (size_t)n * W * H;
  *hprev = tape + 3 * e;
  *rh = tape + 4 * e;
}

void launch_gru4_fwd(const float* xw, const uint8_t* gmask, const float* h0, const float* img, float* tape, float* hlast,
                     int n, int W, int H, hipStream_t s) {
  const int KQ = gru4_kq(H);
  const size_t e = (size_t)n * W * H, ie = (size_t)KQ * 4 * H;
  Gru4Fwd a{xw, gmask, h0, img, img + 2 * ie, tape, tape + 2 * e, tape + 3 * e, tape + 4 * e, hlast, n, W, H};
  const size_t lds = gru4_fwd_lds(H, KQ);
  const dim3 grid(cdiv(n, 4)), block(G4_THREADS);
#define LSTUR_G4F(Q)                                              \
  {                                                               \
    set_lds(gru4_fwd_kernel<Q>, lds);                             \
    hipLaunchKernelGGL(gru4_fwd_kernel<Q>, grid, block, lds, s, a); \
  }
  if (KQ == 10) LSTUR_G4F(10) else if (KQ == 32) LSTUR_G4F(32) else LSTUR_G4F(50)
#undef LSTUR_G4F
}

void launch_gru4_bwd(const float* dh_last, const uint8_t* gmask, float* tape, const float* img, float* dgates, float* rh,
                     float* hprev, float* dh0, int n, int W, int H, hipStream_t s) {
  const int KQ = gru4_kq(H);
  const size_t e = (size_t)n * W * H, ie = (size_t)KQ * 4 * H;
  Gru4Bwd a{dh_last, gmask, tape, tape + 2 * e, tape + 3 * e, img + 3 * ie, img + 4 * ie, img + 5 * ie, dgates, dh0, n, W, H};
  const size_t lds = gru4_bwd_lds(H, KQ);
  const dim3 grid(cdiv(n, 4)), block(G4_THREADS);
#define LSTUR_G4B(Q)                                              \
  {                                                               \
    set_lds(gru4_bwd_kernel<Q>, lds);                             \
    hipLaunchKernelGGL(gru4_bwd_kernel<Q>, grid, block, lds, s, a); \
  }
  if (KQ == 10) LSTUR_G4B(10) else if (KQ == 32) LSTUR_G4B(32) else LSTUR_G4B(50)
#undef LSTUR_G4B
  // the forward saved hprev / r*hprev row-major in the tape; hand them out where the caller wants copies
  float *tp_hprev, *tp_rh;
  gru4_tape_rows(tape, n, W, H, &tp_hprev, &tp_rh);
  if (hprev && hprev != tp_hprev) (void)hipMemcpyAsync(hprev, tp_hprev, e * sizeof(float), hipMemcpyDeviceToDevice, s);
  if (rh && rh != tp_rh) (void)hipMemcpyAsync(rh, tp_rh, e * sizeof(float), hipMemcpyDeviceToDevice, s);
}

}  // namespace lstur
