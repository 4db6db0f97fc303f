// Masked Keras-2.2.4 GRU recurrence + BPTT with the recurrent matrix RESIDENT ON CHIP (gfx950), H <= 200.
//
// Same mathematics as gru.hip (reference: task/paper.py:592, 597-613 -- Masking() + GRU(H), gate order z|r|h,
// hard_sigmoid / tanh, reset_after=False, K.rnn mask carry); this file is the fast path, gru.hip the general one.
//
// Why another kernel: the recurrence is a chain of W dependent skinny products [rows x H] . [H x 3H].  With 16
// sample rows per workgroup (one 16x16x4 MFMA row tile) a B=64 batch fills 4 of 256 CUs and every step re-streams
// all 480 KB of U from L2 (>= 3 us at 64 B/clk/CU) behind ~6 us of MFMA issue.  Here a workgroup owns FOUR sample
// rows and computes with v_mfma_f32_4x4x1_16b_f32 in A-broadcast mode (cbsz = 4): one 4-lane block of A carries the
// 4 rows' state for one k, lane l carries U[k][col0 + l], so ONE instruction updates a [4 rows x 64 columns] strip
// (register j / lane l = row j, column col0 + l; layout probed on hardware by tools/probe_mfma4.hip, issue rate
// 8.4 cycles per instruction by tools/probe_mfma4_rate.hip).  4x more workgroups, and U never moves after the
// prologue: the workgroup is 4 waves, one per SIMD, so each wave owns the SIMD's whole 512-entry register file
// (VGPR + AGPR; MFMA reads its B operand from either) and keeps TWO 64-column strips of operand resident (2 x 200
// registers at H = 200); the strips that do not fit (3 of 11 forward, 4 of 12 backward) are read from LDS images
// through a register ring.  A step issues no global load on its critical path except the prefetched gate inputs.
//
// Forward, per step:   phase 1  [z|r] = hs(xw_zr + h U_zr)     2H columns = G1 strips: wave s owns strips s, s+4
//                      phase 2  hh = tanh(xw_h + (r*h) U_h)    H  columns = G2 strips, one per wave
// BPTT, per step:      stage 1  drh = dpre_h U_h^T (waves 0, 2)  and  part_z = dpre_z U_z^T (waves 1, 3)
//                      stage 2  dh_prev = ... + dpre_r U_r^T     one strip per wave
// State and gate vectors live in LDS in the A-operand image [k/4][row][k%4]; a wave reads a whole image with 4
// ds_read_b128 (lane l = k-quad 16i + l/4, row l%4) and selects the quad per MFMA with the ABID field.
// Saved activations / gradients are plain row-major tensors (lane = column => coalesced), addressed through
// buffer descriptors (scalar row/step offset + one per-lane column offset; out-of-range columns are dropped by
// the range check) -- no address VGPRs, no exec-mask branches.
//
// FORWARD, F16 = true (the default since round 3; LSTUR_GRU_F32=1 selects the fp32 instructions): the products run on
// v_mfma_f32_4x4x4_16b_f16 -- FOUR k per instruction at the cycles of the one-k fp32 instruction (tools/probe_mfma4h_rate.hip:
// 8.5 against 8.5 cycles per instruction in two chains) -- on fp16 PAIRS of both operands (bf16_mfma.h: x s = h + l, three
// products hh + hl + lh per fp32-grade product): 3 K / 4 instead of K instructions per 64 columns.  An operand element
// keeps its 4 bytes -- a (k-quad, column) group of the images is [h0 h1 h2 h3 | l0 l1 l2 l3] instead of four floats -- so the
// register / LDS budget of the resident-U design is unchanged.  Scales (powers of two, undone exactly per accumulator): the
// recurrent matrix by max |U| (kept with the operand images, lstur_internal.h: gru4 pack tail); the forward's state
// vectors by max(1, max |h0| of the workgroup's rows) (|h_t| <= max(|h_t-1|, 1): a convex combination with a tanh).
// Measured (tools/bench_gru.py, same box, cycle stamps of the bench-only library): 6 560 against 7 220 cycles per step (155 against
// 166 us at B = 64, W = 50): the MFMA streams shrink by 25 % / 15 %, the epilogues grow by 45 % / 20 % (a pair costs a
// multiply, three conversions, a subtraction and two 16-bit LDS stores where a float cost one store).
// BPTT stays on the fp32 instructions: its fp16 form was built and measured SLOWER (183 against 174 us): its three gradient
// vectors change scale with dh from step to step (a lagged per-step maximum, clamped stores) and every step writes 16
// pairs per lane -- the epilogue work outweighs the shorter MFMA streams.  Its images (itz, itr, ith) therefore stay floats.
#include <utility>

#include "bf16_mfma.h"
#include "lstur_device.h"
#include "lstur_internal.h"

namespace lstur {

typedef _Float16 g4h4 __attribute__((ext_vector_type(4)));
// the h | l halves of a 16-byte pair group
__device__ __forceinline__ g4h4 pair_h(const float4& v) { return __builtin_bit_cast(g4h4, make_float2(v.x, v.y)); }
__device__ __forceinline__ g4h4 pair_l(const float4& v) { return __builtin_bit_cast(g4h4, make_float2(v.z, v.w)); }
template <int ABID>
__device__ __forceinline__ f32x4 mfma4h(g4h4 a, g4h4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c, 4, ABID, 0);
}
// element (k, row) of a pair image: the halfword of its h part (the l part sits 4 halfwords behind it)
__device__ __forceinline__ unsigned short* pair_ptr(float* img, int k) {
  return (unsigned short*)img + 2 * ((k >> 2) * 16) + (k & 3);     // + row * 8
}
// v (already scaled) -> its fp16 pair at p (h) and p + 4 (l); clamped to the fp16 range (a second fence: the scales keep
// the operands far inside it)
template <bool CLAMP = true>
__device__ __forceinline__ void store_pair(unsigned short* p, float v) {
  if (CLAMP) v = fminf(fmaxf(v, -65504.f), 65504.f);
  const _Float16 h = (_Float16)v;
  const _Float16 l = (_Float16)(v - (float)h);
  p[0] = __builtin_bit_cast(unsigned short, h);
  p[4] = __builtin_bit_cast(unsigned short, l);
}

constexpr int G4_THREADS = 256;   // 4 waves, 1 per SIMD -> 512 registers per wave
constexpr size_t G4_LDS_MAX = 160 * 1024;
enum { M_NONE = 0, M_RES = 1, M_LDS = 2 };

// cbsz = 4: ONE 4-lane block of A feeds all 16 blocks; ABID picks which block (lanes 4*ABID .. 4*ABID+3)
template <int ABID>
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, ABID, 0);
}
// position of element (k, row) in an A-operand image [k/4][row][k%4]
__host__ __device__ __forceinline__ int aidx(int k, int row) { return (k >> 2) * 16 + row * 4 + (k & 3); }

// The whole A vector of a strip product in registers: lane l of load i holds k-quad 16i + l/4 of row l&3, i.e. the
// image is read as plain consecutive float4 (1 KiB per wave instruction) and quad q is MFMA block q%16 of load q/16.
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

constexpr int LDS_AHEAD = 6;   // k-quads the LDS operand reads run in front of their MFMAs (~120-cycle round trip)
// Resident k-quads (QR1) of the strip in register slot 1; its last KQ - QR1 quads sit in a per-wave LDS tail
// [q][64 lanes][4].  (2 x 200 operand registers + the working set of a step do not fit 512 at H = 200.)
__host__ __device__ constexpr int fwd_slot1(int KQ) { return KQ > 41 ? 41 : KQ; }   // (41: the state's second image of the fp16 form needs the 4 KB of LDS a 40-quad slot's tail took)
__host__ __device__ constexpr int bwd_slot1(int KQ) { return KQ > 36 ? 36 : KQ; }   // BPTT keeps more values live

// One k-quad of up to two strips that share the A vector.  Strip 0 is resident in slot 0 (or absent); strip 1 is
// resident in slot 1 (+ LDS tail), absent, or read from an LDS image (layout below).  LDS operands pass through
// `ring`, read LDS_AHEAD quads early.  MFMAs of the two strips alternate so that no accumulator is reused within
// 4 issues.
template <int KQ, int QR1, int Q, int M0, int M1, bool F16>
__device__ __forceinline__ void quad_step(const float4& a, const float4 (&u0)[KQ], const float4 (&u1)[QR1],
                                          const float* __restrict__ l1, float4 (&ring)[LDS_AHEAD], f32x4 (&c)[6]) {
  constexpr int QP = Q + LDS_AHEAD;
  float4 b1 = make_float4(0.f, 0.f, 0.f, 0.f);
  bool pinned = false;
  if (M1 == M_RES) {   // l1 = this lane's tail, quad stride 256 floats
    if (Q < QR1) b1 = u1[Q < QR1 ? Q : 0];
    else b1 = ring[Q % LDS_AHEAD];
    if (QP >= QR1 && QP < KQ) ring[QP % LDS_AHEAD] = *(const float4*)(l1 + (QP - QR1) * 256);
    pinned = QP >= QR1;
  }
  if (M1 == M_LDS) {
    b1 = ring[Q % LDS_AHEAD];
    if (QP < KQ) ring[QP % LDS_AHEAD] = *(const float4*)(l1 + QP * 4);
    pinned = true;
  }
  if (F16) {   // three products of the fp16 pairs, four k each, in THREE chains per strip: an accumulator comes round again
               // after 3 (one strip) or 6 (two strips) instructions -- with l h and h h in one chain the single-strip
               // stream ran at the chain's latency (12.1 cycles per MFMA, tools/bench_gru.py stamps) instead of its issue rate
    if (M0 == M_RES) c[0] = mfma4h<Q % 16>(pair_l(a), pair_h(u0[Q]), c[0]);
    if (M1 != M_NONE) c[3] = mfma4h<Q % 16>(pair_l(a), pair_h(b1), c[3]);
    if (M0 == M_RES) c[1] = mfma4h<Q % 16>(pair_h(a), pair_l(u0[Q]), c[1]);
    if (M1 != M_NONE) c[4] = mfma4h<Q % 16>(pair_h(a), pair_l(b1), c[4]);
    if (M0 == M_RES) c[2] = mfma4h<Q % 16>(pair_h(a), pair_h(u0[Q]), c[2]);
    if (M1 != M_NONE) c[5] = mfma4h<Q % 16>(pair_h(a), pair_h(b1), c[5]);
    if (pinned) __builtin_amdgcn_sched_barrier(0);
    return;
  }
  if (M0 == M_RES) c[0] = mfma4<Q % 16>(a.x, u0[Q].x, c[0]);
  if (M1 != M_NONE) c[2] = mfma4<Q % 16>(a.x, b1.x, c[2]);
  if (M0 == M_RES) c[1] = mfma4<Q % 16>(a.y, u0[Q].y, c[1]);
  if (M1 != M_NONE) c[3] = mfma4<Q % 16>(a.y, b1.y, c[3]);
  if (M0 == M_RES) c[0] = mfma4<Q % 16>(a.z, u0[Q].z, c[0]);
  if (M1 != M_NONE) c[2] = mfma4<Q % 16>(a.z, b1.z, c[2]);
  if (M0 == M_RES) c[1] = mfma4<Q % 16>(a.w, u0[Q].w, c[1]);
  if (M1 != M_NONE) c[3] = mfma4<Q % 16>(a.w, b1.w, c[3]);
  if (pinned) __builtin_amdgcn_sched_barrier(0);   // pin: the read of quad Q+LDS_AHEAD stays in front of quad Q
}
template <int KQ, int QR1, int M0, int M1, bool F16, size_t... Q>
__device__ __forceinline__ void strips_impl(const float4 (&u0)[KQ], const float4 (&u1)[QR1],
                                            const float* __restrict__ l1, const AVec<KQ>& a, f32x4& r0, f32x4& r1,
                                            std::index_sequence<Q...>) {
  f32x4 c[6];      // fp32: strip 0 -> c[0], c[1]; strip 1 -> c[2], c[3].  F16: strip 0 -> c[0 .. 2]; strip 1 -> c[3 .. 5]
#pragma unroll
  for (int i = 0; i < 6; ++i) c[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 ring[LDS_AHEAD];
  if (M1 == M_LDS) {
#pragma unroll
    for (int d = 0; d < LDS_AHEAD; ++d) ring[d] = *(const float4*)(l1 + (d < KQ ? d : 0) * 4);
  }
  (quad_step<KQ, QR1, (int)Q, M0, M1, F16>(a.v[Q / 16], u0, u1, l1, ring, c), ...);
  if (F16) {       // small terms first
    r0 = (c[0] + c[1]) + c[2];
    r1 = (c[3] + c[4]) + c[5];
  } else {
    r0 = c[0] + c[1];
    r1 = c[2] + c[3];
  }
}
// r0 = A . strip0, r1 = A . strip1 for the 4 rows (register j) at this lane's strip column.
// l1: M_RES -> this lane's slot-1 tail; M_LDS -> this lane's column of an LDS operand image.
template <int KQ, int QR1, int M0, int M1, bool F16 = false>
__device__ __forceinline__ void run_strips(const float4 (&u0)[KQ], const float4 (&u1)[QR1],
                                           const float* __restrict__ l1, const float* __restrict__ aimg, int lane,
                                           f32x4& r0, f32x4& r1) {
  AVec<KQ> a;
  a.load(aimg, lane);
  strips_impl<KQ, QR1, M0, M1, F16>(u0, u1, l1, a, r0, r1, std::make_index_sequence<KQ>{});
}
// LDS operand images are column-major [column][KQP][4] with KQP = KQ | 1 quads per column: the k-quads of a lane's
// column are at immediate offsets (no address arithmetic in the MFMA stream), and an odd quad count per column
// spreads the 16 lanes of every ds_read_b128 lane group over all 64 banks.
__host__ __device__ constexpr int kqp(int KQ) { return KQ | 1; }

// tanh through v_exp_f32 / v_rcp_f32: (1 - e) / (1 + e), e = exp(-2|x|); absolute error ~1e-7 (libm tanhf costs
// ~250 cycles per value in the step epilogue)
__device__ __forceinline__ float fast_tanh(float x) {
  const float e = __expf(-2.0f * fabsf(x));
  const float t = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
  return copysignf(t, x);
}
__device__ __forceinline__ float hs_grad4(float g) { return (g > 0.f && g < 1.f) ? 0.2f : 0.f; }

// A value that was handed to a global store stays in its register until `keep` is reached: hipcc makes a wave wait
// for the store's completion before it lets anything overwrite the store's data register, and the registers freed
// by an epilogue are exactly the ones the next MFMA stream (A vector, accumulators) takes.  Keeping the stored
// values alive over that stream moves the wait to a point where the store has long been acknowledged.
__device__ __forceinline__ void keep(float& v) { asm volatile("" : "+v"(v)); }

// Placement hint for the resident operand strips: ask for the value in an arch VGPR / an AGPR at this point.  MFMA
// reads its B operand from either file; left alone the allocator parks ~85 operand registers per step in AGPRs as
// spill slots and copies them back (v_accvgpr_read + a wait state) in front of their MFMA.
__device__ __forceinline__ void pin_v(float4& x) { asm volatile("" : "+v"(x.x), "+v"(x.y), "+v"(x.z), "+v"(x.w)); }
__device__ __forceinline__ void pin_a(float4& x) { asm volatile("" : "+a"(x.x), "+a"(x.y), "+a"(x.z), "+a"(x.w)); }
constexpr int kVgprQuads = 36;   // k-quads of the slot-0 strip kept in arch VGPRs (the rest + slot 1 live in AGPRs)

// Step masks of the 4 rows of this workgroup as wave-uniform bit words (bit i = step base+i), 64 steps at a time:
// one vector load + ballot per row and chunk instead of a byte load (and its wait) on every step.
struct RowMasks {
  unsigned long long bits[4];
  int chunk;
  const uint8_t* gm;
  uint32_t rbase[4];
  int W, lane, st;
  __device__ __forceinline__ void init(const uint8_t* gmask, int b0, int n, int W_, int lane_, int sb_, int st_) {
    gm = gmask; W = W_; lane = lane_; chunk = -1; st = st_;
#pragma unroll
    for (int j = 0; j < 4; ++j) rbase[j] = (uint32_t)min(b0 + j, n - 1) * (uint32_t)sb_;
  }
  __device__ __forceinline__ void seek(int t) {   // wave-uniform t
    if ((t >> 6) == chunk) return;
    chunk = t >> 6;
    const int tt = chunk * 64 + lane;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint8_t v = tt < W ? gm[rbase[j] + (uint32_t)(tt * st)] : (uint8_t)0;
      bits[j] = __ballot(v != 0);
    }
  }
  __device__ __forceinline__ bool at(int j, int t) const { return (bits[j] >> (t & 63)) & 1ull; }
};

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
int gru4_pack_kq(int H) { return gru4_kq(H); }
// behind the six images: a tail of kGruTailWords 32-bit words -- [0] sU (the power of two the pair images carry), [1] 1 / sU,
// and from word kWmaxStride on three rotating slots of max |U| (the scheme of the conv weight image, lstur_internal.h:
// wmax_word; convx.hip: convx_pack_w_kernel).  fp32 images ignore it.
size_t gru4_image_floats(int H) {
  const int KQ = gru4_kq(H);
  return KQ ? (size_t)KQ * 4 * 6 * H : 0;
}
size_t gru4_pack_floats(int H) { return gru4_kq(H) ? gru4_image_floats(H) + kGruTailWords : 0; }
bool gru4_f16_env() { return getenv("LSTUR_GRU_F32") == nullptr; }
// element U[k][c] of the recurrent matrix -> its place(s) in the images; f16: the fp16 pair of v * sU instead of the float
__device__ __forceinline__ void gru4_put(float* __restrict__ img, size_t group4, int j, float v, bool f16, float sU) {
  if (!f16) {
    img[group4 + j] = v;
    return;
  }
  store_pair((unsigned short*)img + 2 * group4 + j, v * sU);
}
__global__ __launch_bounds__(256) void gru4_pack_kernel(const float* __restrict__ U, int H, int KQ, float* __restrict__ img,
                                                        int f16, int seq, float grow) {
  const size_t e = (size_t)KQ * 4 * H;   // floats of an N = H image
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t* tail = (uint32_t*)(img + 6 * e);
  float sU = 1.0f;
  if (f16) {   // the scale of the previous pack's maximum (+ what one optimizer step can add), this pack's maximum for the next
    uint32_t m = tail[gru_wmax_word(seq % 3, threadIdx.x & (kWmaxLanes - 1))];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
    sU = pow2_scale(__float_as_uint(__uint_as_float(m) + grow), 13);
    if (i == 0) {
      ((float*)tail)[0] = sU;
      ((float*)tail)[1] = 1.0f / sU;
    }
    if (blockIdx.x == 0 && threadIdx.x < kWmaxLanes) tail[gru_wmax_word((seq + 2) % 3, threadIdx.x)] = 0u;
    // (this pack's own maximum: the caller's absmax launch wrote slot seq % 3 from U itself and mirrors it into the next slot)
  }
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
  gru4_put(img, i & ~(size_t)3, (int)(i & 3), v, f16 != 0 && i < 3 * e, sU);   // pairs: the FORWARD's images (izr, ih)
}
// Stand-alone pack (a first pack, or the caller changed the weights): max |U| from U itself, then the images.  The dense
// Adam launch writes the images in the training step (misc.hip: image_scatter) and keeps the rotating maxima going.
void launch_gru4_pack(const float* U, int H, float* img, hipStream_t s, bool f16, int seq) {
  const int KQ = gru4_kq(H);
  if (!KQ) return;
  const size_t total = gru4_image_floats(H);
  uint32_t* tail = (uint32_t*)(img + total);
  if (f16) {
    (void)hipMemsetAsync(tail, 0, (size_t)kGruTailWords * sizeof(uint32_t), s);
    launch_absmax(U, (size_t)3 * H * H, tail + gru_wmax_word(seq % 3, 0), s);
    (void)hipMemcpyAsync(tail + gru_wmax_word((seq + 1) % 3, 0), tail + gru_wmax_word(seq % 3, 0), sizeof(uint32_t),
                         hipMemcpyDeviceToDevice, s);
  }
  hipLaunchKernelGGL(gru4_pack_kernel, dim3((unsigned)cdivz(total, 256)), dim3(256), 0, s, U, H, KQ, img, f16 ? 1 : 0, seq, 0.f);
}

// tape of the fast path: row-major sections ZR [n*W,2H] | HH [n*W,H] | HP [n*W,H] | RH [n*W,H]
size_t gru4_tape_floats(int n, int W, int H) { return gru4_kq(H) ? (size_t)5 * n * W * H : 0; }


// ---- forward -------------------------------------------------------------------------------------
struct Gru4Fwd {
  const float* xw; const uint8_t* gmask; const float* h0; const float* izr; const float* ih; const float* tail;
  float* zr; float* hh; float* hp; float* rh; float* hlast;
  int n, W, H, dbg;   // dbg: measurement knobs (LSTUR_GRU4_DEBUG): 1 no global stores, 8 per-segment cycle stamps
  int sb, st;         // row of (sample b, step t) in xw / gmask / the tape = b * sb + t * st: (W, 1) sample-major, (1, n) slot-major
  GruScore sc;        // sc.d != NULL: the workgroup scores its samples behind the last step (lstur_internal.h)
};

// Strip plan of the forward kernel (host + device).  [z|r] strip s and s+4 are resident on wave s.  A wave whose
// second register slot is free (s+4 >= G1) can keep a candidate strip resident: strips 0..nres-1 sit on waves
// 3, 2, ...; the other candidate strips (hidden columns nres*64 .. H-1, width wl) are LDS-fed on waves 0, 1, ...
struct FwdPlan {
  int G1, G2, nres, wl;
  __host__ __device__ FwdPlan(int H) {
    G1 = cdiv(2 * H, 64);
    G2 = cdiv(H, 64);
    const int nfree = G1 >= 4 ? 8 - G1 : 4;
    nres = nfree < G2 ? nfree : G2;
    wl = H - nres * 64 > 0 ? H - nres * 64 : 0;
  }
};
static size_t gru4_fwd_lds(int H, int KQ) {
  const FwdPlan p(H);
  return ((size_t)4 * KQ * 16 + 32 + (size_t)kqp(KQ) * p.wl * 4 + (size_t)4 * (KQ - fwd_slot1(KQ)) * 256) * sizeof(float);
}

// hA: the state as floats (element-wise reads); F16: h16 = the same state as the MFMA operand (fp16 pairs of h sA), rhA holds
// pairs of r h sA; sA / inv: the operand scale of each of this workgroup's four rows and 1 / (sA sU) -- per ROW: a sample's
// result must not depend on which samples share its workgroup
struct FwdLds { float *hA, *rhA, *zA, *h16, *dummy; const float* uL; int wl; const float* tail; float sA[4], inv[4]; };

// Time loop of one wave role.  NP1: resident [z|r] strips of this wave (slot 0 = strip wv, slot 1 = strip wv+4).
// P2: candidate strip M_NONE | M_RES (slot 1) | M_LDS; c2 = its first hidden column, lcol = its first image column.
template <int KQ, int NP1, int P2, bool F16>
__device__ __forceinline__ void gru4_fwd_loop(const Gru4Fwd& g, const FwdLds& L, int wv, int lane, int c2base, int lcol,
                                              const float4 (&u0)[KQ], const float4 (&u1)[fwd_slot1(KQ)]) {
  const int H = g.H, W = g.W, n = g.n, H2 = 2 * H, H3 = 3 * H;
  const int b0 = blockIdx.x * 4;
  const uint32_t e4 = (uint32_t)n * W * H * 4;   // bytes of an [n*W, H] tensor
  const __amdgpu_buffer_rsrc_t rxw = make_rsrc(g.xw, 3 * e4), rzr = make_rsrc(g.zr, 2 * e4), rhh = make_rsrc(g.hh, e4),
                               rhp = make_rsrc(g.hp, e4), rrh = make_rsrc(g.rh, e4);
  // phase-1 strips: per-lane column, LDS destination of the gate (z -> zA, r*h -> rhA, past 2H -> dummy), h source
  uint32_t o1[2], o1r[2];
  float* dst1[2];
  unsigned short* dst1p[2];      // F16: where an r lane puts the pair of r h sA
  const float* hsrc[2];
  bool isr[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int c = (wv + 4 * s) * 64 + lane;
    const bool v = s < NP1 && c < H2;
    isr[s] = c >= H;
    const int hc = isr[s] ? c - H : c;
    o1[s] = v ? c * 4 : kBufOob;
    o1r[s] = (v && isr[s]) ? hc * 4 : kBufOob;
    dst1[s] = v ? (isr[s] ? (F16 ? L.dummy : L.rhA + aidx(hc, 0)) : L.zA + aidx(hc, 0)) : L.dummy;
    dst1p[s] = (v && isr[s]) ? pair_ptr(L.rhA, hc) : (unsigned short*)L.dummy;
    hsrc[s] = L.hA + aidx(v ? hc : 0, 0);
  }
  const int c2 = c2base + lane;             // hidden column of the candidate strip
  const bool v2 = P2 != M_NONE && c2 < H;
  const uint32_t o2 = v2 ? c2 * 4 : kBufOob, o2x = v2 ? (H2 + c2) * 4 : kBufOob;
  const float* zsrc = L.zA + aidx(v2 ? c2 : 0, 0);
  const float* hsrc2 = L.hA + aidx(v2 ? c2 : 0, 0);
  float* hdst2 = v2 ? L.hA + aidx(c2, 0) : L.dummy;
  unsigned short* hdst2p = v2 ? pair_ptr(L.h16, c2) : (unsigned short*)L.dummy;
  float sA[4], inv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { sA[j] = L.sA[j]; inv[j] = L.inv[j]; }
  const float* uLl = L.uL + (size_t)(P2 == M_LDS ? min(lcol + lane, L.wl - 1) : 0) * kqp(KQ) * 4;
  uint32_t rb[4];   // row * W (the launcher keeps n*W*3H below 2^30)
  bool rok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    rok[j] = b0 + j < n && !(g.dbg & 1);
    rb[j] = (uint32_t)min(b0 + j, n - 1) * (uint32_t)g.sb;
  }
  const uint32_t ST = (uint32_t)g.st;
  RowMasks rm;
  if (P2 != M_NONE) rm.init(g.gmask, b0, n, W, lane, g.sb, g.st);
  // measurement build only (dbg bit 3): per-segment shader cycles of this wave, summed over the steps
  const bool stamp = g.dbg & 8;
  unsigned long long st[6] = {0, 0, 0, 0, 0, 0}, last = 0;
#define LSTUR_STAMP(k)                                     \
  if (stamp) {                                             \
    unsigned long long now = __builtin_amdgcn_s_memtime(); \
    st[k] += now - last;                                   \
    last = now;                                            \
  }
  float k1[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, k1r[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  float k2h[4] = {0.f, 0.f, 0.f, 0.f}, k2p[4] = {0.f, 0.f, 0.f, 0.f};   // values handed to stores (see keep())
  for (int t = 0; t < W; ++t) {
    if (stamp) last = __builtin_amdgcn_s_memtime();
    // ---- phase 1: z | r ----
    if (NP1 > 0) {
      float x1[2][4];
#pragma unroll
      for (int s = 0; s < NP1; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) x1[s][j] = buf_load1(rxw, o1[s], (rb[j] + (uint32_t)t * ST) * H3 * 4);
      float hv[2][4];   // h at this lane's r columns (read before the stream: one LDS round trip less in the epilogue)
#pragma unroll
      for (int s = 0; s < NP1; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) hv[s][j] = hsrc[s][j * 4];
      __builtin_amdgcn_sched_barrier(0);   // keep the gate-input loads in front of the MFMA stream
      f32x4 acc[2];
      run_strips<KQ, fwd_slot1(KQ), M_RES, NP1 == 2 ? M_RES : M_NONE, F16>(u0, u1, L.tail, F16 ? L.h16 : L.hA, lane, acc[0], acc[1]);
      if (P2 != M_NONE) {   // the previous step's hh / hp stores have long completed
#pragma unroll
        for (int j = 0; j < 4; ++j) { keep(k2h[j]); keep(k2p[j]); }
      }
      LSTUR_STAMP(0)
#pragma unroll
      for (int s = 0; s < NP1; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float gt = hard_sigmoid((F16 ? acc[s][j] * inv[j] : acc[s][j]) + x1[s][j]);
          const float val = isr[s] ? gt * hv[s][j] : gt;
          dst1[s][j * 4] = val;
          if (F16) store_pair<false>(dst1p[s] + j * 8, val * sA[j]);     // (z lanes and lanes past 2H: into the sink; |r h sA| <= 2^13)
          k1[s][j] = gt;
          k1r[s][j] = val;
          buf_store1(k1[s][j], rzr, rok[j] ? o1[s] : kBufOob, (rb[j] + (uint32_t)t * ST) * H2 * 4);
          buf_store1(k1r[s][j], rrh, rok[j] ? o1r[s] : kBufOob, (rb[j] + (uint32_t)t * ST) * H * 4);
        }
    }
    float x2[4];
    if (P2 != M_NONE) {   // candidate inputs fly over the barrier and the candidate product
      rm.seek(t);
#pragma unroll
      for (int j = 0; j < 4; ++j) x2[j] = buf_load1(rxw, o2x, (rb[j] + (uint32_t)t * ST) * H3 * 4);
    }
    LSTUR_STAMP(1)
    __syncthreads();
    LSTUR_STAMP(2)
    // ---- phase 2: candidate + state update ----
    if (P2 != M_NONE) {
      float z[4], hp[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        z[j] = zsrc[j * 4];
        hp[j] = hsrc2[j * 4];
      }
      __builtin_amdgcn_sched_barrier(0);
      f32x4 acc, unused;
      run_strips<KQ, fwd_slot1(KQ), M_NONE, P2, F16>(u0, u1, P2 == M_RES ? L.tail : uLl, L.rhA, lane, unused, acc);
      LSTUR_STAMP(3)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float hhv = fast_tanh((F16 ? acc[j] * inv[j] : acc[j]) + x2[j]);
        const float hn = z[j] * hp[j] + (1.0f - z[j]) * hhv;
        const float hst = rm.at(j, t) ? hn : hp[j];   // K.rnn: a masked step carries the previous state/output
        hdst2[j * 4] = hst;
        if (F16) store_pair<false>(hdst2p + j * 8, hst * sA[j]);
        k2h[j] = hhv;
        k2p[j] = hp[j];
        buf_store1(k2h[j], rhh, rok[j] ? o2 : kBufOob, (rb[j] + (uint32_t)t * ST) * H * 4);
        buf_store1(k2p[j], rhp, rok[j] ? o2 : kBufOob, (rb[j] + (uint32_t)t * ST) * H * 4);
      }
    }
    if (NP1 > 0) {   // this step's z / r / r*h stores were issued a whole candidate product ago
#pragma unroll
      for (int s = 0; s < NP1; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) { keep(k1[s][j]); keep(k1r[s][j]); }
    }
    LSTUR_STAMP(4)
    __syncthreads();
    LSTUR_STAMP(5)
  }
#undef LSTUR_STAMP
  if (stamp && lane == 0 && blockIdx.x == 0)
    for (int k = 0; k < 6; ++k) g.hlast[wv * 8 + k] = (float)st[k];
}

template <int KQ, bool F16>
__global__ __launch_bounds__(G4_THREADS) void gru4_fwd_kernel(Gru4Fwd g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = g.H, n = g.n, H2 = 2 * H;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const FwdPlan P(H);
  FwdLds L;
  L.hA = smem;                      // state, A image
  L.rhA = L.hA + KQ * 16;           // r*h, A image
  L.zA = L.rhA + KQ * 16;           // z, same layout
  L.h16 = L.zA + KQ * 16;           // F16: the state as the MFMA operand (pairs)
  L.dummy = L.h16 + KQ * 16;        // sink for lanes past the last column (32 floats: pair stores reach halfword 28)
  float* uL = L.dummy + 32;         // U_h[:, nres*64 : H] image [wl][KQP][4]
  L.uL = uL;
  L.wl = P.wl;
  constexpr int QR1 = fwd_slot1(KQ);
  float* tailL = uL + (size_t)kqp(KQ) * P.wl * 4 + (size_t)wv * (KQ - QR1) * 256 + lane * 4;   // slot-1 tail of this lane
  L.tail = tailL;
  const int b0 = blockIdx.x * 4;

  for (int i = tid; i < 4 * KQ * 16 + 32; i += G4_THREADS) smem[i] = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) L.sA[j] = L.inv[j] = 1.0f;
  if (F16) {   // operand scale of every row: max(1, max |h0| of the row) -> 2^12 (|h_t| never exceeds that maximum)
    float m[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    for (int i = tid; i < 4 * H; i += G4_THREADS) {
      const int row = i / H, col = i - row * H, b = b0 + row;
      const float v = (g.h0 && b < n) ? fabsf(g.h0[(size_t)b * H + col]) : 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) m[j] = row == j ? fmaxf(m[j], v) : m[j];
    }
    float* red = smem + 4 * KQ * 16 + 16;       // (second half of the sink: zeroed above, cleared again below; 16 floats)
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      m[j] = wave_max(m[j]);
      if (lane == 0) red[wv * 4 + j] = m[j];
    }
    __syncthreads();
    const float sU = g.tail[0];                             // the pack's tail: [0] sU
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float mj = fmaxf(fmaxf(red[j], red[4 + j]), fmaxf(red[8 + j], red[12 + j]));
      L.sA[j] = pow2_scale(__float_as_uint(mj), 12);
      L.inv[j] = 1.0f / (L.sA[j] * sU);
    }
    __syncthreads();
    if (tid < 16) red[tid] = 0.f;
  }
  __syncthreads();
  for (int i = tid; i < 4 * H; i += G4_THREADS) {
    const int row = i / H, col = i - row * H, b = b0 + row;
    if (g.h0 && b < n) {
      const float v = g.h0[(size_t)b * H + col];
      L.hA[aidx(col, row)] = v;
      if (F16) store_pair(pair_ptr(L.h16, col) + row * 8, v * (row == 0 ? L.sA[0] : row == 1 ? L.sA[1] : row == 2 ? L.sA[2] : L.sA[3]));
    }
  }
  for (int i = tid; i < KQ * P.wl; i += G4_THREADS) {
    const int q = i / P.wl, col = i - q * P.wl;
    ((float4*)uL)[col * kqp(KQ) + q] = ((const float4*)g.ih)[(size_t)q * H + P.nres * 64 + col];
  }

  // roles (all wave-uniform)
  const int np1 = (wv < P.G1) + (wv + 4 < P.G1);
  int p2 = M_NONE, c2base = 0, lcol = 0;
  if (wv >= 4 - P.nres) { p2 = M_RES; c2base = (3 - wv) * 64; }
  else if (wv < P.G2 - P.nres) { p2 = M_LDS; lcol = wv * 64; c2base = (P.nres + wv) * 64; }
  float4 u0[KQ], u1[QR1];
  {
    const int ca = wv * 64 + lane;                 // slot 0: [z|r] strip wv
    const bool oka = np1 >= 1 && ca < H2;
    const int cb = p2 == M_RES ? c2base + lane : (wv + 4) * 64 + lane;   // slot 1: [z|r] strip wv+4 or candidate strip
    const bool okb = p2 == M_RES ? cb < H : (np1 == 2 && cb < H2);
    const float4* sb = p2 == M_RES ? (const float4*)g.ih : (const float4*)g.izr;
    const int nb = p2 == M_RES ? H : H2;
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const float4 va = ((const float4*)g.izr)[(size_t)q * H2 + (oka ? ca : 0)];
      float4 vb = sb[(size_t)q * nb + (okb ? cb : 0)];
      vb = okb ? vb : make_float4(0.f, 0.f, 0.f, 0.f);
      u0[q] = oka ? va : make_float4(0.f, 0.f, 0.f, 0.f);
      if (q < QR1) u1[q < QR1 ? q : 0] = vb;
      else *(float4*)(tailL + (q - QR1) * 256) = vb;
    }
    if (KQ > 40) {
#pragma unroll
      for (int q = 0; q < KQ; ++q)
        if (q < kVgprQuads) pin_v(u0[q]); else pin_a(u0[q]);
#pragma unroll
      for (int q = 0; q < QR1; ++q) pin_a(u1[q]);
    }
  }
  __syncthreads();
#define LSTUR_ROLE(A, B) \
  if (np1 == A && p2 == B) gru4_fwd_loop<KQ, A, B, F16>(g, L, wv, lane, c2base, lcol, u0, u1); else
  LSTUR_ROLE(2, M_LDS) LSTUR_ROLE(2, M_NONE) LSTUR_ROLE(1, M_RES) LSTUR_ROLE(1, M_NONE) LSTUR_ROLE(0, M_RES)
  gru4_fwd_loop<KQ, 0, M_NONE, F16>(g, L, wv, lane, c2base, lcol, u0, u1);
#undef LSTUR_ROLE

  for (int i = tid; i < 4 * H; i += G4_THREADS) {
    const int row = i / H, col = i - row * H, b = b0 + row;
    if (b < n && !(g.dbg & 8)) g.hlast[(size_t)b * H + col] = L.hA[aidx(col, row)];
  }
  // LSTUR-ini: the last state IS the user vector -- wave j scores sample b0 + j against its candidates here (the scorer's
  // own kernel was 10 us of launch and dependent loads on the step's critical path for 64 x 5 dot products): the same
  // arithmetic in the same order, the state read from its LDS image instead of the row just written
  if (g.sc.d && b0 + wv < n && H <= 256) {
    float uv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) uv[c] = lane + 64 * c < H ? L.hA[aidx(lane + 64 * c, wv)] : 0.f;
    score_row<8, 4>(uv, g.sc.d, g.sc.logits, g.sc.probs, g.sc.loss_i, g.sc.correct_i, g.sc.du, g.sc.dd, b0 + wv, lane, g.sc.K1, H,
                    g.sc.inv_n);
  }
}

// ---- BPTT ----------------------------------------------------------------------------------------
//   dhn = m ? dh : 0 ; carry = m ? 0 : dh
//   dpre_h = dhn (1-z)(1-hh^2) ; dpre_z = dhn (hp-hh) hs'(z)
//   drh = dpre_h U_h^T ; dpre_r = drh hp hs'(r)
//   dh_prev = carry + dhn z + drh r + dpre_z U_z^T + dpre_r U_r^T
// Strip j of a product covers hidden columns 64j .. 64j+63 (G2 strips).
//   stage 1: wave 0: drh 0, 1      wave 1: part_z 0, 1      wave 2: drh 2, 3      wave 3: part_z 2, 3
//   stage 2 (dpre_r U_r^T + the element-wise head of the next step): strip 0 on wave 2, 1 on wave 3, 2 on wave 0,
//            3 on wave 1
//   resident: both stage-1 strips of waves 0/1, strip 2 of waves 2/3 (slot 0) and the stage-2 strip of waves 2/3
//   (slot 1); from LDS: the stage-1 tail strips 3 (columns 192..H-1) and the stage-2 strips 2 and 3.
struct Gru4Bwd {
  const float* dh_last; const uint8_t* gmask; const float* zr; const float* hh; const float* hp;
  const float* itz; const float* itr; const float* ith;
  float* dgates; float* dh0;
  int n, W, H, dbg;
  int sb, st;         // row of (sample b, step t) = b * sb + t * st (like Gru4Fwd)
  int t_hi, t_lo;     // this launch walks steps t_hi .. t_lo (W - 1 .. 0: the whole recurrence); dh_last = dh behind step t_hi,
};                    // dh0 = dh in front of step t_lo
static size_t gru4_bwd_lds(int H, int KQ) {
  const int G2 = cdiv(H, 64);
  const int w2 = G2 >= 3 ? std::min(64, H - 128) : 0, w3 = G2 >= 4 ? H - 192 : 0;
  return ((size_t)5 * KQ * 16 + 16 + (size_t)kqp(KQ) * 4 * (w2 + 3 * w3) + (size_t)4 * (KQ - bwd_slot1(KQ)) * 256) *
         sizeof(float);
}
struct BwdLds {
  float *dphA, *dpzA, *dprA, *n2, *n3, *dummy;
  const float* tail;        // this lane's LDS tail of the slot-1 strip
  const float *la, *lb;     // LDS operand images of this wave's stage-1 tail strip / stage-2 strip (lane-adjusted)
};

// Time loop of one wave role.  KIND 1: the stage-1 strips are drh (A = dpre_h), KIND 2: part_z (A = dpre_z).
// MA0: M_NONE | M_RES (slot 0) strip ga; MA1: M_NONE | M_RES (slot 1) | M_LDS strip ga+1;
// MB: stage-2 strip gb, M_NONE | M_RES (slot 1) | M_LDS.
template <int KQ, int KIND, int MA0, int MA1, int MB>
__device__ __forceinline__ void gru4_bwd_loop(const Gru4Bwd& g, const BwdLds& L, int ga, int gb, int lane,
                                              const float4 (&u0)[KQ], const float4 (&u1)[bwd_slot1(KQ)]) {
  const int H = g.H, W = g.W, n = g.n, H2 = 2 * H, H3 = 3 * H;
  const int b0 = blockIdx.x * 4;
  const uint32_t e4 = (uint32_t)n * W * H * 4;   // bytes of an [n*W, H] tensor
  const __amdgpu_buffer_rsrc_t rzr = make_rsrc(g.zr, 2 * e4), rhh = make_rsrc(g.hh, e4), rhp = make_rsrc(g.hp, e4),
                               rdg = make_rsrc(g.dgates, 3 * e4);
  constexpr int NA = (MA0 != M_NONE) + (MA1 != M_NONE);
  // stage-1 strips
  uint32_t oa[2], oar[2];
  float *dsta[2], *dstn[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int c = (ga + s) * 64 + lane;
    const bool v = (s == 0 ? MA0 != M_NONE : MA1 != M_NONE) && c < H;
    oa[s] = v ? c * 4 : kBufOob;
    oar[s] = v ? (H + c) * 4 : kBufOob;
    dsta[s] = v ? L.dprA + aidx(c, 0) : L.dummy;
    dstn[s] = v ? (KIND == 1 ? L.n2 : L.n3) + aidx(c, 0) : L.dummy;
  }
  // stage-2 strip
  const int cb = gb * 64 + lane;
  const bool vb = MB != M_NONE && cb < H;
  const uint32_t ob = vb ? cb * 4 : kBufOob, obh = vb ? (H2 + cb) * 4 : kBufOob;
  const float* n2s = L.n2 + aidx(vb ? cb : 0, 0);
  const float* n3s = L.n3 + aidx(vb ? cb : 0, 0);
  float* dphd = vb ? L.dphA + aidx(cb, 0) : L.dummy;
  float* dpzd = vb ? L.dpzA + aidx(cb, 0) : L.dummy;
  uint32_t rb[4];   // row * W (the launcher keeps n*W*3H below 2^30)
  bool rok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    rok[j] = b0 + j < n && !(g.dbg & 1);
    rb[j] = (uint32_t)min(b0 + j, n - 1) * (uint32_t)g.sb;
  }
  const uint32_t ST = (uint32_t)g.st;
  RowMasks rm;
  if (MB != M_NONE) rm.init(g.gmask, b0, n, W, lane, g.sb, g.st);

  // element-wise head of step t for the (row, column) elements this lane owns in stage 2
  float n1[4] = {0.f, 0.f, 0.f, 0.f};
  float kz[4] = {0.f, 0.f, 0.f, 0.f}, kh[4] = {0.f, 0.f, 0.f, 0.f};   // values handed to stores (see keep())
  float kr[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  auto head = [&](int t, const float (&dh)[4], const float (&z)[4], const float (&hh)[4], const float (&hp)[4]) {
    rm.seek(t);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool m = rm.at(j, t);
      const float dhn = m ? dh[j] : 0.f;
      const float dph = dhn * (1.0f - z[j]) * (1.0f - hh[j] * hh[j]);
      const float dpz = dhn * (hp[j] - hh[j]) * hs_grad4(z[j]);
      n1[j] = (m ? 0.f : dh[j]) + dhn * z[j];
      dphd[j * 4] = dph;
      dpzd[j * 4] = dpz;
      kz[j] = dpz;
      kh[j] = dph;
      buf_store1(kz[j], rdg, rok[j] ? ob : kBufOob, (rb[j] + (uint32_t)t * ST) * H3 * 4);
      buf_store1(kh[j], rdg, rok[j] ? obh : kBufOob, (rb[j] + (uint32_t)t * ST) * H3 * 4);
    }
  };
  if (MB != M_NONE) {
    float dh[4], z[4], hh[4], hp[4];
    const int t = g.t_hi;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      dh[j] = (b0 + j < n && vb) ? g.dh_last[(size_t)(b0 + j) * H + cb] : 0.f;
      z[j] = buf_load1(rzr, ob, (rb[j] + (uint32_t)t * ST) * H2 * 4);
      hh[j] = buf_load1(rhh, ob, (rb[j] + (uint32_t)t * ST) * H * 4);
      hp[j] = buf_load1(rhp, ob, (rb[j] + (uint32_t)t * ST) * H * 4);
    }
    head(t, dh, z, hh, hp);
  }
  __syncthreads();

  for (int t = g.t_hi; t >= g.t_lo; --t) {
    // ---- stage 1 ----
    if (NA > 0) {
      float r[2][4], hp[2][4];
      if (KIND == 1) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            r[s][j] = buf_load1(rzr, oar[s], (rb[j] + (uint32_t)t * ST) * H2 * 4);
            hp[s][j] = buf_load1(rhp, oa[s], (rb[j] + (uint32_t)t * ST) * H * 4);
          }
        __builtin_amdgcn_sched_barrier(0);   // keep the loads in front of the MFMA stream
      }
      f32x4 acc[2];
      run_strips<KQ, bwd_slot1(KQ), MA0, MA1>(u0, u1, MA1 == M_RES ? L.tail : L.la, KIND == 1 ? L.dphA : L.dpzA, lane, acc[0], acc[1]);
      if (MB != M_NONE) {   // the head's dz / dh stores were issued a whole product ago
#pragma unroll
        for (int j = 0; j < 4; ++j) { keep(kz[j]); keep(kh[j]); }
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (s == 0 ? MA0 == M_NONE : MA1 == M_NONE) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (KIND == 1) {
            const float drh = acc[s][j];
            const float dpr = drh * hp[s][j] * hs_grad4(r[s][j]);
            dsta[s][j * 4] = dpr;
            dstn[s][j * 4] = drh * r[s][j];
            kr[s][j] = dpr;
            buf_store1(kr[s][j], rdg, rok[j] ? oar[s] : kBufOob, (rb[j] + (uint32_t)t * ST) * H3 * 4);
          } else {
            dstn[s][j * 4] = acc[s][j];
          }
        }
      }
    } else if (MB != M_NONE) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { keep(kz[j]); keep(kh[j]); }
    }
    __syncthreads();
    // ---- stage 2 (+ the element-wise head of step t-1) ----
    if (MB != M_NONE) {
      float z[4], hh[4], hp[4];
      const int tp = max(t - 1, g.t_lo);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        z[j] = buf_load1(rzr, ob, (rb[j] + (uint32_t)tp * ST) * H2 * 4);
        hh[j] = buf_load1(rhh, ob, (rb[j] + (uint32_t)tp * ST) * H * 4);
        hp[j] = buf_load1(rhp, ob, (rb[j] + (uint32_t)tp * ST) * H * 4);
      }
      __builtin_amdgcn_sched_barrier(0);
      f32x4 acc, unused;
      run_strips<KQ, bwd_slot1(KQ), M_NONE, MB>(u0, u1, MB == M_RES ? L.tail : L.lb, L.dprA, lane, unused, acc);
      float dh[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) dh[j] = n1[j] + n2s[j * 4] + n3s[j * 4] + acc[j];
      if (t > g.t_lo) {
        head(t - 1, dh, z, hh, hp);
      } else if (vb) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (b0 + j < n) g.dh0[(size_t)(b0 + j) * H + cb] = dh[j];
      }
    }
    if (KIND == 1 && NA > 0) {   // this step's dr stores were issued a whole product ago
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) keep(kr[s][j]);
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
  BwdLds L;
  L.dphA = smem;                    // dpre_h, A image
  L.dpzA = L.dphA + KQ * 16;        // dpre_z, A image
  L.dprA = L.dpzA + KQ * 16;        // dpre_r, A image
  L.n2 = L.dprA + KQ * 16;          // drh * r          (same layout)
  L.n3 = L.n2 + KQ * 16;            // dpre_z . U_z^T
  L.dummy = L.n3 + KQ * 16;
  constexpr int KQP = kqp(KQ);
  float* Lr2 = L.dummy + 16;                  // itr[:, 128:128+w2]   [w2][KQP][4]
  float* Lh3 = Lr2 + (size_t)KQP * w2 * 4;    // ith[:, 192:H]
  float* Lz3 = Lh3 + (size_t)KQP * w3 * 4;    // itz[:, 192:H]
  float* Lr3 = Lz3 + (size_t)KQP * w3 * 4;    // itr[:, 192:H]
  constexpr int QR1 = bwd_slot1(KQ);
  float* tailL = Lr3 + (size_t)KQP * w3 * 4 + (size_t)wv * (KQ - QR1) * 256 + lane * 4;   // slot-1 tail of this lane
  L.tail = tailL;

  for (int i = tid; i < 5 * KQ * 16 + 16; i += G4_THREADS) smem[i] = 0.f;
  for (int i = tid; i < KQ * w2; i += G4_THREADS) {
    const int q = i / w2, c = i - q * w2;
    ((float4*)Lr2)[c * KQP + q] = ((const float4*)g.itr)[(size_t)q * H + 128 + c];
  }
  for (int i = tid; i < KQ * w3; i += G4_THREADS) {
    const int q = i / w3, c = i - q * w3;
    ((float4*)Lh3)[c * KQP + q] = ((const float4*)g.ith)[(size_t)q * H + 192 + c];
    ((float4*)Lz3)[c * KQP + q] = ((const float4*)g.itz)[(size_t)q * H + 192 + c];
    ((float4*)Lr3)[c * KQP + q] = ((const float4*)g.itr)[(size_t)q * H + 192 + c];
  }

  // ---- roles (wave-uniform) ----
  const int kind = (wv & 1) ? 2 : 1;           // waves 0, 2: drh; waves 1, 3: part_z
  const int ga = wv < 2 ? 0 : 2;               // first stage-1 strip
  const int gb = wv >= 2 ? wv - 2 : wv + 2;    // stage-2 strip
  const int ma0 = ga < G2 ? M_RES : M_NONE;
  const int ma1 = ga + 1 < G2 ? (wv < 2 ? M_RES : M_LDS) : M_NONE;
  const int mb = gb < G2 ? (wv >= 2 ? M_RES : M_LDS) : M_NONE;
  L.la = (kind == 1 ? Lh3 : Lz3) + (size_t)min(lane, max(w3 - 1, 0)) * KQP * 4;
  L.lb = gb == 2 ? Lr2 + (size_t)min(lane, max(w2 - 1, 0)) * KQP * 4 : Lr3 + (size_t)min(lane, max(w3 - 1, 0)) * KQP * 4;

  float4 u0[KQ], u1[QR1];
  {
    const float4* simg = (const float4*)(kind == 1 ? g.ith : g.itz);
    const int ca = ga * 64 + lane;
    const bool oka = ma0 == M_RES && ca < H;
    // slot 1: second stage-1 strip (waves 0, 1) or the stage-2 strip (waves 2, 3)
    const float4* sb = wv < 2 ? simg : (const float4*)g.itr;
    const int cbb = wv < 2 ? (ga + 1) * 64 + lane : gb * 64 + lane;
    const bool okb = (wv < 2 ? ma1 == M_RES : mb == M_RES) && cbb < H;
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const float4 va = simg[(size_t)q * H + (oka ? ca : 0)];
      float4 vb = sb[(size_t)q * H + (okb ? cbb : 0)];
      vb = okb ? vb : make_float4(0.f, 0.f, 0.f, 0.f);
      u0[q] = oka ? va : make_float4(0.f, 0.f, 0.f, 0.f);
      if (q < QR1) u1[q < QR1 ? q : 0] = vb;
      else *(float4*)(tailL + (q - QR1) * 256) = vb;
    }
    if (KQ > 40) {
#pragma unroll
      for (int q = 0; q < KQ; ++q)
        if (q < kVgprQuads) pin_v(u0[q]); else pin_a(u0[q]);
#pragma unroll
      for (int q = 0; q < QR1; ++q) pin_a(u1[q]);
    }
  }
  __syncthreads();

#define LSTUR_ROLE(K, A0, A1, B) \
  if (kind == K && ma0 == A0 && ma1 == A1 && mb == B) gru4_bwd_loop<KQ, K, A0, A1, B>(g, L, ga, gb, lane, u0, u1); else
#define LSTUR_ROLES(K)                                                                                      \
  LSTUR_ROLE(K, M_RES, M_RES, M_LDS) LSTUR_ROLE(K, M_RES, M_RES, M_NONE) LSTUR_ROLE(K, M_RES, M_NONE, M_NONE) \
  LSTUR_ROLE(K, M_RES, M_LDS, M_RES) LSTUR_ROLE(K, M_RES, M_NONE, M_RES) LSTUR_ROLE(K, M_NONE, M_NONE, M_RES)
  LSTUR_ROLES(1) LSTUR_ROLES(2)
  gru4_bwd_loop<KQ, 1, M_NONE, M_NONE, M_NONE>(g, L, ga, gb, lane, u0, u1);
#undef LSTUR_ROLES
#undef LSTUR_ROLE
}

// measurement knobs (wrong results: no stores / cycle stamps) exist in the bench-only library alone (-DLSTUR_ABLATE)
static int gru4_dbg() {
#ifdef LSTUR_ABLATE
  const char* e = getenv("LSTUR_GRU4_DEBUG");
  return e ? atoi(e) : 0;
#else
  return 0;
#endif
}

// ---- launchers -------------------------------------------------------------------------------------
template <typename K>
static void set_lds(K kern, size_t lds) {
  if (lds > 65536) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

bool gru4_supported(int H) {
  const int KQ = gru4_kq(H);
  if (!KQ) return false;
  const FwdPlan p(H);
  return p.G1 <= 7 && p.G2 <= 4 && gru4_fwd_lds(H, KQ) <= G4_LDS_MAX && gru4_bwd_lds(H, KQ) <= G4_LDS_MAX;
}

void gru4_tape_rows(float* tape, int n, int W, int H, float** hprev, float** rh) {
  const size_t e = (size_t)n * W * H;
  *hprev = tape + 3 * e;
  *rh = tape + 4 * e;
}

void launch_gru4_fwd(const float* xw, const uint8_t* gmask, const float* h0, const float* img, float* tape, float* hlast,
                     int n, int W, int H, hipStream_t s, bool f16, const GruScore* sc) {
  const int KQ = gru4_kq(H);
  const size_t e = (size_t)n * W * H, ie = (size_t)KQ * 4 * H;
  Gru4Fwd a{xw, gmask, h0, img, img + 2 * ie, img + 6 * ie, tape, tape + 2 * e, tape + 3 * e, tape + 4 * e, hlast, n, W, H, gru4_dbg(),
            W, 1, sc ? *sc : GruScore{}};
  const size_t lds = gru4_fwd_lds(H, KQ);
  const dim3 grid(cdiv(n, 4)), block(G4_THREADS);
#define LSTUR_G4F(Q)                                                            \
  {                                                                             \
    if (f16) {                                                                  \
      set_lds(gru4_fwd_kernel<Q, true>, lds);                                   \
      launch_k((gru4_fwd_kernel<Q, true>), grid, block, lds, s, a);             \
    } else {                                                                    \
      set_lds(gru4_fwd_kernel<Q, false>, lds);                                  \
      launch_k((gru4_fwd_kernel<Q, false>), grid, block, lds, s, a);            \
    }                                                                           \
  }
  if (KQ == 10) LSTUR_G4F(10) else if (KQ == 32) LSTUR_G4F(32) else LSTUR_G4F(50)
#undef LSTUR_G4F
}

void launch_gru4_bwd(const float* dh_last, const uint8_t* gmask, float* tape, const float* img, float* dgates, float* rh,
                     float* hprev, float* dh0, int n, int W, int H, hipStream_t s) {
  const int KQ = gru4_kq(H);
  const size_t e = (size_t)n * W * H, ie = (size_t)KQ * 4 * H;
  Gru4Bwd a{dh_last, gmask, tape, tape + 2 * e, tape + 3 * e, img + 3 * ie, img + 4 * ie, img + 5 * ie, dgates, dh0,
            n, W, H, gru4_dbg(), W, 1, W - 1, 0};
  const size_t lds = gru4_bwd_lds(H, KQ);
  const dim3 grid(cdiv(n, 4)), block(G4_THREADS);
#define LSTUR_G4B(Q)                                              \
  {                                                               \
    set_lds(gru4_bwd_kernel<Q>, lds);                             \
    launch_k(gru4_bwd_kernel<Q>, grid, block, lds, s, a);          \
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
