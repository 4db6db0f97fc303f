// Conv1D('same', relu) x (id != 0) of the cnnatt news encoder (task/paper.py:146-155) as an fp32-GRADE contraction on
// the bf16 matrix cores of gfx950 ("bf16x3"): every fp32 operand is split into three bf16 parts h + m + l (split3,
// bf16_mfma.h: exact to ~2^-25) and the six products hh + hm + mh + hl + lh + mm are accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16.  The dropped terms (ml, lm, ll) are below 2^-24 of |a||b|, i.e. the result carries the
// same error as an fp32 FMA chain in another summation order -- it is NOT a reduced-precision kernel (tests hold it
// to the same tolerances as the exact-fp32 MFMA kernels of conv.hip, and to 2e-6 against them directly).
// The fp32 MFMA (v_mfma_f32_16x16x4_f32) runs at 1/16 of the bf16 rate, so six bf16 products cost 6/16 of the time.
//
// Forward: implicit im2col -- the row of token (n, t) is the contiguous window E_flat[(n*Tp + t)*D .. + k*D) of the
// zero-padded embedding layout -- over the compacted list of non-pad tokens.  One workgroup = 128 token rows x ALL F
// columns (<= 448), 8 waves as 4 (rows) x 2 (7 + up to 7 column tiles of 32); BK = 16: per k-step a wave reads one A
// fragment per split part (ds_read_b128) and the B fragments by transposed LDS reads (ds_read_b64_tr_b16) from the
// row-major [k][n] weight tile, 42 MFMAs per barrier.  The weight's three parts are split once per step
// (conv3_pack_w); the embedding rows are split on their way into LDS.
#include "bf16_mfma.h"
#include "lstur_internal.h"

namespace lstur {

constexpr int C3_BM = 128, C3_BK = 16, C3_SA = 24, C3_SB = 480, C3_NTW = 7, C3_THREADS = 512;
constexpr int C3_A_IMG = C3_BM * C3_SA, C3_B_IMG = C3_BK * C3_SB;          // elements per split part
constexpr int C3_STAGE = 3 * C3_A_IMG + 3 * C3_B_IMG;                      // elements per pipeline stage
constexpr size_t C3_LDS = 2 * C3_STAGE * sizeof(unsigned short) + 2 * C3_BM * sizeof(int32_t);

bool conv3_supported(int D, int ksize, int F) { return D % 4 == 0 && F >= 1 && F <= 448; }
static int conv3_kp(int K) { return cdiv(K, C3_BK) * C3_BK; }
static int conv3_np(int F) { return cdiv(F, 32) * 32; }
size_t conv3_w3_elems(int D, int ksize, int F) { return (size_t)3 * conv3_kp(ksize * D) * conv3_np(F); }

// W [K, F] fp32 -> three bf16 parts [3][Kp][Np], zero padded
__global__ void conv3_pack_w_kernel(const float* __restrict__ W, int K, int F, int Kp, int Np, unsigned short* __restrict__ W3) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)Kp * Np) return;
  const int n = (int)(i % Np), k = (int)(i / Np);
  const float v = (k < K && n < F) ? W[(size_t)k * F + n] : 0.f;
  u16x4 h, m, l;
  split3(make_float4(v, 0.f, 0.f, 0.f), h, m, l);
  W3[i] = h[0];
  W3[(size_t)Kp * Np + i] = m[0];
  W3[(size_t)2 * Kp * Np + i] = l[0];
}
void launch_conv3_pack_w(const float* W, int K, int F, unsigned short* W3, hipStream_t s) {
  const int Kp = conv3_kp(K), Np = conv3_np(F);
  const size_t total = (size_t)Kp * Np;
  hipLaunchKernelGGL(conv3_pack_w_kernel, dim3((unsigned)cdivz(total, 256)), dim3(256), 0, s, W, K, F, Kp, Np, W3);
}

__global__ __launch_bounds__(C3_THREADS) void conv3_fwd_kernel(const float* __restrict__ E, uint32_t e_bytes,
                                                               const unsigned short* __restrict__ W3,
                                                               const float* __restrict__ bias,
                                                               const int32_t* __restrict__ ids, float* __restrict__ C,
                                                               const int32_t* __restrict__ vidx,
                                                               const int32_t* __restrict__ vcount, int M, int T, int Tp, int D,
                                                               int F, int Kp, int Np) {
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  int32_t* rowm = (int32_t*)(smem + 2 * C3_STAGE);        // output row of each of the 128 tile rows (-1: none)
  float* rowkeep = (float*)(rowm + C3_BM);                // 1 / 0: id != 0
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int count = vcount ? vcount[0] : M;
  const int m0 = blockIdx.x * C3_BM;
  if (m0 >= count) return;
  const int NT = Np / 32;                                  // column tiles of 32
  const int wm = wv >> 1, wn = wv & 1;
  const int nt_lo = wn * C3_NTW, my_nt = max(0, min(C3_NTW, NT - nt_lo));

  // A loader role: thread -> (row, 4 k) of the 128 x 16 tile; the row's window offset in E is fixed for the kernel
  const int arow = tid >> 2, aseg = tid & 3;
  uint32_t a_off;
  {
    const int i = min(m0 + arow, count - 1);
    const int m = vidx ? vidx[i] : i;
    const int title = m / T, t = m - title * T;
    a_off = ((uint32_t)(title * Tp + t) * (uint32_t)D + (uint32_t)aseg * 4u) * 4u;
    if (aseg == 0) {
      rowm[arow] = (m0 + arow < count) ? m : -1;
      rowkeep[arow] = ids[m] != 0 ? 1.f : 0.f;
    }
  }
  // B loader role: up to 6 pieces of 16 bytes (8 columns) of the 3 x [16][Np] weight tile
  const int C8 = Np / 8, b_total = 3 * C3_BK * C8;
  uint32_t b_goff[6];
  int b_loff[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int idx = tid + i * C3_THREADS;
    const int img = idx / (C3_BK * C8), rem = idx - img * (C3_BK * C8), krow = rem / C8, ch = rem - krow * C8;
    b_goff[i] = idx < b_total ? (((uint32_t)img * (uint32_t)Kp + (uint32_t)krow) * (uint32_t)Np + (uint32_t)ch * 8u) * 2u : kBufOob;
    b_loff[i] = idx < b_total ? 3 * C3_A_IMG + img * C3_B_IMG + krow * C3_SB + ch * 8 : -1;
  }
  const __amdgpu_buffer_rsrc_t rsE = make_rsrc(E, e_bytes);
  const __amdgpu_buffer_rsrc_t rsW = make_rsrc(W3, (uint32_t)((size_t)3 * Kp * Np * 2));

  f32x16 acc[C3_NTW];
#pragma unroll
  for (int j = 0; j < C3_NTW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  float4 ra;
  float4 rb[6];
  auto load = [&](int k0) {
    ra = buf_load4(rsE, a_off, (uint32_t)k0 * 4u);
#pragma unroll
    for (int i = 0; i < 6; ++i) rb[i] = buf_load4(rsW, b_goff[i], (uint32_t)k0 * (uint32_t)Np * 2u);
  };
  auto store = [&](int buf) {
    unsigned short* st = smem + buf * C3_STAGE;
    u16x4 h, m, l;
    split3(ra, h, m, l);
    *(u16x4*)(st + arow * C3_SA + aseg * 4) = h;
    *(u16x4*)(st + C3_A_IMG + arow * C3_SA + aseg * 4) = m;
    *(u16x4*)(st + 2 * C3_A_IMG + arow * C3_SA + aseg * 4) = l;
#pragma unroll
    for (int i = 0; i < 6; ++i)
      if (b_loff[i] >= 0) *(float4*)(st + b_loff[i]) = rb[i];
  };

  const int nit = Kp / C3_BK;
  load(0);
  for (int it = 0; it < nit; ++it) {
    const int buf = it & 1;
    store(buf);
    __syncthreads();
    if (it + 1 < nit) load((it + 1) * C3_BK);
    const unsigned short* st = smem + buf * C3_STAGE;
    const unsigned short* ap = st + (32 * wm + (lane & 31)) * C3_SA + 8 * (lane >> 5);
    const u16x8 ah = *(const u16x8*)ap, am = *(const u16x8*)(ap + C3_A_IMG), al = *(const u16x8*)(ap + 2 * C3_A_IMG);
    const unsigned short* bp = st + 3 * C3_A_IMG;
#pragma unroll
    for (int j = 0; j < C3_NTW; ++j) {
      if (j < my_nt) {                                   // wave-uniform: EXEC stays full for the transposed reads
        const int x0 = (nt_lo + j) * 32;
        const u16x8 bh = tr_frag(bp, C3_SB, 0, x0, lane);
        const u16x8 bm = tr_frag(bp + C3_B_IMG, C3_SB, 0, x0, lane);
        const u16x8 bl = tr_frag(bp + 2 * C3_B_IMG, C3_SB, 0, x0, lane);
        f32x16 c = acc[j];
        c = mfma32b(al, bh, c);                          // small terms first
        c = mfma32b(ah, bl, c);
        c = mfma32b(am, bm, c);
        c = mfma32b(am, bh, c);
        c = mfma32b(ah, bm, c);
        c = mfma32b(ah, bh, c);
        acc[j] = c;
      }
    }
  }

  // epilogue: + bias, ReLU, x (id != 0); rows scattered to the token's place, branch-free through a buffer descriptor
  const int hh = lane >> 5, c32 = lane & 31;
  const __amdgpu_buffer_rsrc_t rsC = make_rsrc(C, (uint32_t)std::min((size_t)M * F * 4, (size_t)0xFFFFFFFFu));
  const __amdgpu_buffer_rsrc_t rsBias = make_rsrc(bias, (uint32_t)F * 4u);
  int orow[16];
  float okeep[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int lr = 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * hh;
    orow[r] = rowm[lr];
    okeep[r] = rowkeep[lr];
  }
#pragma unroll
  for (int j = 0; j < C3_NTW; ++j) {
    const int col = (nt_lo + j) * 32 + c32;
    const bool cok = j < my_nt && col < F;
    const float bv = buf_load1(rsBias, cok ? (uint32_t)col * 4u : kBufOob, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r)
      buf_store1(fmaxf(acc[j][r] + bv, 0.f) * okeep[r], rsC,
                 (cok && orow[r] >= 0) ? ((uint32_t)orow[r] * (uint32_t)F + (uint32_t)col) * 4u : kBufOob, 0);
  }
}

void launch_conv3_fwd(const float* emb, const unsigned short* W3, const float* bias, const int32_t* ids, float* out,
                      int n_titles, int T, int D, int ksize, int F, const int32_t* vidx, const int32_t* vcount,
                      hipStream_t s) {
  const int M = n_titles * T;
  if (M == 0) return;
  const int Tp = T + ksize - 1, K = ksize * D;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv3_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C3_LDS);
    attr_set = true;
  }
  // the padded embedding layout + its zero slack (kEmbSlack floats): the k-tail of the last window reads into it
  const uint32_t e_bytes = (uint32_t)std::min(((size_t)n_titles * Tp * D + kEmbSlack) * 4, (size_t)0xFFFFFFFFu);
  hipLaunchKernelGGL(conv3_fwd_kernel, dim3(cdiv(M, C3_BM)), dim3(C3_THREADS), C3_LDS, s, emb, e_bytes, W3, bias, ids, out,
                     vidx, vcount, M, T, Tp, D, F, conv3_kp(K), conv3_np(F));
}

}  // namespace lstur
