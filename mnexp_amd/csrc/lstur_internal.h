// Host-side launcher prototypes (one per kernel family); implemented in the *.hip files,
// orchestrated by lstur_capi.hip.  All launch on the given stream and never synchronise.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stddef.h>
#include <stdint.h>

#include <algorithm>

namespace lstur {

// ---- deferred partial-sum reductions (misc.hip) -----------------------------------------------------
// Split-K GEMMs, two-stage column sums and the conv kernel gradient leave per-slice partial sums; instead of one
// tiny reduction launch each (a dependent launch costs ~5 us on its own) they describe the reduction as a job and
// ONE kernel at the end of the backward pass runs them all, in fixed slice order (bitwise reproducible):
//   dst[r * ld + c] = alpha * sum_{s < nslices} src[s * stride + r * cols + c]  (+ beta * dst when beta != 0)
struct ReduceJob {
  const float* src;
  float* dst;
  int rows, cols, ld, nslices;
  size_t stride;
  float alpha;
  float beta;
};
constexpr int kMaxReduceJobs = 41;   // launch_multi_reduce refuses more (returns false)

// Fork points without a bubble: hipEventRecord between two kernels of a stream is a barrier packet of its own and costs
// ~6 us of idle GPU on this stack.  A launcher that goes through launch_k attaches the event to the kernel's OWN completion
// signal instead (hipExtLaunchKernelGGL's stopEvent) when the orchestration armed one with arm_stop_event(); it consumes
// the armed event, so the caller can tell (stop_event_pending()) whether it still has to record the classic way.
extern thread_local hipEvent_t g_stop_event;
inline void arm_stop_event(hipEvent_t e) { g_stop_event = e; }
inline bool stop_event_pending() { return g_stop_event != nullptr; }
// The same for timing brackets (bench.py's roofline kernels inside its timed region): arm_bracket(start, stop) attaches
// `start` to the next launch_k launch and `stop` to the launch in front of which the launcher calls last_launch() -- a
// family of kernels is bracketed from its first to its last launch without a single record packet (two records per
// family cost the step 17-19 us over gather + conv forward + conv kernel gradient).
extern thread_local hipEvent_t g_start_event, g_stop_pending;
inline void arm_bracket(hipEvent_t start, hipEvent_t stop) {
  g_start_event = start;
  g_stop_pending = stop;
}
inline void last_launch() {
  if (g_stop_pending) {
    g_stop_event = g_stop_pending;
    g_stop_pending = nullptr;
  }
}
template <typename F, typename... Args>
inline void launch_k(F kern, dim3 grid, dim3 block, size_t lds, hipStream_t s, Args... args) {
  if (g_stop_event || g_start_event) {
    hipEvent_t e0 = g_start_event, e1 = g_stop_event;
    g_start_event = g_stop_event = nullptr;
    hipExtLaunchKernelGGL(kern, grid, block, (uint32_t)lds, s, e0, e1, 0, args...);
  } else {
    hipLaunchKernelGGL(kern, grid, block, lds, s, args...);
  }
}
bool launch_multi_reduce(const ReduceJob* jobs, int njobs, hipStream_t s);



__host__ __device__ static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ static inline size_t cdivz(size_t a, size_t b) { return (a + b - 1) / b; }
__host__ __device__ static inline int ceil16(int a) { return (a + 15) & ~15; }

// floats of slack the padded embedding buffer carries past its last row so that MFMA operand
// loads running past k*D (K padded to 16 / 64) stay inside the allocation.
constexpr int kEmbSlack = 256;

// ---- misc.hip ----------------------------------------------------------------------------------
void launch_gather(const float* table, const int32_t* ids, float* out, int n_titles, int T, int D,
                   int pad_l, int pad_r, float drop, uint32_t key, hipStream_t s);
// the same rows rounded to bf16 (the mhsa projections' A operand)
void launch_gather_bf16(const float* table, const int32_t* ids, unsigned short* out, int n_titles, int T, int D, int pad_l,
                        int pad_r, float drop, uint32_t key, hipStream_t s);
void launch_expand_titles(const int32_t* table, const int32_t* doc_ids, int32_t* out, int n, int T,
                          hipStream_t s);
// stable list of non-pad tokens: vidx[0..*count) ; scratch = compact_scratch_ints(n) int32
size_t compact_scratch_ints(int n);
void launch_compact_valid(const int32_t* ids, int n, int32_t* vidx, int32_t* count, int32_t* scratch, hipStream_t s);
// the same compaction in two fused launches: block counts come from launch_expand_count (ids form, together with the
// title expansion) or launch_compact_count; launch_compact_write_offsets then writes vidx, the count and (offA != NULL)
// the conv kernel gradient's per-token row offsets
void launch_expand_count(const int32_t* table, const int32_t* hist_ids, int n_hist, const int32_t* cand_ids, int n_cand,
                         int T, int32_t* out, int32_t* block_counts, hipStream_t s);
void launch_compact_count(const int32_t* ids, int n, int32_t* block_counts, hipStream_t s);
void launch_compact_write_offsets(const int32_t* ids, int n, const int32_t* block_counts, int32_t* vidx, int32_t* count,
                                  int T, int Tp, int D, int F, int32_t* offA, int32_t* offB, hipStream_t s,
                                  uint32_t* live = nullptr,    // live != NULL: + the titles' non-pad bit masks
                                  int split_at = -1);          // count[1] = non-pad tokens in front of token split_at
size_t pack_b_floats(int K, int N);
void launch_pack_b(const float* B, int ldb, int K, int N, float* packed, hipStream_t s);
void launch_pack_b_t(const float* Bt, int ldb, int K, int N, float* packed, hipStream_t s);
// out[r][c] = in[c][r]  (in is [rows_in, cols_in] with leading dim ld_in)
void launch_transpose(const float* in, int ld_in, int rows_in, int cols_in, float* out, hipStream_t s);
void launch_colsum(const float* x, int rows, int cols, int ld, float* out, float scale, hipStream_t s);
constexpr int kColsumSlices = 64;
void launch_colsum2(const float* x, int rows, int cols, int ld, float* out, float scale, float* scratch,
                    hipStream_t s, ReduceJob* defer = nullptr);

void launch_adam_flat(float* p, const float* g, float* m, float* v, size_t n, float lr_t, float b1,
                      float b2, float eps, hipStream_t s);
// Adam over the dense bucket + the weight images of the NEXT forward in the same launch: an updated conv_w element also
// lands (split into three bf16 parts) in the conv kernels' W3 image, an updated gru_u element in the two resident-U images
// that hold it.  The images' padding stays as the stand-alone pack kernels wrote it once.
struct ImagePlan {
  size_t conv_off = 0, conv_cnt = 0;   // conv_w slot of the bucket ([ksize * D, F]); 0: no split weight image
  int D = 0, Dp = 0, F = 0, Np = 0;
  unsigned short* w3 = nullptr;        // the conv kernels' split weight image Wx (convx.hip)
  size_t w3_part = 0;                  // elements of one part of it (Kp * Np)
  int np = 3;                          // parts: 3 = bf16x3, 2 = f16x2 (then cs / seq: the scale block and the pack's
  float* cs = nullptr;                 // sequence number, convx_pack_w_kernel's rule; grow: bound on one step's |dp|)
  int seq = 0;
  float grow = 0.f;
  size_t gru_off = 0;                  // gru_u slot ([H, 3H]); H == 0: no resident-U images
  int H = 0, KQ = 0;
  float* gimg = nullptr;
  int gru_f16 = 0, gru_seq = 0;        // fp16-pair images: scale from the tail's rotating maxima (gru4.hip), + ggrow
  float ggrow = 0.f;
};
void launch_adam_flat_pack(float* p, const float* g, float* m, float* v, size_t n, float lr_t, float b1, float b2, float eps,
                           const ImagePlan& plan, hipStream_t s);
// the same sweep + launch_user_grad_combine(rows, rg, ...) in one launch
void launch_adam_flat_combine(float* p, const float* g, float* m, float* v, size_t n, float lr_t, float b1, float b2,
                              float eps, const int32_t* rows, const float* rg, int n_rows, int dim, float* gsum,
                              int32_t* slot, hipStream_t s, const ImagePlan* plan = nullptr, const float* rows_f = nullptr,
                              int32_t* rows_out = nullptr);   // rows_f: the row numbers as floats (-> rows_out as int32)
// dense Adam sweep of the user table; slot[row] = index into gsum or -1
void launch_adam_table(float* p, float* m, float* v, const int32_t* slot, const float* gsum, int n_rows,
                       int dim, float lr_t, float b1, float b2, float eps, hipStream_t s);
// dedupe rows -> slot map + summed grads (ascending sample order); first[i] scratch int32 [n]
void launch_user_grad_combine(const int32_t* rows, const float* g, int n, int dim, int32_t* first,
                              float* gsum, int32_t* slot, hipStream_t s);
// data-parallel exchange of the sparse user rows through the dense bucket's all-reduce (see misc.hip)
void launch_dp_stage(const int32_t* rows, const float* g, const float* g2, int n, int B, int Ue, float* tail, int rank,
                     int world, hipStream_t s);
void launch_dp_unpack(const float* tail, int n, int B, int Ue, int world, int32_t* rows, float* g, float* g2, hipStream_t s);
int dp_stage_jobs(const float* rows_f, const float* g, const float* g2, int n, int B, int Ue, float* tail, int rank, int world,
                  ReduceJob* jobs);   // the staging as <= 9 jobs of the backward pass's reduction launch
void launch_user_slot_clear(const int32_t* rows, int n, int32_t* slot, hipStream_t s);
// lazy dense Adam of large user tables (misc.hip): a row takes its pending zero-gradient updates when it is next read or
// updated; last[row] = the iteration it is current for, lrt[t] = the lr_t of iteration t
void launch_lazy_catchup(float* p, float* m, float* v, float* p2, float* m2, float* v2, const int32_t* rows, int n,
                         int32_t* first, int32_t* last, const float* lrt, int upto, int dim, float b1, float b2, float eps,
                         int mask, hipStream_t s);    // (p2 != NULL: both tables; first: n int32 of scratch)
void launch_lazy_rows_adam(float* p, float* m, float* v, const int32_t* rows, const int32_t* slot, const float* gsum, int n,
                           int dim, const int32_t* last, const float* lrt, int t, float lr_t, float b1, float b2, float eps,
                           int mask, hipStream_t s);
void launch_lazy_rows_mark(const int32_t* rows, const int32_t* slot, int n, int32_t* last, float* lrt, int t, float lr_t,
                           int mask, hipStream_t s);   // behind the row steps of every table
void launch_lazy_flush(float* p, float* m, float* v, const int32_t* last, const float* lrt, int upto, int n_rows, int dim,
                       float b1, float b2, float eps, int mask, hipStream_t s);   // (lrt is indexed t & mask)
// the sweep in two halves (fused step): mark the batch's rows, move every OTHER row with its zero gradient (independent of
// the step's backward), later finish the marked rows from their combined gradients (first occurrences, slot[row] == i)
void launch_user_rows_mark(const int32_t* rows, int n, int32_t* slot, hipStream_t s);
void launch_adam_table_unmarked(float* p, float* m, float* v, const int32_t* slot, int n_rows, int dim, float lr_t,
                                float b1, float b2, float eps, int max_blocks, hipStream_t s);
void launch_adam_rows(float* p, float* m, float* v, const int32_t* rows, const int32_t* slot, const float* gsum, int n,
                      int dim, float lr_t, float b1, float b2, float eps, hipStream_t s);
void launch_defer_rows(float* p, float* m, float* v, float* p2, float* m2, float* v2, const int32_t* rows, int n, int32_t* slot,
                       int dim, float lr_t, float b1, float b2, float eps, hipStream_t s);
void launch_fill_i32(int32_t* p, int32_t v, size_t n, hipStream_t s);
// history mask (ComputeMasking) applied to doc vectors + GRU step mask:  hv[b,w,:] *= any(ids != 0);
// gmask[b,w] = any(hv != 0)
// + (table != NULL) the user-embedding gather of the batch in the same launch: user_rows[b] = user[b], uvec[b] = table[user[b]]
void launch_hist_mask(const int32_t* hist_ids, float* hv, uint8_t* hmask, uint8_t* gmask, int nW, int T, int U,
                      hipStream_t s, const float* table = nullptr, const int32_t* user = nullptr,
                      int32_t* user_rows = nullptr, float* uvec = nullptr, int n = 0, int Ue = 0,
                      float* user_rows_f = nullptr);
void launch_scale_rows_u8(float* x, const uint8_t* mask, int rows, int cols, hipStream_t s);
void launch_gather_rows(const float* table, const int32_t* rows, float* out, int n, int dim, hipStream_t s);
// concat: out[b, :a_cols] = a[b], out[b, a_cols:] = bm[b]
void launch_concat2(const float* a, int a_cols, const float* bm, int b_cols, float* out, int n, hipStream_t s);
void launch_split2(const float* in, int a_cols, int b_cols, float* a, float* bm, int n, hipStream_t s);
// listwise score models ddot / dnn (paper.py:448-455) and the dgru whole-row dropout (paper.py:610)
void launch_tanh_inplace(float* x, size_t n, hipStream_t s);
void launch_tanh_bwd(float* g, const float* y, size_t n, hipStream_t s);
void launch_concat_rep(const float* u, int uc, int K1, const float* d, int dc, float* out, int rows, hipStream_t s);
void launch_mlp_dhid(const float* pre, const float* ds, const float* w2, float* dhid, int rows, int U, hipStream_t s);
void launch_split_sum(const float* dcat, int uc, int dc, int K1, float* du, float* dd, int n, hipStream_t s);
void launch_fill_f32(float* p, float v, size_t n, hipStream_t s);
// niavg: masked mean over the history (forward: du == NULL; backward: writes dx)
void launch_masked_mean(const float* x, const uint8_t* mask, float* u, const float* du, float* dx, int n, int W, int U,
                        hipStream_t s);
void launch_add2(const float* a, const float* b, float* out, size_t n, hipStream_t s);
void launch_drop_rows(float* x, int n, int dim, float p, uint32_t key, hipStream_t s);
void launch_score(const float* u, const float* d, float* logits, float* probs, float* loss_i, float* correct_i,
                  float* du, float* dd, int n, int K1, int U, float inv_n, hipStream_t s);
void launch_predict_score(const float* u, const float* d, float* scores, int n, int U, hipStream_t s);
// stats[0] = mean(loss_i), stats[1] = mean(correct_i)
void launch_mean2(const float* a, const float* b, int n, float* stats, hipStream_t s);

// ---- conv.hip ----------------------------------------------------------------------------------
// vidx/vcount (nullable): compacted list of output rows to compute (others are left untouched)
void launch_conv_fwd(const float* emb, const float* w_packed, const float* bias, const int32_t* ids,
                     float* out, int n_titles, int T, int D, int ksize, int F, const int32_t* vidx,
                     const int32_t* vcount, hipStream_t s);
size_t conv_wgrad_scratch_bytes(int D, int ksize, int F);
// offA/offB/zeros (nullable): per-token float offsets from launch_token_offsets + >= 16 B of zeros; with them
// the F == 400 configuration stages its operands by LDS-DMA
void launch_conv_wgrad(const float* emb, const float* dout, float* dw, void* scratch, int n_titles, int T,
                       int D, int ksize, int F, const int32_t* vidx, const int32_t* vcount, hipStream_t s,
                       const int32_t* offA = nullptr, const int32_t* offB = nullptr, const float* zeros = nullptr,
                       ReduceJob* defer = nullptr);
// conv input gradient scattered into the word-table gradient (trainable word table); scratch: conv_dgrad_scratch_floats
size_t conv_dgrad_scratch_floats(int n_titles, int T, int D, int ksize, int F);
void launch_conv_dgrad_scatter(const float* dconv, const float* conv_w, const int32_t* ids, float* word_grad,
                               float* scratch, int n_titles, int T, int D, int ksize, int F, float drop, uint32_t key,
                               hipStream_t s);
void launch_token_offsets(const int32_t* vidx, const int32_t* vcount, int M, int T, int Tp, int D, int F, int32_t* offA,
                          int32_t* offB, hipStream_t s);

// ---- convx.hip: Conv1D as an fp32-grade contraction on the 16-bit matrix cores, both operands PRE-SPLIT into np parts:
// np = 2 "f16x2" (two fp16 parts of the power-of-two scaled value, three products), np = 3 "bf16x3" (three bf16 parts, six
// products).  The file holds the gather that writes the split embeddings "Ex", the weight pack, forward, kernel gradient.
// Scale block `cs` (np == 2; kCsWords 32-bit words of device memory, launch_convx_scales_init):
constexpr int kCsE = 0;       // float: power of two the embeddings are staged with
constexpr int kCsW = 1;       // float: ... the current weight image was packed with
constexpr int kCsFwd = 2;     // float: 1 / (sE sW), the forward's un-scale
constexpr int kCsInvE = 3;    // float: 1 / sE
constexpr int kCsEmax = 7;    // uint32: bits of max |word table|
constexpr int kCsGmax = 8;    // float: max |dC| of a stand-alone kernel-gradient call (in the step: the attention backward's
                              // per-workgroup maxima, lstur_internal.h: launch_attpool_bwd gparts)
// max |conv_w| as seen by the last packs: 3 rotating slots (convx.hip) of kWmaxLanes words, kWmaxStride words apart -- a
// workgroup adds its maximum to word (workgroup % kWmaxLanes) of a slot, a reader takes the maximum over the slot's words:
// thousands of atomicMax on ONE address cost the dense Adam launch 60 us (they serialise in one L2 channel)
constexpr int kWmaxLanes = 64, kWmaxStride = 64;
constexpr int kCsWmax = 64;
constexpr int kCsWords = kCsWmax + 3 * kWmaxLanes * kWmaxStride;
__host__ __device__ inline int wmax_word(int slot, int lane) { return kCsWmax + (slot * kWmaxLanes + lane) * kWmaxStride; }
// the same for the recurrent matrix, in the tail behind the GRU's operand images (gru4.hip): [0] sU, [1] 1 / sU, the slots
constexpr int kGruTailWords = kWmaxStride + 3 * kWmaxLanes * kWmaxStride;
__host__ __device__ inline int gru_wmax_word(int slot, int lane) { return kWmaxStride + (slot * kWmaxLanes + lane) * kWmaxStride; }
bool convx_supported(int D, int ksize, int F);                    // D % 4 == 0, F <= 448, ksize <= 8
int convx_dp(int D);                                              // embedding row length of Ex (D rounded up to 8)
size_t convx_e_elems(int n_titles, int T, int D, int ksize, int np);   // 16-bit elements of Ex incl. slack
size_t convx_w_elems(int D, int ksize, int F, int np);                 // 16-bit elements of the split weight image
void convx_w_dims(int D, int ksize, int F, int np, int* Kp, int* Np);   // padded shape of the image ([np][Kp][Np], k' = tap * Dp + d)
void launch_convx_scales_init(float* cs, hipStream_t s);
void launch_absmax(const float* x, size_t n, uint32_t* out_bits, hipStream_t s);   // atomicMax of the bits of |x[i]|
void launch_absmax_rows(const float* x, int rows, int cols, const int32_t* vidx, const int32_t* vcount, uint32_t* out_bits,
                        hipStream_t s);                                             // ... over the listed rows only
void launch_convx_escale(const float* table, size_t count, float drop, float* cs, hipStream_t s);
// skip_unread: rows no window of a non-pad token covers are neither read nor written (the buffer must hold finite values)
void launch_gatherx(const float* table, const int32_t* ids, unsigned short* ex, int n_titles, int T, int D, int pad_l,
                    int pad_r, float drop, uint32_t key, bool skip_unread, int np, const float* cs, hipStream_t s,
                    int max_blocks = 0, const uint32_t* live = nullptr);
// fresh: the scale comes from W as it is; else from the previous pack's maximum + grow (how far a step can move a weight)
void launch_convx_pack_w(const float* W, int D, int ksize, int F, unsigned short* Wx, int np, float* cs, int seq, bool fresh,
                         float grow, hipStream_t s);
// scratch (convx_fwd_scratch_bytes, or NULL): slabs of the k-sliced tail tiles
size_t convx_fwd_scratch_bytes(int ncu = 256);
void launch_convx_fwd(const unsigned short* ex, const unsigned short* Wx, const float* bias, const int32_t* ids, float* out,
                      int n_titles, int T, int D, int ksize, int F, const int32_t* vidx, const int32_t* vcount, int np,
                      const float* cs, void* scratch, hipStream_t s, int variant = 0);
// kernel gradient on the pre-split embeddings: dw through ksize reduce jobs (one per tap); returns the job count.
// offA: per-token window offsets in 16-byte pieces of Ex, offB: per-token row offsets in floats of dout.
size_t convx_wgrad_scratch_bytes(int n_titles, int T, int D, int ksize, int F);
int convx_wgrad_slabs(int plan_titles, int T, int D, int ksize, int F);   // slabs a launch planned for these titles writes
size_t convx_wgrad_slab_floats(int D, int ksize, int F);                  // floats of one slab ([k * Dp, F])
int launch_convx_wgrad(const unsigned short* ex, const float* dout, float* dw, void* scratch, int n_titles, int T, int D,
                       int ksize, int F, const int32_t* vcount, hipStream_t s, ReduceJob* jobs, const int32_t* offA,
                       const int32_t* offB, int np, const float* cs, const float* gparts, int g_n, float alpha = 1.0f,
                       int range_sel = 0, int slab_base = 0, int plan_titles = 0, int* slabs_out = nullptr, int variant = 0);

// ---- gemm3.hip: C = A[M,K] . op(B) + bias, fp32-grade on the bf16 matrix cores (bf16x3), for the dense layers on the
// step's critical path.  b_nk: B is [N][K] (C = A . B^T), else [K][N].  Check gemm3_supported first.
bool gemm3_supported(int lda, int ldb, int M, int N, int K);
void launch_gemm3(const float* A, int lda, const float* B, int ldb, bool b_nk, float* C, int ldc, const float* bias, int M,
                  int N, int K, hipStream_t s);

// C[M,N] = alpha * A^T . B, A [K][M], B [K][N] (parameter gradients), same arithmetic; K is split into slabs in `scratch`
// (gemm3_tn_scratch_bytes) that *job sums in slab order
bool gemm3_tn_supported(int lda, int ldb, int M, int N, int K);
size_t gemm3_tn_scratch_bytes(int M, int N, int K);
void launch_gemm3_tn(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K, float alpha,
                     void* scratch, hipStream_t s, ReduceJob* job);

// ---- attpool.hip -------------------------------------------------------------------------------
// ids (nullable): rows with id == 0 are taken as zero without reading c / writing dc.  live (nullable): the titles' non-pad
// bit masks of launch_compact_write_offsets for the same ids -- one scalar per title instead of an ids read per row
void launch_attpool_fwd(const float* c, const int32_t* ids, const float* att_w, const float* att_b, float* pooled,
                        float* a, float* w, int n_titles, int T, int F, float drop, uint32_t key, hipStream_t s,
                        const uint32_t* live = nullptr, const uint32_t* kbits = nullptr);
// kbits: the dropout keep decisions of (key, drop) for these titles in the title kernels' thread layout, made ahead of time
// (attpool_keep_bits_words(n_titles) words; false: not applicable to this shape / drop == 0 -- pass NULL to the passes)
size_t attpool_keep_bits_words(int n_titles);
bool launch_attpool_keep_bits(const uint32_t* live, uint32_t* kbits, int n_titles, int T, int F, float drop, uint32_t key,
                              hipStream_t s);
int attpool_bwd_parts(int n_titles);
// row_scratch: n_titles*T floats (dw -> ds)
void launch_attpool_bwd(const float* c, const int32_t* ids, const float* att_w, const float* a, const float* w,
                        const float* dpooled, float* dc, float* parts, float* row_scratch, int n_titles, int T, int F,
                        float drop, uint32_t key, hipStream_t s, int ldp = 0, int off_aw = 0, int off_cb = 0,
                        int off_ab = 0, int relu = 1, const uint32_t* live = nullptr, const uint32_t* kbits = nullptr,
                        int title0 = 0, int title_count = -1,
                        float* gparts = nullptr,      // != NULL: gparts[workgroup] = max |dc| over the rows it wrote (title
                                                      // path: one entry per title, indexed by title; else attpool_bwd_parts)
                        int tpb = 1, int row0 = 0);   // title path, tpb > 1: a workgroup takes tpb titles and leaves ONE
                                                      // partial-sum row / maximum: rows row0 .. row0 + cdiv(count, tpb) - 1
bool attpool_title_path(int T, int F);   // the title-resident kernels (the only ones that take a title sub-range)

// ---- mhsa.hip (build-defined multi-head self-attention token stage; T <= 32, F/heads in {16, 32, 64}) -------
bool mhsa_supported(int T, int F, int heads);
// qkv [n_titles*T, 3F] = Q | K | V (head h = columns h*dh .. of each); out [n_titles*T, F] (rows of pad tokens untouched)
void launch_mhsa_fwd(const float* qkv, const int32_t* ids, float* out, int n_titles, int T, int F, int heads,
                     hipStream_t s);
// dout [n_titles*T, F] (rows of pad tokens are not read) -> dqkv [n_titles*T, 3F] (rows of pad tokens = 0)
void launch_mhsa_bwd(const float* qkv, const float* dout, const int32_t* ids, float* dqkv, int n_titles, int T, int F,
                     int heads, hipStream_t s);

// the same two kernels on bf16 Q|K|V / dQ|dK|dV rows (the form the step uses: gemm_bf16.hip writes / reads them)
void launch_mhsa_fwd_bf16(const unsigned short* qkv, const int32_t* ids, float* out, int n_titles, int T, int F,
                          int heads, hipStream_t s);
void launch_mhsa_bwd_bf16(const unsigned short* qkv, const float* dout, const int32_t* ids, unsigned short* dqkv,
                          int n_titles, int T, int F, int heads, hipStream_t s);

// ---- gemm_bf16.hip (projections of the mhsa encoder: bf16 MFMA operands, fp32 accumulation) ------------------
bool proj_bf16_supported(int D, int N);
// out bf16 [M, N] = E[M, K] . W[K, N] + bias
void launch_proj_fwd_bf16(const float* E, const float* W, const float* bias, unsigned short* out, int M, int N, int K,
                          hipStream_t s);
size_t proj_wgrad_scratch_bytes(int D, int N, int K);
// dW [D + 1, N] = [E | 1]^T . dqkv (row D = bias gradient; dW and db contiguous in the bucket), K = tokens
void launch_proj_wgrad_bf16(const float* E, const unsigned short* dqkv, float* dW, void* scratch, int D, int N, int K,
                            hipStream_t s, ReduceJob* defer);
// both operands bf16 in HBM already (launch_gather_bf16 rows, launch_round_bf16 weight): what the step runs
void launch_proj_fwd_bf16e(const unsigned short* E16, const unsigned short* W16, const float* bias, unsigned short* out, int M,
                           int N, int K, hipStream_t s);
void launch_proj_wgrad_bf16e(const unsigned short* E16, const unsigned short* dqkv, float* dW, void* scratch, int D, int N, int K,
                             hipStream_t s, ReduceJob* defer);
void launch_round_bf16(const float* src, unsigned short* dst, size_t n, hipStream_t s);

// ---- pointwise head (misc.hip) --------------------------------------------------------------------
// point_dot: s = u . d; point_mlp: s = relu(pre) . w2 + b2 (pre = [u|d] W1 + b1 computed by the caller).
// p = sigmoid(s); loss_i = Seq2Vec.loss term (task/seq2vec.py:213-216); ok_i = ((p > .5) == label);
// ds = dloss/ds * scale (scale = grad_scale / n).  Backward outputs may be NULL (forward only).
void launch_point_dot(const float* u, const float* d, const float* labels, float* logits, float* probs, float* loss_i,
                      float* ok_i, float* du, float* dd, int n, int U, float gain, int K, float scale, hipStream_t s);
void launch_point_mlp_out(const float* pre, const float* w2, const float* b2, const float* labels, float* logits,
                          float* probs, float* loss_i, float* ok_i, float* ds, float* dhid, int n, int U, float gain,
                          int K, float scale, hipStream_t s);
// dw2[j] = sum_i relu(pre[i][j]) ds[i]; db2 = sum_i ds[i]
void launch_point_w2_grad(const float* pre, const float* ds, float* dw2, float* db2, int n, int U, hipStream_t s);

// ---- gemm.hip ----------------------------------------------------------------------------------
size_t gemm_scratch_bytes(int M, int N, int K);
// defer != NULL (and bias == NULL): a split-K product leaves its slabs in `scratch` and fills *defer
// (nslices == 0 when no reduction is needed); the scratch must then stay untouched until launch_multi_reduce.
void launch_gemm(const float* A, int lda, int ta, const float* B, int ldb, int tb, float* C, int ldc,
                 const float* bias, int M, int N, int K, float alpha, void* scratch, hipStream_t s,
                 ReduceJob* defer = nullptr);

// ---- gru.hip / gru4.hip --------------------------------------------------------------------------
// Operand images of the recurrent matrix U [H,3H]: ONE opaque buffer of gru_pack_floats(H) floats (one launch).
size_t gru_pack_floats(int H);
// f16 (fast path only): the FORWARD's recurrent products as fp16 pairs on v_mfma_f32_4x4x4_16b_f16 (gru4.hip) -- pack and
// forward of one context must agree on it (the BPTT's operand images stay floats).  seq: the pack's sequence number (rotating maxima of |U|, like the conv weight image).
bool gru4_f16_env();                        // !LSTUR_GRU_F32 (read where a context / a stand-alone call is set up)
void launch_gru_pack(const float* U, int H, float* packed, hipStream_t s, bool f16 = false, int seq = 0);
// tape: gru_tape_floats(n, W, H) floats of scratch shared by forward and BPTT (layout private to the kernels).
// xw [n*W,3H] row-major; hlast [n,H]; dgates [n*W,3H], rh / hprev [n*W,H] row-major.  Requires H <= 256.
size_t gru_tape_floats(int n, int W, int H);
// The listwise scorer (misc.hip: launch_score, K1 <= 8 candidates, U = H <= 256) as the tail of the forward recurrence's
// kernel (fast path only): d = the candidates' doc vectors [n * K1, H], the other fields as launch_score's.  d == NULL: none.
struct GruScore {
  const float* d = nullptr;
  float *logits = nullptr, *probs = nullptr, *loss_i = nullptr, *correct_i = nullptr, *du = nullptr, *dd = nullptr;
  int K1 = 0;
  float inv_n = 0.f;
};
bool gru_score_fusable(int H, int K1);   // the fast recurrence kernel runs for this H and can score K1 candidates
void launch_gru_fwd(const float* xw, const uint8_t* gmask, const float* h0, const float* packed, float* tape,
                    float* hlast, int n, int W, int H, hipStream_t s, bool f16 = false, const GruScore* sc = nullptr);
void launch_gru_bwd(const float* dh_last, const uint8_t* gmask, float* tape, const float* packed, float* dgates,
                    float* rh, float* hprev, float* dh0, int n, int W, int H, hipStream_t s);
// On the fast path the forward already leaves hprev and r*hprev row-major inside the tape: returns true and the two
// pointers (pass them to launch_gru_bwd to skip its copies); false on the general path.
bool gru_tape_rows(float* tape, int n, int W, int H, float** hprev, float** rh);

// gru4.hip (fast path: four sample rows per workgroup, U resident in VGPRs + LDS; H <= 200)
bool gru4_supported(int H);
int gru4_pack_kq(int H);            // k-quads per column of the resident-U images (0: the general kernels' layout is used)
size_t gru4_pack_floats(int H);
size_t gru4_tape_floats(int n, int W, int H);
void gru4_tape_rows(float* tape, int n, int W, int H, float** hprev, float** rh);
size_t gru4_image_floats(int H);     // floats of the six operand images (the pack's tail follows: scale, rotating maxima)
void launch_gru4_pack(const float* U, int H, float* img, hipStream_t s, bool f16 = false, int seq = 0);
void launch_gru4_fwd(const float* xw, const uint8_t* gmask, const float* h0, const float* img, float* tape, float* hlast,
                     int n, int W, int H, hipStream_t s, bool f16 = false, const GruScore* sc = nullptr);
void launch_gru4_bwd(const float* dh_last, const uint8_t* gmask, float* tape, const float* img, float* dgates, float* rh,
                     float* hprev, float* dh0, int n, int W, int H, hipStream_t s);

}  // namespace lstur
