/*
 * lstur.h -- C-ABI of liblstur_hip.so: the MI355X (gfx950) LSTUR training path.
 *
 * The reference (nvagus/mnexp) has no FFI: its seam is the Python task-handler protocol
 * (task/__init__.py:15-16, main.py:66-94) and everything under `build_model()` is a Keras
 * graph (task/paper.py:635-665).  This library replaces that graph.  Each entry point cites
 * the reference construct it stands in for.  Conventions:
 *   - extern "C", plain pointers and sizes, no torch types; every function returns
 *     0 on success or a negative lstur_status; lstur_last_error() gives the message
 *     (thread-local).  No exception crosses the boundary.
 *   - the CALLER owns every device buffer (parameters, optimizer state, workspace) and the
 *     HIP stream; the library only launches kernels on the stream it is given and never
 *     allocates, frees or synchronises.  A context is re-entrant across contexts, not
 *     thread-safe within one.
 *   - all tensors are dense row-major fp32 / int32 device memory unless stated.
 *
 * Title batch layout: N_t = n*(W) history titles (sample-major) followed by n*K1 candidate
 * titles (sample-major); candidate 0 of a training sample is the clicked (positive) one.
 */
#ifndef LSTUR_H
#define LSTUR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lstur_ctx lstur_ctx;
typedef void* lstur_stream;               /* hipStream_t */

enum lstur_status {
  LSTUR_OK = 0,
  LSTUR_ERR_ARG = -1,        /* bad argument / unsupported shape */
  LSTUR_ERR_HIP = -2,        /* a HIP call or kernel launch failed */
  LSTUR_ERR_STATE = -3,      /* call order violated (e.g. buffers not bound) */
  LSTUR_ERR_NOGPU = -4       /* no gfx950 device visible */
};

enum lstur_arch {            /* user encoders of Seq2VecPaperSoftmaxId.get_user_encoder, paper.py:584-633 */
  LSTUR_ARCH_IGRU = 0,       /* LSTUR-ini: GRU(U)(x, initial_state=user_vec)            paper.py:612-613 */
  LSTUR_ARCH_GRU = 1,        /* LSTUR-con: Dense(U)(concat[GRU(U)(x), user_vec])        paper.py:596-599 */
  LSTUR_ARCH_HGRU = 2,       /* LSTUR-con: concat[GRU(U/2)(x), Embedding(N_u, U/2)]     paper.py:603-607 */
  LSTUR_ARCH_NIGRU = 3,      /* no user id: GRU(U)(x) from a zero state (paper.py:627-628 `nigru`; `gru` of the
                                id-less Seq2VecPaperSoftmax, paper.py:206-208); user arguments are ignored     */
  LSTUR_ARCH_NGRU = 4,       /* LSTUR-con, plain concat[GRU(U)(x), user_vec]: a 2U-wide user vector (paper.py:600-602);
                                needs score_model ddot or dnn                                                  */
  LSTUR_ARCH_DGRU = 5,       /* ngru with Dropout(0.5, noise_shape=(None,1)) on the user embedding: a sample's whole
                                embedding is dropped with p = 0.5 in training (paper.py:608-611; the masking of
                                the LSTUR paper); dropout stream 2 of the step                                 */
  LSTUR_ARCH_PGRU = 6,       /* GRU(U)(x) + user_vec                                      paper.py:624-626       */
  LSTUR_ARCH_VO = 7,         /* the user embedding alone; the history is not used         paper.py:622-623       */
  LSTUR_ARCH_NIAVG = 8,      /* no user id: sum_w x_w / (sum_w mask_w + 1e-7), models.GlobalAveragePoolingMaskSupport
                                (models.py:433-435)            paper.py:629-630; id-less `avg`, paper.py:213-215 */
  LSTUR_ARCH_NIATT = 9,      /* no user id: SimpleAttentionMaskSupport over the history vectors (models.py:474-489);
                                the id-less tasks' `att`, paper.py:207-209; parameters uatt_w [U], uatt_b [1]   */
  LSTUR_ARCH_IIGRU = 10,     /* Dense(U)(concat[GRU(U)(x, initial_state=user_vec), user_vec2]) with a SECOND user table
                                (paper.py:614-621): needs lstur_buffers.user_emb2 / user_m2 / user_v2; the data-parallel
                                tail (lstur_dp_stage) then carries the row gradients of both tables               */
  LSTUR_ARCH_IICGRU = 11     /* concat[GRU(U)(x, initial_state=user_vec), user_vec2], second table, NO Dense: the `iigru`
                                of the pointwise Seq2VecPaperId (paper.py:343-348); a 2U-wide user vector, so it needs
                                head POINT_MLP (or score_model ddot / dnn); same buffers and limits as IIGRU      */
};                           /* vo / niavg have no GRU: gru_w / gru_u / gru_b stay in the bucket with zero gradients */

enum lstur_score_model {     /* Seq2VecPaperSoftmax._score_model (--score-model), paper.py:443-458; listwise head only */
  LSTUR_SCORE_DOT = 0,       /* dot([u, d])                                                                     */
  LSTUR_SCORE_DDOT = 1,      /* dot([tanh(Dense(U)(u)), tanh(Dense(U)(d))]); parameters sdu_w/sdu_b/sdd_w/sdd_b */
  LSTUR_SCORE_DNN = 2        /* Dense(1)(relu(Dense(U)(concat[u, d]))); parameters sn1_w/sn1_b/sn2_w/sn2_b      */
};                           /* test model (lstur_predict): sigmoid(score_model([u, d])), paper.py:661-665      */

enum lstur_head {            /* scorer + loss on top of (user_vec, doc_vec) */
  LSTUR_HEAD_SOFTMAX = 0,    /* listwise: dot, softmax over K1 candidates, categorical_crossentropy  paper.py:446-464,658 */
  LSTUR_HEAD_POINT_MLP = 1,  /* pointwise sibling: concat -> Dense(U,relu) -> Dense(1,sigmoid)      paper.py:223-227   */
  LSTUR_HEAD_POINT_DOT = 2   /* pointwise sibling: sigmoid(u . d)  (Seq2VecPaperDot)                paper.py:258-262   */
};                           /* pointwise loss: Seq2Vec.loss, task/seq2vec.py:213-216 (needs cands == 1)               */

enum lstur_encoder {         /* token stage of the news encoder (--news-encoder) */
  LSTUR_ENC_CNNATT = 0,      /* Conv1D(F, k, same, relu) x (id != 0)                     paper.py:146-155 */
  LSTUR_ENC_MHSA = 1         /* build-defined multi-head self-attention over the title tokens (BASELINE.json configs[4]);
                                the reference has no such encoder (paper.py:119-130), parity is against
                                oracle/lstur_oracle.py: mhsa_fwd only.  [Q|K|V] = E.qkv_w + qkv_b, per head
                                softmax(Q K^T / sqrt(dh)) V with pad keys masked, bf16 MFMA operands / fp32
                                accumulation; filters = heads * dh, dh in {16,32,64}, title_len <= 32; `kernel` is
                                ignored; parameters qkv_w [D, 3F], qkv_b [3F] replace conv_w / conv_b          */
};

/* ABI guard: the caller writes sizeof(lstur_cfg) of ITS build into cfg.struct_size (first field).  Every entry point that
 * takes a lstur_cfg refuses a value that differs from the library's own sizeof(lstur_cfg) with LSTUR_ERR_ARG instead of
 * reading past a shorter struct; lstur_abi_version() (LSTUR_ABI_VERSION of the library build) lets a binding check the
 * rest of this header before it calls anything else. */
#define LSTUR_ABI_VERSION 3

typedef struct lstur_cfg {
  uint32_t struct_size;      /* = sizeof(lstur_cfg) of the caller's build (see above)                  */
  int32_t max_batch;         /* largest n passed to train_step (-b)                                    */
  int32_t window;            /* W  history length (-w / window_size)                                   */
  int32_t title_len;         /* T  (--title-shape)                                                     */
  int32_t cands;             /* K1 = negative_samples + 1 (-n)                                         */
  int32_t vocab;             /* V  rows of the word table (utils.load_textual_embedding)               */
  int32_t word_dim;          /* D  (--textual-embedding-dim); must be a multiple of 4                  */
  int32_t filters;           /* F  (--title-filter-shape F k)                                          */
  int32_t kernel;            /* k                                                                      */
  int32_t user_dim;          /* U  (--user-embedding-dim) = doc-vector dim                             */
  int32_t n_users;           /* rows of the user table = len(ClickData.tsv), paper.py:589              */
  int32_t arch;              /* lstur_arch                                                             */
  float dropout;             /* -d, applied after the embedding and before the attention, paper.py:147,158 */
  uint32_t seed;             /* dropout stream seed                                                    */
  float beta1, beta2, eps;   /* Keras Adam defaults .9 / .999 / 1e-7 (paper.py:657)                    */
  int32_t head;              /* lstur_head; 0 = the listwise model                                     */
  float gain;                /* pointwise loss: weight of the positive term (-g / --gain)              */
  int32_t loss_negatives;    /* pointwise loss: K = negative_samples of Seq2Vec.loss (-n)              */
  int32_t word_trainable;    /* != 0: --textual-embedding-trainable (paper.py:132-138): the word table receives the
                                conv's input gradient (scatter-add over the batch's tokens, row 0 = pad id included)
                                and a dense Keras-Adam update; needs lstur_buffers.word_grad / word_m / word_v   */
  int32_t freeze_encoder;    /* != 0: the news encoder is a loaded, non-trainable model (--enable-pretrain-encoder
                                without --pretrain-encoder-trainable, paper.py:103-107): its backward pass is skipped,
                                its gradients stay 0 (dropout inside it stays active in training, as in Keras)   */
  int32_t news_encoder;      /* lstur_encoder; 0 = cnnatt                                              */
  int32_t heads;             /* mhsa: number of attention heads (filters % heads == 0)                 */
  int32_t score_model;       /* lstur_score_model; 0 = dot                                              */
  float id_keep;             /* probability that a training sample keeps its user embedding (the LSTUR paper's user-
                                embedding masking; reference task/cook.py:139-142: idx_vec * Dropout(1 - id_keep)(mask),
                                kept rows scaled by 1 / id_keep).  <= 0 or >= 1: off.  dgru always masks with 0.5
                                (paper.py:610); iigru masks the same rows of both tables                        */
} lstur_cfg;

/* Sizes (in floats unless named _bytes) the caller must allocate for a configuration. */
typedef struct lstur_sizes {
  size_t word_emb;           /* V*D                                                     */
  size_t dense;              /* flat bucket of all dense parameters (see lstur_param)   */
  size_t user_emb;           /* n_users * Ue  (Ue = U, or U/2 for hgru)                 */
  size_t workspace_bytes;    /* scratch for one train_step / predict at max_batch       */
  int32_t user_emb_dim;      /* Ue                                                      */
  int32_t gru_dim;           /* H                                                       */
  int32_t max_predict_rows;  /* largest n for lstur_predict with this workspace         */
} lstur_sizes;

typedef struct lstur_buffers {
  float* word_emb;           /* [V, D]   word table (Embedding weights, paper.py:132-138); frozen unless word_trainable */
  float* dense_param;        /* [dense]  flat parameter bucket                          */
  float* dense_grad;         /* [dense]  flat gradient bucket (all-reduced by the host under DP) */
  float* dense_m;            /* [dense]  Adam first moment                              */
  float* dense_v;            /* [dense]  Adam second moment                             */
  float* user_emb;           /* [n_users, Ue] trainable user table (paper.py:589)       */
  float* user_m;             /* [n_users, Ue]                                           */
  float* user_v;             /* [n_users, Ue]                                           */
  void* workspace;           /* workspace_bytes, 256-byte aligned                       */
  size_t workspace_bytes;
  float* word_grad;          /* [V, D]  only with cfg.word_trainable (else NULL): gradient of the word table */
  float* word_m;             /* [V, D]  its Adam moments                                 */
  float* word_v;
  float* user_emb2;          /* [n_users, Ue]  only with arch IIGRU (else NULL): the second user table + its moments */
  float* user_m2;
  float* user_v2;
} lstur_buffers;

const char* lstur_last_error(void);
const char* lstur_version(void);
int lstur_abi_version(void);              /* LSTUR_ABI_VERSION the library was built from */

/* Sizes for a configuration (no GPU needed). */
int lstur_query_sizes(const lstur_cfg* cfg, lstur_sizes* out);

/* Named slices of the dense bucket, in Keras-weight terms:
 *   conv_w [k,D,F], conv_b [F]            Conv1D            paper.py:146
 *   att_w [F], att_b [1]                  SimpleAttentionMaskSupport  models.py:456-468
 *   dense_w [F,U], dense_b [U]            Dense(U)          paper.py:159
 *   gru_w [U,3H], gru_u [H,3H], gru_b [3H] GRU (z|r|h)      paper.py:597-613
 *   con_w [H+Ue,U], con_b [U]             Dense after concat (arch gru only) paper.py:599
 *   score_w1 [2U,U], score_b1 [U], score_w2 [U], score_b2 [1]   concat_dense / socre_dense (head POINT_MLP) paper.py:223-227
 * Returns offset/count in floats; LSTUR_ERR_ARG for an unknown name. */
int lstur_param(const lstur_cfg* cfg, const char* name, size_t* offset, size_t* count);
/* i-th parameter name in bucket order, NULL past the end. */
const char* lstur_param_name(const lstur_cfg* cfg, int i);

/* Replaces Seq2VecPaperSoftmaxId._build_model (paper.py:635-665): binds caller-owned buffers. */
int lstur_create(const lstur_cfg* cfg, const lstur_buffers* buf, lstur_ctx** out);
int lstur_destroy(lstur_ctx* ctx);

/* Optional resident title table int32 [n_docs, T] (the reference keeps titles on the host,
 * task/seq2vec.py:85-110); enables the *_ids entry points. */
int lstur_set_title_table(lstur_ctx* ctx, const int32_t* titles, int32_t n_docs);

/* One forward + backward of model.train_on_batch (inside fit_generator, main.py:73-78) WITHOUT
 * the optimizer update: fills dense_grad (mean over the n samples, scaled by grad_scale) and the
 * per-sample user-row gradients.
 *   user  int32 [n]; hist int32 [n,W,T]; cands int32 [n,K1,T]   (token ids; 0 = pad)
 *   step  = global step counter (selects the dropout stream)
 *   stats = 4 device floats: [0] mean categorical_crossentropy, [1] categorical_accuracy (paper.py:655-659) of
 *           this batch; [2], [3] += the same two values (running sums the host zeroes per epoch, so that Keras'
 *           epoch means need no extra launch per step)
 * grad_scale lets data-parallel callers pre-divide by the world size (1.0 otherwise). */
int lstur_forward_backward(lstur_ctx* ctx, const int32_t* user, const int32_t* hist, const int32_t* cands,
                           int32_t n, uint32_t step, float grad_scale, float* stats, lstur_stream stream);
/* Same with doc ids: hist_ids int32 [n,W], cand_ids int32 [n,K1] into the resident title table. */
int lstur_forward_backward_ids(lstur_ctx* ctx, const int32_t* user, const int32_t* hist_ids,
                               const int32_t* cand_ids, int32_t n, uint32_t step, float grad_scale,
                               float* stats, lstur_stream stream);
/* Optional software pipelining across steps: announce the doc-id batch of the NEXT training step (its hist_ids / cand_ids
 * device arrays, n and step number) BEFORE calling lstur_forward_backward_ids for the current one.  The library then
 * stages that batch -- title expansion, compaction of the non-pad tokens, embedding gather incl. its dropout mask -- on
 * a stream of its own while the current step's GRU recurrence runs (a latency chain on 16 CUs: HBM and 240 CUs idle), and
 * the next lstur_forward_backward_ids call with the SAME arrays, n and step picks the staged batch up instead of staging
 * it inline.  The arrays must stay alive and unchanged until that call; any other call sequence simply stages inline.
 * Same kernels on the same data: results are bit-identical with and without the announcement.  No-op for configurations
 * that stage inline anyway (trainable word table, exact-fp32 conv path, pointwise heads, mhsa). */
int lstur_prefetch_ids(lstur_ctx* ctx, const int32_t* hist_ids, const int32_t* cand_ids, int32_t n, uint32_t step);
/* ... and the user ids of that next batch (call BEFORE lstur_prefetch_ids; optional): with a lazily updated table
 * (lstur_flush_tables) the rows that batch will read are brought up to date under the current step's recurrence as well. */
int lstur_prefetch_rows(lstur_ctx* ctx, const int32_t* user);
/* how many training calls picked up a prefetched batch so far (diagnostics / tests) */
int32_t lstur_prefetch_hits(lstur_ctx* ctx);

/* Per-sample sparse user-table gradient of the last forward_backward: rows int32 [n], grads [n,Ue]
 * (device pointers into the workspace; what a DP host all-gathers). */
int lstur_user_grad(lstur_ctx* ctx, const int32_t** rows, const float** grads, int32_t* n);

/* Data parallelism with ONE collective per step (the reference is single-device).  The caller allocates
 * dense_grad with lstur_dp_tail_floats(cfg, world) extra floats behind the `dense` gradients.  After
 * forward_backward (grad_scale = 1/world) every rank calls lstur_dp_stage: it writes its n (row number, gradient)
 * pairs into slot `rank` of that tail and zeroes the other slots, so ONE all-reduce(sum) of the whole buffer both
 * averages the dense gradients and concatenates the sparse rows of all ranks.  lstur_dp_unpack then yields rows /
 * grads device pointers (rank order, every rank passed the same n) for lstur_adam_step(..., world * n). */
size_t lstur_dp_tail_floats(const lstur_cfg* cfg, int32_t world);
int lstur_dp_stage(lstur_ctx* ctx, float* tail, int32_t rank, int32_t world, lstur_stream stream);
int lstur_dp_unpack(lstur_ctx* ctx, const float* tail, int32_t world, int32_t n, const int32_t** rows,
                    const float** grads, lstur_stream stream);
/* The same exchange without launches of its own: after lstur_dp_attach(ctx, tail, rank, world) every forward_backward
 * stages the tail inside its final reduction launch (no lstur_dp_stage; tail == NULL detaches), and
 * lstur_adam_step_dp(ctx, lr, t, tail, world, n) takes the REDUCED tail as it stands -- with full batches (n == max_batch)
 * the row-gradient matrix is read in place and the row numbers are converted inside the dense Adam launch; a ragged
 * last batch goes through lstur_dp_unpack internally.  The tail is the caller's memory: it must stay valid, and hold
 * lstur_dp_tail_floats(cfg, world) floats, for as long as it is attached. */
int lstur_dp_attach(lstur_ctx* ctx, float* tail, int32_t rank, int32_t world);
int lstur_adam_step_dp(lstur_ctx* ctx, float lr, int32_t t, const float* tail, int32_t world, int32_t n, lstur_stream stream);

/* keras.optimizers.Adam update (paper.py:657) of the dense bucket from dense_grad, and the DENSE
 * update of the whole user table (TF-1 densifies the IndexedSlices gradient) from n_rows sparse
 * rows (duplicates are summed in ascending order).  t = 1-based Adam iteration. */
int lstur_adam_step(lstur_ctx* ctx, float lr, int32_t t, const int32_t* rows, const float* row_grads,
                    int32_t n_rows, lstur_stream stream);

/* Large user tables (n_users >= 200 000; LSTUR_LAZY_ADAM=0 / 1 overrides) are updated LAZILY: TF-1 Keras densifies the
 * embedding gradient, so every row takes an Adam update every step -- for a row outside the batch the zero-gradient one.
 * Instead of sweeping the table every step (24 B of HBM traffic per element: 1.1 of 7 ms at 1 M users) a row takes its
 * pending zero-gradient updates, in order and with each iteration's lr_t, when it is next read by a forward or updated with a
 * gradient: the same fp32 operations per element, bit-identical values, computed later.  lstur_flush_tables brings every row
 * of user_emb / user_m / user_v (and the second table) up to date -- call it before reading those buffers on the host.
 * lstur_lazy_adam: 1 when the context updates lazily.  The Adam iteration t must advance by one per step (anything else
 * flushes and re-bases). */
int lstur_flush_tables(lstur_ctx* ctx, lstur_stream stream);
int lstur_lazy_adam(lstur_ctx* ctx);
/* The caller wrote parameters (dense_param, word_emb) from outside -- set_weights, a checkpoint load: the weight images
 * the library keeps for its MFMA kernels (packed behind every Adam step, for the NEXT forward) and a batch staged by
 * lstur_prefetch_ids are stale.  Not needed after lstur_adam_step / lstur_train_step. */
int lstur_params_changed(lstur_ctx* ctx);

/* model.train_on_batch in ONE call for a single process (no gradient exchange between backward and optimizer):
 * lstur_(point_)forward_backward(_ids) with grad_scale 1 followed by lstur_adam_step(lr, t, own rows) -- same arithmetic,
 * bit-identical results -- but scheduled as one unit: the dense sweep of the user table(s) starts when the recurrence has
 * produced the row gradients and runs on a stream of its own under the news encoder's backward.  `labels` is read by the
 * pointwise heads only (NULL for the listwise head).  (main.py:73-78: fit_generator -> train_on_batch.) */
int lstur_train_step(lstur_ctx* ctx, const int32_t* user, const int32_t* hist, const int32_t* cands, const float* labels,
                     int32_t n, uint32_t step, float lr, int32_t t, float* stats, lstur_stream stream);
int lstur_train_step_ids(lstur_ctx* ctx, const int32_t* user, const int32_t* hist_ids, const int32_t* cand_ids,
                         const float* labels, int32_t n, uint32_t step, float lr, int32_t t, float* stats,
                         lstur_stream stream);

/* test_model.predict (paper.py:661-665; task/seq2vec.py:206): sigmoid(u . d), one candidate per row.
 *   user int32 [n]; hist int32 [n,W,T]; cand int32 [n,T]; scores float [n]; n <= max_predict_rows */
int lstur_predict(lstur_ctx* ctx, const int32_t* user, const int32_t* hist, const int32_t* cand,
                  int32_t n, float* scores, lstur_stream stream);
int lstur_predict_ids(lstur_ctx* ctx, const int32_t* user, const int32_t* hist_ids, const int32_t* cand_ids,
                      int32_t n, float* scores, lstur_stream stream);

/* model.evaluate_generator step (main.py:83-87): forward only, no dropout; writes stats[0..1] only. */
int lstur_evaluate(lstur_ctx* ctx, const int32_t* user, const int32_t* hist, const int32_t* cands,
                   int32_t n, float* stats, lstur_stream stream);

/* Pointwise sibling (Seq2VecPaper / Seq2VecPaperDot / Seq2VecPaperId, paper.py:228-256, 356-383; contexts created
 * with head POINT_MLP or POINT_DOT and cands == 1): one candidate and one label per sample.
 *   user int32 [n]; hist int32 [n,W,T]; cand int32 [n,T]; labels float [n] in {0,1}
 *   stats = 4 device floats (as above): Seq2Vec.loss (task/seq2vec.py:213-216) and the fraction of (p > 0.5) == label;
 *   the probabilities of the last call are readable through lstur_peek("probs") for the host-side AUC metric.
 * lstur_predict(_ids) returns the scorer's sigmoid output for these heads; lstur_adam_step is shared. */
int lstur_point_forward_backward(lstur_ctx* ctx, const int32_t* user, const int32_t* hist, const int32_t* cand,
                                 const float* labels, int32_t n, uint32_t step, float grad_scale, float* stats,
                                 lstur_stream stream);
int lstur_point_forward_backward_ids(lstur_ctx* ctx, const int32_t* user, const int32_t* hist_ids,
                                     const int32_t* cand_ids, const float* labels, int32_t n, uint32_t step,
                                     float grad_scale, float* stats, lstur_stream stream);
int lstur_point_evaluate(lstur_ctx* ctx, const int32_t* user, const int32_t* hist, const int32_t* cand,
                         const float* labels, int32_t n, float* stats, lstur_stream stream);

/* Measurement (bench.py): when enabled, each kernel family of the step is bracketed by HIP events on
 * the launch stream.  lstur_profile(ctx, 1) also resets the counters.  lstur_profile_read SYNCHRONISES on
 * the recorded events (the only call that does) and returns the summed duration and launch count of the
 * index-th family; it returns 1 past the last family.  Families: gather, conv_fwd, attpool_fwd, gru_fwd,
 * gru_bwd, attpool_bwd, conv_wgrad, adam_user_table, forward_total, backward_total.
 * enable = 2: only the roofline kernels (gather, conv_fwd, conv_wgrad; mhsa: qkv_proj, qkv_wgrad) are bracketed -- every
 * bracket costs the step ~6 us (two event records that also drain the launch pipeline), so the timed region of the
 * benchmark carries these three only and the full breakdown comes from a separate, untimed run of steps. */
int lstur_profile(lstur_ctx* ctx, int enable);
int lstur_profile_read(lstur_ctx* ctx, int index, const char** name, float* total_ms, int32_t* launches);

/* How the Conv1D contractions of this context run: 0 = none (mhsa encoder); 3 = "f16x2" (the default: both operands as two
 * fp16 parts of the power-of-two scaled value, three fp16 MFMA products per fp32-grade product); 1 = "bf16x3" (three bf16
 * parts, six products: trainable word table, or LSTUR_CONV_SCHEME=bf16x3) -- both in mnexp_amd/csrc/convx.hip, both
 * fp32-grade (error of the order of an fp32 FMA chain in another order), not reduced-precision kernels; 2 = exact fp32 MFMA
 * (mnexp_amd/csrc/conv.hip: filters > 448, kernel > 8, or LSTUR_CONV_F32 set in the environment).  Negative: error. */
int lstur_conv_scheme(lstur_ctx* ctx);

/* Debug/inspection: device pointers to intermediates of the last call (valid until the next call).
 * names: "doc_vec" [N_t,U], "user_vec" [n,U], "logits" [n,K1], "probs" [n,K1], "conv_out" [N_t,T,F] (rows of
 *        non-pad tokens only), "att_w" [N_t,T], "emb" [N_t,T+k-1,D], "h_last" [n,H], "dgates" [n,W,3H] */
int lstur_peek(lstur_ctx* ctx, const char* name, const float** ptr, size_t* count);

/* ---- per-kernel entry points (unit parity tests, micro-benchmarks) --------------------------- */

/* keras.layers.Embedding gather (+ inverted Dropout, paper.py:132-147): out[i,:] = table[ids[i],:].
 * With pad_rows = k-1 the output is the zero-padded per-title layout [N_t, T+k-1, D] the conv reads.
 * key = 0 and drop = 0 disables dropout. */
int lstur_gather_f32(const float* table, const int32_t* ids, float* out, int32_t n_titles, int32_t title_len,
                     int32_t dim, int32_t pad_left, int32_t pad_right, float drop, uint32_t key,
                     lstur_stream stream);
/* Stable list of the non-pad tokens of a batch: vidx[0 .. *count) = ascending flat indices i with ids[i] != 0.
 * The conv output of a pad token is multiplied by 0 (paper.py:150-155) and its gradient is 0, so the two conv
 * contractions may run over this list only (identical results).  scratch: lstur_valid_tokens_scratch bytes. */
size_t lstur_valid_tokens_scratch(int32_t n_tokens);
int lstur_valid_tokens(const int32_t* ids, int32_t n_tokens, int32_t* vidx, int32_t* count, int32_t* scratch,
                       lstur_stream stream);
/* Conv1D('same', relu) * (id != 0)  (paper.py:146-155) on the padded embedding layout.
 * w_packed from lstur_pack_b(conv_w as [k*D, F]).  emb_padded needs 256 floats of zero slack at its end.
 * vidx/vcount (device, may both be NULL = all rows): only the listed output rows are computed and written. */
int lstur_conv1d_relu_mask_fwd(const float* emb_padded, const float* w_packed, const float* bias,
                               const int32_t* ids, float* out, int32_t n_titles, int32_t title_len,
                               int32_t dim, int32_t ksize, int32_t filters, const int32_t* vidx,
                               const int32_t* vcount, lstur_stream stream);
/* ---- The Conv1D contractions the training step uses: fp32-GRADE products on the 16-bit matrix cores with BOTH operands
 * pre-split into `parts` 16-bit pieces (mnexp_amd/csrc/convx.hip; Embedding + Dropout + Conv1D of task/paper.py:132-155):
 *   parts = 2 "f16x2":  x s = h + l, two fp16 values of the power-of-two scaled element; products hh + hl + lh; the scales
 *                       (embeddings: from max |word table|, weight: from max |conv_w|, gradient: from max |dout|) live in a
 *                       device block of lstur_convx_scales_bytes() bytes that the calls below read and maintain;
 *   parts = 3 "bf16x3": x = h + m + l, three bf16 values; six products; `scales` may be NULL.
 * Error ~2^-22 of |a||b| per product, random in sign -- below what the fp32 accumulation of a k*D-long dot product rounds
 * away; tests/test_gpu_kernels.py holds both to the bound of the exact-fp32 MFMA kernel against fp64.
 * Layouts: "Ex" = [title][title_len + ksize - 1][ceil8(dim) / 8][part][8] 16-bit (lstur_convx_e_elems elements incl. slack;
 * written by lstur_convx_gather: out = table[ids] (x dropout mask / (1 - drop)) (x scale), zero rows left and right;
 * skip_unread != 0: rows that no window of a non-pad token covers are left untouched -- ex must then hold finite values;
 * live (or NULL; title_len <= 32): bit t of live[title] = token t is not a pad -- with skip_unread the kernel then takes the
 * covered rows from these words instead of reading the neighbours' ids);
 * weight image wx = lstur_convx_pack_w(conv_w [ksize * dim, filters]) of lstur_convx_w_elems elements.
 * Call order for parts = 2: lstur_convx_scales_init, lstur_convx_embedding_scale(table), lstur_convx_pack_w, then gather /
 * forward / kernel gradient in any order. */
size_t lstur_convx_scales_bytes(void);
int lstur_convx_scales_init(float* scales, lstur_stream stream);
int lstur_convx_embedding_scale(const float* table, size_t count, float drop, float* scales, lstur_stream stream);
size_t lstur_convx_e_elems(int32_t n_titles, int32_t title_len, int32_t dim, int32_t ksize, int32_t parts);
int lstur_convx_gather(const float* table, const int32_t* ids, uint16_t* ex, int32_t n_titles, int32_t title_len, int32_t dim,
                       int32_t pad_left, int32_t pad_right, float drop, uint32_t key, int32_t skip_unread, int32_t parts,
                       const float* scales, const uint32_t* live, lstur_stream stream);
size_t lstur_convx_w_elems(int32_t dim, int32_t ksize, int32_t filters, int32_t parts);
int lstur_convx_pack_w(const float* w, int32_t dim, int32_t ksize, int32_t filters, uint16_t* wx, int32_t parts, float* scales,
                       lstur_stream stream);
/* Conv1D('same', relu) * (id != 0) on Ex; vidx / vcount as in lstur_conv1d_relu_mask_fwd.  scratch: lstur_convx_fwd_scratch()
 * bytes for the k-sliced tiles of the last partial round of workgroups, or NULL (every tile runs whole). */
size_t lstur_convx_fwd_scratch(void);
int lstur_convx_fwd(const uint16_t* ex, const uint16_t* wx, const float* bias, const int32_t* ids, float* out, int32_t n_titles,
                    int32_t title_len, int32_t dim, int32_t ksize, int32_t filters, const int32_t* vidx, const int32_t* vcount,
                    int32_t parts, const float* scales, void* scratch, lstur_stream stream);
/* ... and its kernel gradient dw [ksize * dim, filters] = im2col(E)^T . dout on the same pre-split embeddings (dout fp32
 * [n_titles * title_len, filters]; rows of pad tokens are not read when the compacted list is given; for parts = 2 this
 * stand-alone call first takes max |dout| over the rows it reads).  scratch: lstur_convx_wgrad_scratch() bytes;
 * offsets: 2 * n_titles * title_len int32 of scratch (per-token window / row offsets, filled by the call). */
size_t lstur_convx_wgrad_scratch(int32_t n_titles, int32_t title_len, int32_t dim, int32_t ksize, int32_t filters);
int lstur_convx_wgrad(const uint16_t* ex, const float* dout, float* dw, void* scratch, int32_t* offsets, int32_t n_titles,
                      int32_t title_len, int32_t dim, int32_t ksize, int32_t filters, const int32_t* vidx,
                      const int32_t* vcount, int32_t parts, float* scales, lstur_stream stream);
/* Conv1D kernel gradient: dW[k*D,F] = im2col(emb)^T . dOut ; scratch >= lstur_conv1d_wgrad_scratch() bytes. */
size_t lstur_conv1d_wgrad_scratch(int32_t dim, int32_t ksize, int32_t filters);
int lstur_conv1d_wgrad(const float* emb_padded, const float* dout, float* dw, void* scratch,
                       int32_t n_titles, int32_t title_len, int32_t dim, int32_t ksize, int32_t filters,
                       const int32_t* vidx, const int32_t* vcount, lstur_stream stream);
/* Pack a row-major B[K,N] (leading dim ldb) into the MFMA operand image [ceil(K/16)][ceil16(N)][16]. */
size_t lstur_pack_b_floats(int32_t K, int32_t N);
int lstur_pack_b(const float* B, int32_t ldb, int32_t K, int32_t N, float* packed, lstur_stream stream);
/* Same image for B = Bt^T, Bt stored row-major [N, K] with leading dim ldb. */
int lstur_pack_bt(const float* Bt, int32_t ldb, int32_t K, int32_t N, float* packed, lstur_stream stream);
/* Masking + Dropout + SimpleAttentionMaskSupport (paper.py:157-158, models.py:474-489).
 * conv_out [N_t,T,F] -> pooled [N_t,F]; saves att_tanh [N_t,T], att_weight [N_t,T].
 * ids (may be NULL): rows of pad tokens are taken as zero without being read (they are zero by definition). */
int lstur_attpool_fwd(const float* conv_out, const int32_t* ids, const float* att_w, const float* att_b,
                      float* pooled, float* att_tanh, float* att_weight, int32_t n_titles, int32_t title_len,
                      int32_t filters, float drop, uint32_t key, lstur_stream stream);
/* Backward of the above + ReLU'; dconv [N_t,T,F]; partial sums [n_part, 2F+1] reduced by the caller
 * through lstur_colsum: layout per row = datt_w[F] | dconv_b[F] | datt_b.  row_scratch: N_t*T floats. */
int32_t lstur_attpool_bwd_parts(int32_t n_titles);
int lstur_attpool_bwd(const float* conv_out, const int32_t* ids, const float* att_w, const float* att_tanh,
                      const float* att_weight, const float* dpooled, float* dconv, float* parts,
                      float* row_scratch, int32_t n_titles, int32_t title_len, int32_t filters, float drop,
                      uint32_t key, lstur_stream stream);
/* Token stage of the build-defined `mhsa` news encoder (LSTUR_ENC_MHSA; no reference counterpart, the specification
 * is oracle/lstur_oracle.py: mhsa_fwd / mhsa_bwd).  qkv [N_t*T, 3F] = Q | K | V, head h = columns h*dh .. h*dh+dh-1
 * of each third; per (title, head): P = softmax_keys(Q K^T / sqrt(dh)) with the keys of pad tokens (ids == 0) masked,
 * out = P V.  MFMA operands are rounded to bf16, accumulation / softmax fp32.  Rows of pad tokens: `out` is not
 * written, `dout` is not read, `dqkv` is written as 0.  title_len <= 32, filters <= 512, filters/heads in {16,32,64}. */
int lstur_mhsa_fwd(const float* qkv, const int32_t* ids, float* out, int32_t n_titles, int32_t title_len,
                   int32_t filters, int32_t heads, lstur_stream stream);
int lstur_mhsa_bwd(const float* qkv, const float* dout, const int32_t* ids, float* dqkv, int32_t n_titles,
                   int32_t title_len, int32_t filters, int32_t heads, lstur_stream stream);
/* Projections of the mhsa encoder on bf16 MFMA (operands rounded to nearest-even bf16, fp32 accumulation):
 *   lstur_proj_fwd_bf16  : out bf16 [M, N] = E[M, K] . W[K, N] + bias[N]   (E, W, bias fp32; bias may be NULL)
 *   lstur_proj_wgrad_bf16: dw fp32 [D + 1, N] = [E | 1]^T . dout           (E fp32 [K, D], dout bf16 [K, N]; row D of dw
 *                          = column sums of dout = the bias gradient); scratch: lstur_proj_wgrad_scratch(D, N, K) bytes.
 * N % 8 == 0; K % 4 == 0 (forward); D % 4 == 0 and D <= 316 (gradient). */
int lstur_proj_fwd_bf16(const float* E, const float* W, const float* bias, uint16_t* out, int32_t M, int32_t N, int32_t K,
                        lstur_stream stream);
size_t lstur_proj_wgrad_scratch(int32_t D, int32_t N, int32_t K);
int lstur_proj_wgrad_bf16(const float* E, const uint16_t* dout, float* dw, void* scratch, int32_t D, int32_t N, int32_t K,
                          lstur_stream stream);
/* out[j] (+)= sum_i x[i, j]   (bias gradients, partial-sum reduction) */
int lstur_colsum(const float* x, int32_t rows, int32_t cols, int32_t ld, float* out, float scale,
                 lstur_stream stream);
/* C[M,N] = alpha * op(A) . op(B) + bias (row broadcast) ; op = transpose when ta/tb != 0.
 * scratch (split-K partials) may be NULL for small K; size from lstur_gemm_scratch. */
size_t lstur_gemm_scratch(int32_t M, int32_t N, int32_t K);
int lstur_gemm_f32(const float* A, int32_t lda, int32_t ta, const float* B, int32_t ldb, int32_t tb,
                   float* C, int32_t ldc, const float* bias, int32_t M, int32_t N, int32_t K,
                   float alpha, void* scratch, lstur_stream stream);
/* C[M,N] = A[M,K] . op(B) + bias with fp32 operands, fp32-grade on the bf16 matrix cores (three bf16 parts per operand,
 * six products, fp32 accumulate: csrc/gemm3.hip) -- the kernel of the Dense(U) / GRU input projections on the step's
 * critical path (paper.py:159, 592-613).  b_nk != 0: B is stored [N][K] (C = A . B^T), else [K][N].  K, lda, ldb must be
 * multiples of 4 (LSTUR_ERR_ARG otherwise: use lstur_gemm_f32). */
int lstur_gemm_bf16x3(const float* A, int32_t lda, const float* B, int32_t ldb, int32_t b_nk, float* C, int32_t ldc,
                      const float* bias, int32_t M, int32_t N, int32_t K, lstur_stream stream);
/* The parameter-gradient form of the same kernel: C[M,N] = A^T . B with A [K][M], B [K][N] (K = tokens / history slots),
 * split along K into slabs (scratch: lstur_gemm_bf16x3_tn_scratch bytes) that are summed in slab order. */
size_t lstur_gemm_bf16x3_tn_scratch(int32_t M, int32_t N, int32_t K);
int lstur_gemm_bf16x3_tn(const float* A, int32_t lda, const float* B, int32_t ldb, float* C, int32_t ldc, int32_t M, int32_t N,
                         int32_t K, void* scratch, lstur_stream stream);
/* Operand images of the recurrent matrix gru_u [H,3H] (z|r|h) for the two GRU kernels, written in one launch into
 * ONE opaque buffer of lstur_gru_pack_floats(H) floats (re-pack after every update of gru_u). */
size_t lstur_gru_pack_floats(int32_t H);
int lstur_gru_pack_u(const float* gru_u, int32_t H, float* packed, lstur_stream stream);
/* Masked Keras-2.2.4 GRU over W steps (paper.py:592,597-613; K.rnn mask semantics).
 *   xw [n,W,3H] input projection incl. bias (row-major, gates z|r|h); gmask uint8 [n,W]; h0 [n,H] or NULL (zeros)
 *   hlast [n,H] = output of the last step.  tape: lstur_gru_tape_floats(n,W,H) floats of scratch that carries the
 *   saved activations to lstur_gru_bwd (layout opaque to the caller).  H <= 256; H <= 200 runs the kernels that
 *   keep the recurrent matrix resident in registers + LDS (set LSTUR_GRU_GENERIC=1 to force the general ones); their
 *   forward computes the recurrent products as fp32-grade fp16 pairs (three v_mfma_f32_4x4x4_16b_f16 products on
 *   power-of-two scaled operands, like the conv kernels; LSTUR_GRU_F32=1: fp32 MFMAs) -- pack and forward read the switch. */
size_t lstur_gru_tape_floats(int32_t n, int32_t W, int32_t H);
int lstur_gru_fwd(const float* xw, const uint8_t* gmask, const float* h0, const float* packed, float* tape,
                  float* hlast, int32_t n, int32_t W, int32_t H, lstur_stream stream);
/* BPTT over the tape of the matching lstur_gru_fwd call: dh_last [n,H] ->
 *   dgates [n,W,3H] (gradient wrt the gate pre-activations z|r|h), rh [n,W,H] (= r * h_prev, operand of dU_h),
 *   hprev [n,W,H] (= h_{t-1}, operand of dU_zr), dh0 [n,H] (gradient wrt the initial state). */
int lstur_gru_bwd(const float* dh_last, const uint8_t* gmask, float* tape, const float* packed, float* dgates,
                  float* rh, float* hprev, float* dh0, int32_t n, int32_t W, int32_t H, lstur_stream stream);
/* dot scorer + softmax + categorical_crossentropy (paper.py:446-447,463,658) and its gradient.
 *   u [n,U]; d [n,K1,U]; outputs logits/probs [n,K1], loss_i [n], correct_i [n], du [n,U], dd [n,K1,U]
 *   (du/dd may be NULL: forward only).  inv_n scales the gradient (1/batch * grad_scale). */
int lstur_score_softmax_ce(const float* u, const float* d, float* logits, float* probs, float* loss_i,
                           float* correct_i, float* du, float* dd, int32_t n, int32_t K1, int32_t U,
                           float inv_n, lstur_stream stream);
/* Keras Adam on a flat fp32 array. */
int lstur_adam_flat(float* p, const float* g, float* m, float* v, size_t count, float lr_t, float beta1,
                    float beta2, float eps, lstur_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* LSTUR_H */
