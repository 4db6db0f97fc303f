// C-ABI of liblstur_hip.so (include/lstur.h): context, workspace carving and the per-step kernel
// sequence that replaces the Keras graph of Seq2VecPaperSoftmaxId._build_model
// (reference task/paper.py:635-665).  No allocation, no synchronisation, no torch types.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/lstur.h"
#include "lstur_device.h"
#include "lstur_internal.h"

using namespace lstur;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

struct ParamSlot {
  const char* name;
  size_t offset, count;
};

struct Dims {
  int B, W, T, K1, V, D, F, k, U, Nu, arch, H, Ue, Tp, KD, head, word_trainable, enc, heads, Uu, score;
  int nt_max;   // titles per call
  int n_cap;    // per-sample buffers
  int rows_cap; // sparse user-grad rows per adam step
  int lazy;     // lazy dense Adam of the user table(s) (large tables; misc.hip)
  int np;       // split-operand conv kernels (convx.hip): 16-bit parts per element, 2 = f16x2 (default), 3 = bf16x3 (trainable
                // word table: its maximum moves every step; LSTUR_CONV_SCHEME=bf16x3); 0: exact-fp32 MFMA kernels of conv.hip
                // (filters > 448, kernel > 8, LSTUR_CONV_F32=1) or no conv at all (mhsa)
};
constexpr int kLazyLog = 1 << 16;        // logged lr_t values (a full flush at the latest every kLazyLog - 2 iterations)
constexpr int kLazyMinRows = 200000;     // tables from this size on update lazily (LSTUR_LAZY_ADAM=0 / 1 overrides)

// iigru / iicgru: a SECOND user table, gathered with the same rows (paper.py:614-621, 343-348)
static inline bool two_tables(const Dims& d) { return d.arch == LSTUR_ARCH_IIGRU || d.arch == LSTUR_ARCH_IICGRU; }

constexpr size_t kParamAlign = 64;  // floats

// Environment switches of the step orchestration.  All of them select between variants that compute the SAME results (A/B
// measurements, tests/): they are read once, when the context is created, so a step can never see two settings.
struct Opts {
  bool conv_f32;       // LSTUR_CONV_F32=1: exact-fp32 MFMA conv kernels (conv.hip) instead of the split-operand ones
  bool conv_bf16x3;    // LSTUR_CONV_SCHEME=bf16x3: three bf16 parts / six products instead of two fp16 parts / three products
  bool wg_stream;      // LSTUR_WG_STREAM=0: parameter-gradient GEMMs on the caller's stream
  bool cand_split;     // LSTUR_CAND_SPLIT=0: one launch per kernel over all titles in the backward pass
  bool fused_adam;     // LSTUR_FUSED_ADAM=0: lstur_train_step = forward_backward + adam_step, unscheduled
  bool adam_pack;      // LSTUR_ADAM_PACK=0: weight images packed by their own launches behind the Adam
  bool dp_defer;       // LSTUR_DP_DEFER=0: the data-parallel step sweeps the user table inside lstur_adam_step
  bool prof_attach;    // LSTUR_PROF_ATTACH=0: timing brackets recorded around the kernels instead of riding them
  bool wgrad_bf16x3;   // LSTUR_WGRAD_BF16X3=1: parameter-gradient GEMMs on the bf16x3 kernel
  bool gru_f16;        // LSTUR_GRU_F32=1: the recurrence on fp32 4x4x1 MFMAs instead of fp16-pair 4x4x4 ones (gru4.hip)
  bool score_fuse;     // LSTUR_SCORE_FUSE=0: the listwise scorer as a launch of its own behind the recurrence
  int sweep_blocks;    // LSTUR_SWEEP_BLOCKS=n: workgroups of the thin table sweep (-1: planned)
  int prefetch_blocks; // LSTUR_PREFETCH_BLOCKS=n: workgroups of a prefetched gather (0: all)
  int lazy_log;        // LSTUR_LAZY_LOG=n (power of two): entries of the lazy table's lr_t ring (tests: force wraps)
  static bool is(const char* name, const char* v) {
    const char* e = getenv(name);
    return e && !strcmp(e, v);
  }
  static Opts read() {
    Opts o;
    o.conv_f32 = getenv("LSTUR_CONV_F32") != nullptr;
    o.conv_bf16x3 = is("LSTUR_CONV_SCHEME", "bf16x3");
    o.wg_stream = !is("LSTUR_WG_STREAM", "0");
    o.cand_split = !is("LSTUR_CAND_SPLIT", "0");
    o.fused_adam = !is("LSTUR_FUSED_ADAM", "0");
    o.adam_pack = !is("LSTUR_ADAM_PACK", "0");
    o.dp_defer = !is("LSTUR_DP_DEFER", "0");
    o.prof_attach = !is("LSTUR_PROF_ATTACH", "0");
    o.wgrad_bf16x3 = getenv("LSTUR_WGRAD_BF16X3") != nullptr;
    o.gru_f16 = gru4_f16_env();
    o.score_fuse = !is("LSTUR_SCORE_FUSE", "0");
    o.sweep_blocks = getenv("LSTUR_SWEEP_BLOCKS") ? atoi(getenv("LSTUR_SWEEP_BLOCKS")) : -1;
    o.prefetch_blocks = getenv("LSTUR_PREFETCH_BLOCKS") ? atoi(getenv("LSTUR_PREFETCH_BLOCKS")) : 0;
    o.lazy_log = kLazyLog;
    if (const char* e = getenv("LSTUR_LAZY_LOG")) {
      const int v = atoi(e);
      if (v >= 8 && v <= kLazyLog && (v & (v - 1)) == 0) o.lazy_log = v;
    }
    return o;
  }
};


bool make_dims(const lstur_cfg* c, Dims* d, std::string* why) {
  if (!c) {
    *why = "cfg is NULL";
    return false;
  }
  if (c->struct_size != (uint32_t)sizeof(lstur_cfg)) {   // a caller built against another lstur.h: refuse, never read past it
    *why = "lstur_cfg.struct_size is " + std::to_string(c->struct_size) + ", this library's sizeof(lstur_cfg) is " +
           std::to_string(sizeof(lstur_cfg)) + " (ABI version " + std::to_string(LSTUR_ABI_VERSION) + "): rebuild the binding from include/lstur.h";
    return false;
  }
  d->B = c->max_batch; d->W = c->window; d->T = c->title_len; d->K1 = c->cands; d->V = c->vocab;
  d->D = c->word_dim; d->F = c->filters; d->k = c->kernel; d->U = c->user_dim; d->Nu = c->n_users;
  d->arch = c->arch;
  if (d->B < 1 || d->W < 1 || d->T < 1 || d->K1 < 1 || d->V < 1 || d->D < 1 || d->F < 1 || d->k < 1 || d->U < 1 ||
      d->Nu < 1) {
    *why = "all dimensions must be >= 1";
    return false;
  }
  if (d->D % 4) {
    *why = "word_dim must be a multiple of 4 (16-byte embedding rows)";
    return false;
  }
  if (d->K1 > 64) {
    *why = "cands (negative_samples + 1) must be <= 64";
    return false;
  }
  if (d->F > 1024) {
    *why = "filters must be <= 1024";
    return false;
  }
  if (d->arch < LSTUR_ARCH_IGRU || d->arch > LSTUR_ARCH_IICGRU) {
    *why = "unknown arch";
    return false;
  }
  if (d->arch == LSTUR_ARCH_HGRU && (d->U % 2)) {
    *why = "hgru needs an even user_dim";
    return false;
  }
  d->enc = c->news_encoder;
  d->heads = c->heads;
  if (d->enc != LSTUR_ENC_CNNATT && d->enc != LSTUR_ENC_MHSA) {
    *why = "unknown news_encoder";
    return false;
  }
  if (d->enc == LSTUR_ENC_MHSA) {
    d->k = 1;   // no neighbourhood: the embeddings are a plain [N_tok, D] matrix
    if (!mhsa_supported(d->T, d->F, d->heads)) {
      *why = "mhsa needs title_len <= 32, filters <= 512, filters % heads == 0 and filters / heads in {16, 32, 64}";
      return false;
    }
    if (c->word_trainable) {
      *why = "mhsa: a trainable word table is not on the native path";
      return false;
    }
    if (d->D > 316) {
      *why = "mhsa: word_dim must be <= 316 (one 320-row tile holds the projection's weight + bias gradient)";
      return false;
    }
    const size_t ntok = (size_t)d->B * (d->W + d->K1) * d->T;
    if (ntok * 3 * d->F * 2 >= ((size_t)1 << 32) || ntok * d->D * 4 >= ((size_t)1 << 32)) {
      *why = "mhsa: max_batch too large (the projection kernels address a step's tokens through 32-bit offsets)";
      return false;
    }
  }
  d->word_trainable = c->word_trainable != 0;
  if (d->word_trainable && c->freeze_encoder) {
    *why = "word_trainable and freeze_encoder exclude each other";
    return false;
  }
  if (d->word_trainable && (c->filters % 4)) {
    *why = "word_trainable needs filters % 4 == 0";
    return false;
  }
  d->head = c->head;
  if (d->head < LSTUR_HEAD_SOFTMAX || d->head > LSTUR_HEAD_POINT_DOT) {
    *why = "unknown head";
    return false;
  }
  if (d->head != LSTUR_HEAD_SOFTMAX) {
    if (d->K1 != 1) {
      *why = "the pointwise heads take one candidate per sample (cands == 1)";
      return false;
    }
    // Seq2VecPaperId `gru` / `iigru`: 2U-wide concatenations in front of the concat-MLP scorer (paper.py:339-348)
    const bool wide_mlp = (d->arch == LSTUR_ARCH_NGRU || d->arch == LSTUR_ARCH_IICGRU) && d->head == LSTUR_HEAD_POINT_MLP;
    if (d->arch != LSTUR_ARCH_IGRU && d->arch != LSTUR_ARCH_NIGRU && d->arch != LSTUR_ARCH_NIAVG &&
        d->arch != LSTUR_ARCH_NIATT && d->arch != LSTUR_ARCH_VO && !wide_mlp) {
      *why = "the pointwise heads support arch igru, vo, nigru, niavg, niatt (and ngru / iicgru with the MLP scorer)";
      return false;
    }
    if (!(c->gain > 0.f) || c->loss_negatives < 1) {
      *why = "pointwise loss needs gain > 0 and loss_negatives >= 1";
      return false;
    }
  }
  if (!(c->dropout >= 0.f && c->dropout < 1.f)) {
    *why = "dropout must be in [0, 1)";
    return false;
  }
  d->H = d->arch == LSTUR_ARCH_HGRU ? d->U / 2 : d->U;
  d->Ue = d->arch == LSTUR_ARCH_HGRU ? d->U / 2 : d->U;
  d->Uu = (d->arch == LSTUR_ARCH_NGRU || d->arch == LSTUR_ARCH_DGRU || d->arch == LSTUR_ARCH_IICGRU) ? 2 * d->U : d->U;   // user vector
  d->score = c->score_model;
  if (d->score < LSTUR_SCORE_DOT || d->score > LSTUR_SCORE_DNN) {
    *why = "unknown score_model";
    return false;
  }
  if (d->score != LSTUR_SCORE_DOT && d->head != LSTUR_HEAD_SOFTMAX) {
    *why = "score_model ddot / dnn belong to the listwise head";
    return false;
  }
  if (d->head == LSTUR_HEAD_SOFTMAX && d->score == LSTUR_SCORE_DOT && d->Uu != d->U) {
    *why = "ngru / dgru user vectors are 2U wide: they need score_model ddot or dnn (paper.py:600-602)";
    return false;
  }
  if (d->H > 256) {
    *why = "GRU width must be <= 256";
    return false;
  }
  d->Tp = d->T + d->k - 1;
  d->KD = d->k * d->D;
  d->nt_max = d->B * (d->W + d->K1);
  {
    // every kernel addresses a step's token-level tensors through 32-bit buffer offsets (descriptor size and per-lane
    // offset both wrap at 4 GiB): refuse such batches instead of dropping stores silently
    const size_t nt = (size_t)d->nt_max, lim = (size_t)1 << 32;
    const size_t emb = (nt * d->Tp * (size_t)std::max(d->D, convx_dp(d->D) * 3 / 2) + 1024) * 4;   // fp32 rows or Ex (<= 6 B / element)
    if (emb >= lim || nt * d->T * d->F * 4 >= lim) {
      *why = "max_batch too large: a step's embeddings / conv outputs exceed 4 GiB (the kernels address them through 32-bit "
             "buffer offsets); split the batch";
      return false;
    }
  }
  int pred_rows = d->nt_max / (d->W + 1);
  d->n_cap = ceil16(std::max(d->B, pred_rows));   // GRU kernels process whole 16-sample row tiles
  d->rows_cap = d->B * 8;
  {   // the dense sweep of a small table hides under the recurrences (fused step); a large one does not -- it goes lazy
    const char* e = getenv("LSTUR_LAZY_ADAM");
    const bool has_table = d->arch != LSTUR_ARCH_NIGRU && d->arch != LSTUR_ARCH_NIAVG && d->arch != LSTUR_ARCH_NIATT;
    d->lazy = has_table && (e ? !strcmp(e, "1") : d->Nu >= kLazyMinRows);
  }
  {
    const bool cx = d->enc == LSTUR_ENC_CNNATT && convx_supported(d->D, d->k, d->F) && !getenv("LSTUR_CONV_F32");
    d->np = !cx ? 0 : (d->word_trainable || Opts::is("LSTUR_CONV_SCHEME", "bf16x3")) ? 3 : 2;
  }
  // the backward pass hands its deferred reductions to ONE launch: 4 (GRU) + 2 (Dense) + 1 (conv_b / attention) + kernel (one
  // per tap) + 4 (batch statistics) + 1 (mhsa) + 9 (data-parallel staging of the user rows, lstur_dp_attach) jobs
  if (21 + d->k > kMaxReduceJobs) {
    *why = "kernel (conv width) too large for the step's single reduction launch";
    return false;
  }
  return true;
}

std::vector<ParamSlot> param_layout(const Dims& d, size_t* total) {
  std::vector<ParamSlot> v;
  size_t off = 0;
  auto add = [&](const char* n, size_t cnt) {
    v.push_back({n, off, cnt});
    off += (cnt + kParamAlign - 1) / kParamAlign * kParamAlign;
  };
  if (d.enc == LSTUR_ENC_MHSA) {
    add("qkv_w", (size_t)d.D * 3 * d.F);
    add("qkv_b", (size_t)3 * d.F);
  } else {
    add("conv_w", (size_t)d.KD * d.F);
    add("conv_b", d.F);
  }
  add("att_w", d.F);
  add("att_b", 1);
  add("dense_w", (size_t)d.F * d.U);
  add("dense_b", d.U);
  add("gru_w", (size_t)d.U * 3 * d.H);
  add("gru_u", (size_t)d.H * 3 * d.H);
  add("gru_b", 3 * d.H);
  if (d.arch == LSTUR_ARCH_GRU || d.arch == LSTUR_ARCH_IIGRU) {
    add("con_w", (size_t)(d.H + d.Ue) * d.U);
    add("con_b", d.U);
  }
  if (d.arch == LSTUR_ARCH_NIATT) {
    add("uatt_w", d.U);
    add("uatt_b", 1);
  }
  if (d.score == LSTUR_SCORE_DDOT) {
    add("sdu_w", (size_t)d.Uu * d.U);
    add("sdu_b", d.U);
    add("sdd_w", (size_t)d.U * d.U);
    add("sdd_b", d.U);
  } else if (d.score == LSTUR_SCORE_DNN) {
    add("sn1_w", (size_t)(d.Uu + d.U) * d.U);
    add("sn1_b", d.U);
    add("sn2_w", d.U);
    add("sn2_b", 1);
  }
  if (d.head == LSTUR_HEAD_POINT_MLP) {
    add("score_w1", (size_t)(d.Uu + d.U) * d.U);   // [u | d] -> U; u is 2U wide for the pointwise `gru` (paper.py:339-342)
    add("score_b1", d.U);
    add("score_w2", d.U);
    add("score_b2", 1);
  }
  *total = off;
  return v;
}

// Workspace carving: identical in lstur_query_sizes and lstur_create.
struct Workspace {
  int32_t* ids; float* emb; float* conv; float* dconv; float* att_a; float* att_wt; float* pooled; float* dpooled;
  float* docvec; float* ddocvec; uint8_t* hmask; uint8_t* gmask; float* uvec; float* xw; float* gru_tape;
  float* dgates; float* rh; float* hprev; float* dh0; float* hlast; float* dhlast;
  float* cat; float* dcat; float* uservec; float* duser; float* du; float* logits; float* probs; float* loss_i;
  float* correct_i; float* parts; float* conv_wp; float* gru_up;
  void* wgrad_scratch; void* gemm_scratch; size_t gemm_scratch_bytes; void* gemm_scratch_k[4]; float* colsum_scratch_k[3]; int32_t* first; float* gsum; int32_t* slot;
  float* scat; float* spre; float* sds; float* sdhid; float* sdcat;
  int32_t* lazy_last; float* lazy_lrt; int32_t* lazy_first; int32_t* dp_rows; int32_t* pend_rows; float* dp_grads; float* dp_grads2; float* dgrad_scratch; float* lw_a; float* lw_b; float* lw_c; float* lw_d; float* lw_s; float* lw_ds; float* lw_ones; float* ua_a; float* ua_w; float* ua_parts; float* ua_ds; float* uvec2; float* duser2; float* gsum2; unsigned short* conv_w3; unsigned short* e3; float* cs; float* gparts; void* fwd2_scratch; unsigned short* qkv; unsigned short* dqkv; unsigned short* qkv_w16; void* mh_scratch;
  int32_t* user_rows; float* user_rows_f; float* scores; float* colsum_scratch; int32_t* vidx; int32_t* vcount; int32_t* vscratch; float* att_ds; int32_t* offA; int32_t* offB;
  // the staged form of a batch (token ids, their compacted list + offsets, the gathered embeddings) exists twice: while a
  // step runs on one set, the NEXT batch can be staged into the other (lstur_prefetch_ids); ids / e3 / vidx / ... above
  // always point at the set of the running step
  uint32_t* live;   // per title: bit t = token t is not a pad (written with the token compaction, read by the attention kernels)
  uint32_t* kbits;  // conv-output dropout keep bits of a PREFETCHED batch (attention kernels' thread layout)
  bool kbits_valid = false;
  struct StageSet { int32_t* ids; unsigned short* e3; int32_t* vidx; int32_t* vcount; int32_t* vscratch; int32_t* offA; int32_t* offB; uint32_t* live; uint32_t* kbits; } set[2];
  size_t bytes;
};

size_t max_gemm_scratch(const Dims& d) {
  int nW = d.n_cap * d.W, nt = d.nt_max;
  size_t m = 0;
  auto upd = [&](int M, int N, int K) { m = std::max(m, gemm_scratch_bytes(M, N, K)); };
  upd(nt, d.U, d.F);            // dense fwd
  upd(nW, 3 * d.H, d.U);        // gru input projection
  upd(d.F, d.U, nt);            // ddense_w
  upd(nt, d.F, d.U);            // dpooled
  upd(d.U, 3 * d.H, nW);        // dgru_w
  upd(d.H, 2 * d.H, nW);        // dgru_u zr
  upd(d.H, d.H, nW);            // dgru_u h
  upd(nW, d.U, 3 * d.H);        // dx
  upd(d.n_cap, d.U, d.H + d.Ue);
  upd(d.H + d.Ue, d.U, d.n_cap);
  upd(d.n_cap, d.H + d.Ue, d.U);
  if (d.score != LSTUR_SCORE_DOT) {   // listwise ddot / dnn score models
    const int rows = d.n_cap * d.K1, wc = d.Uu + d.U;
    upd(rows, d.U, wc);
    upd(wc, d.U, rows);
    upd(rows, wc, d.U);
    upd(d.n_cap, d.U, d.Uu);
    upd(d.Uu, d.U, d.n_cap);
    upd(d.n_cap, d.Uu, d.U);
    upd(d.U, d.U, rows);
  }
  upd(d.n_cap, d.U, d.Uu + d.U);   // pointwise MLP head
  upd(d.Uu + d.U, d.U, d.n_cap);
  upd(d.n_cap, d.Uu + d.U, d.U);
  // split-K slab bound for any smaller runtime batch: (256/tiles + 1) * tiles * 64*64 floats < 8 MiB
  return std::max(m, (size_t)8 << 20);
}

void carve(const Dims& d, void* base, Workspace* w) {
  size_t off = 0;
  auto take = [&](size_t bytes) -> void* {
    void* p = base ? (void*)((char*)base + off) : nullptr;
    off += (bytes + 255) / 256 * 256;
    return p;
  };
  const size_t nt = d.nt_max, nc = d.n_cap, nW = (size_t)d.n_cap * d.W, f4 = sizeof(float);
  w->ids = (int32_t*)take(nt * d.T * 4);
  w->emb = (float*)take((nt * d.Tp * d.D + kEmbSlack) * f4);
  w->conv = (float*)take(nt * d.T * d.F * f4);
  w->dconv = (float*)take(nt * d.T * d.F * f4);
  w->att_a = (float*)take(nt * d.T * f4);
  w->att_wt = (float*)take(nt * d.T * f4);
  w->pooled = (float*)take(nt * d.F * f4);
  w->dpooled = (float*)take(nt * d.F * f4);
  w->docvec = (float*)take(nt * d.U * f4);
  w->ddocvec = (float*)take(nt * d.U * f4);
  w->hmask = (uint8_t*)take(nW);
  w->gmask = (uint8_t*)take(nW);
  w->uvec = (float*)take(nc * d.Ue * f4);
  w->xw = (float*)take(nW * 3 * d.H * f4);
  w->gru_tape = (float*)take(gru_tape_floats((int)nc, d.W, d.H) * f4);
  w->dgates = (float*)take(nW * 3 * d.H * f4);
  w->rh = (float*)take(nW * d.H * f4);
  w->hprev = (float*)take(nW * d.H * f4);
  w->dh0 = (float*)take(nc * d.H * f4);
  w->hlast = (float*)take(nc * d.H * f4);
  w->dhlast = (float*)take(nc * d.H * f4);
  w->cat = (float*)take(nc * (d.H + d.Ue) * f4);
  w->dcat = (float*)take(nc * (d.H + d.Ue) * f4);
  w->uservec = (float*)take(nc * d.Uu * f4);
  w->duser = (float*)take(nc * d.Ue * f4);
  w->du = (float*)take(nc * d.Uu * f4);
  w->logits = (float*)take(nc * d.K1 * f4);
  w->probs = (float*)take(nc * d.K1 * f4);
  w->loss_i = (float*)take(nc * f4);
  w->correct_i = (float*)take(nc * f4);
  w->parts = (float*)take(nt * (3 * ((d.F + kParamAlign - 1) / kParamAlign * kParamAlign) + 64) * f4);
  w->conv_wp = (float*)take(pack_b_floats(d.KD, d.F) * f4);
  w->gru_up = (float*)take(gru_pack_floats(d.H) * f4);
  w->wgrad_scratch = take(std::max(conv_wgrad_scratch_bytes(d.D, d.k, d.F),
                                   d.np ? convx_wgrad_scratch_bytes((int)nt, d.T, d.D, d.k, d.F) : 0));
  w->gemm_scratch_bytes = max_gemm_scratch(d);
  w->gemm_scratch = take(w->gemm_scratch_bytes);
  // weight-gradient GEMMs of the backward pass keep their split-K slabs until the step's single reduction launch
  auto tn_bytes = [](int M, int N, int K) { return std::max(std::max(gemm_scratch_bytes(M, N, K), gemm3_tn_scratch_bytes(M, N, K)), (size_t)4 << 20); };
  w->gemm_scratch_k[0] = take(tn_bytes(d.U, 3 * d.H, (int)nW));
  w->gemm_scratch_k[1] = take(tn_bytes(d.H, 2 * d.H, (int)nW));
  w->gemm_scratch_k[2] = take(tn_bytes(d.H, d.H, (int)nW));
  w->gemm_scratch_k[3] = take(tn_bytes(d.F, d.U, (int)nt));
  w->first = (int32_t*)take((size_t)d.rows_cap * 4);
  w->gsum = (float*)take((size_t)d.rows_cap * d.Ue * f4);
  w->slot = (int32_t*)take((size_t)d.Nu * 4);
  w->lazy_last = d.lazy ? (int32_t*)take((size_t)d.Nu * 4) : nullptr;
  w->lazy_lrt = d.lazy ? (float*)take((size_t)kLazyLog * f4) : nullptr;   // (LSTUR_LAZY_LOG uses a prefix of it)
  w->lazy_first = d.lazy ? (int32_t*)take((size_t)std::max(d.n_cap, d.rows_cap) * 4) : nullptr;
  w->scat = (float*)take(nc * (d.Uu + d.U) * f4);
  w->spre = (float*)take(nc * d.U * f4);
  w->sds = (float*)take(nc * f4);
  w->sdhid = (float*)take(nc * d.U * f4);
  w->sdcat = (float*)take(nc * (d.Uu + d.U) * f4);
  w->dgrad_scratch = d.word_trainable ? (float*)take(conv_dgrad_scratch_floats((int)nt, d.T, d.D, d.k, d.F) * f4) : nullptr;
  const bool lw = d.score != LSTUR_SCORE_DOT;   // scratch of the listwise ddot / dnn score models
  const size_t lrows = nc * d.K1;
  w->lw_a = lw ? (float*)take(lrows * (d.Uu + d.U) * f4) : nullptr;
  w->lw_b = lw ? (float*)take(lrows * (d.Uu + d.U) * f4) : nullptr;
  w->lw_c = lw ? (float*)take(lrows * d.U * f4) : nullptr;
  w->lw_d = lw ? (float*)take(lrows * d.U * f4) : nullptr;
  w->lw_s = lw ? (float*)take(lrows * f4) : nullptr;
  w->lw_ds = lw ? (float*)take(lrows * f4) : nullptr;
  w->lw_ones = lw ? (float*)take(nc * f4) : nullptr;
  const bool ua = d.arch == LSTUR_ARCH_NIATT;   // attention over the history: saved tanh / weights, gradient partials
  w->ua_a = ua ? (float*)take(nW * f4) : nullptr;
  w->ua_w = ua ? (float*)take(nW * f4) : nullptr;
  w->ua_parts = ua ? (float*)take(nc * (size_t)(2 * d.U + 1) * f4) : nullptr;
  w->ua_ds = ua ? (float*)take(nW * f4) : nullptr;
  const bool ii = two_tables(d);   // second user table: gathered rows, their gradients, combined sums
  w->uvec2 = ii ? (float*)take(nc * d.Ue * f4) : nullptr;
  w->duser2 = ii ? (float*)take(nc * d.Ue * f4) : nullptr;
  w->gsum2 = ii ? (float*)take((size_t)d.rows_cap * d.Ue * f4) : nullptr;
  const bool mh = d.enc == LSTUR_ENC_MHSA;
  // split-operand conv path (convx.hip): split weight image Wx, pre-split embeddings Ex, the scale block, slabs of the
  // k-sliced tail tiles
  const bool c3 = d.np != 0;
  w->conv_w3 = c3 ? (unsigned short*)take(convx_w_elems(d.D, d.k, d.F, d.np) * 2) : nullptr;
  w->e3 = c3 ? (unsigned short*)take(convx_e_elems((int)nt, d.T, d.D, d.k, d.np) * 2) : nullptr;
  w->cs = c3 ? (float*)take(kCsWords * sizeof(float)) : nullptr;
  w->gparts = c3 ? (float*)take(nt * f4) : nullptr;   // max |dC| per title, left by the attention backward (f16x2 kernel gradient)
  w->fwd2_scratch = c3 ? take(convx_fwd_scratch_bytes()) : nullptr;
  w->qkv = mh ? (unsigned short*)take(nt * d.T * 3 * d.F * 2) : nullptr;     // bf16 Q | K | V rows
  w->dqkv = mh ? (unsigned short*)take(nt * d.T * 3 * d.F * 2) : nullptr;
  w->qkv_w16 = mh ? (unsigned short*)take((size_t)d.D * 3 * d.F * 2) : nullptr;   // the projection weight rounded to bf16, once per step
  w->mh_scratch = mh ? take(proj_wgrad_scratch_bytes(d.D, 3 * d.F, (int)(nt * d.T))) : nullptr;
  w->dp_rows = (int32_t*)take((size_t)d.rows_cap * 4);
  w->pend_rows = (int32_t*)take((size_t)d.rows_cap * 4);   // rows of a deferred table sweep (marked until it has run)
  w->dp_grads = (float*)take((size_t)d.rows_cap * d.Ue * f4);
  w->dp_grads2 = ii ? (float*)take((size_t)d.rows_cap * d.Ue * f4) : nullptr;
  w->user_rows = (int32_t*)take(nc * 8);   // + the same numbers as floats behind them (data-parallel staging)
  w->user_rows_f = w->user_rows ? (float*)(w->user_rows + nc) : nullptr;
  w->scores = (float*)take(nc * f4);
  w->colsum_scratch = (float*)take((size_t)kColsumSlices * std::max(std::max(3 * d.H, d.U), (int)(3 * ((d.F + kParamAlign - 1) / kParamAlign * kParamAlign) + 64)) * f4);
  for (int k = 0; k < 3; ++k)
    w->colsum_scratch_k[k] = (float*)take((size_t)kColsumSlices * std::max(std::max(3 * d.H, d.U), (int)(3 * ((d.F + kParamAlign - 1) / kParamAlign * kParamAlign) + 64)) * f4);
  w->vidx = (int32_t*)take(nt * d.T * 4);
  w->vcount = (int32_t*)take(256);
  w->att_ds = (float*)take(nt * d.T * f4);
  w->offA = (int32_t*)take(nt * d.T * 4);
  w->offB = (int32_t*)take(nt * d.T * 4);
  w->vscratch = (int32_t*)take(compact_scratch_ints((int)(nt * d.T)) * 4);
  w->live = (uint32_t*)take(nt * 4);
  w->kbits = nullptr;
  w->set[0] = Workspace::StageSet{w->ids, w->e3, w->vidx, w->vcount, w->vscratch, w->offA, w->offB, w->live, nullptr};
  w->set[1] = w->set[0];
  if (c3 && d.head == LSTUR_HEAD_SOFTMAX && !d.word_trainable) {   // second staging set (prefetch of the next batch)
    w->set[1].ids = (int32_t*)take(nt * d.T * 4);
    w->set[1].e3 = (unsigned short*)take(convx_e_elems((int)nt, d.T, d.D, d.k, d.np) * 2);
    w->set[1].vidx = (int32_t*)take(nt * d.T * 4);
    w->set[1].vcount = (int32_t*)take(256);
    w->set[1].offA = (int32_t*)take(nt * d.T * 4);
    w->set[1].offB = (int32_t*)take(nt * d.T * 4);
    w->set[1].vscratch = (int32_t*)take(compact_scratch_ints((int)(nt * d.T)) * 4);
    w->set[1].live = (uint32_t*)take(nt * 4);
    w->set[0].kbits = (uint32_t*)take(attpool_keep_bits_words((int)nt) * 4);   // (only prefetched batches carry them)
    w->set[1].kbits = (uint32_t*)take(attpool_keep_bits_words((int)nt) * 4);
  }
  w->bytes = off;
}

}  // namespace

struct ProfKernel {
  const char* name;
  std::vector<hipEvent_t> start, stop;
  size_t used = 0;
};

struct lstur_ctx {
  Opts opt;
  // the listwise scorer as the tail of the forward recurrence's kernel (run_forward / run_listwise_score): asked for by the
  // training / evaluation entry point for this call, done = the recurrence took it
  struct { bool want = false, backward = false, done = false; float inv_n = 0.f; } sfuse;
  float* dp_tail = nullptr;   // lstur_dp_attach: the backward pass stages this rank's user rows here
  int dp_rank = 0, dp_world = 1;
  bool prof_on = false;
  bool prof_roofline_only = false;   // lstur_profile(ctx, 2): brackets around gather / conv_fwd / conv_wgrad only
  std::deque<ProfKernel> prof;  // deque: nested ProfScopes hold pointers across push_back
  ProfKernel* prof_slot(const char* name) {
    for (auto& k : prof)
      if (!strcmp(k.name, name)) return &k;
    prof.push_back(ProfKernel());
    prof.back().name = name;
    return &prof.back();
  }
  lstur_cfg cfg;
  Dims d;
  lstur_buffers buf;
  Workspace ws;
  // weight-gradient side stream (run_backward): the GRU / Dense parameter-gradient GEMMs only feed the step's final
  // reduction, so they run beside the chain d(doc vectors) -> attention backward -> conv kernel gradient
  hipStream_t side_w = nullptr;
  hipEvent_t ev_w1 = nullptr, ev_w2 = nullptr, ev_w3 = nullptr;
  hipEvent_t ev_c0 = nullptr, ev_c1 = nullptr;   // candidate titles' encoder backward: fork behind the scorer, done
  bool cand_forked = false;                       // ev_c0 rides on this step's scorer launch
  // data-parallel step: the zero-gradient half of the dense table sweep of iteration t, deferred to the next forward
  struct Deferred {
    bool valid = false;      // rows [pend_rows, n_rows) are updated and MARKED; every other row still owes iteration t
    bool inflight = false;   // the sweep runs on the background stream: ev_t1 tells when the marks are gone
    float lr_t = 0.f;
    int n_rows = 0;
  } defer;
  // staging prefetch (lstur_prefetch_ids): the next batch's title expansion, compaction and embedding gather run on a
  // side stream under the current step's forward recurrence (HBM and 240 CUs idle there)
  struct Pending { const int32_t* hist = nullptr; const int32_t* cands = nullptr; const int32_t* users = nullptr; int n = 0;
                   uint32_t step = 0;
                   bool valid = false, launched = false, kbits = false; } pend, ready;   // announced | staged in the other set
  const int32_t* next_users = nullptr;  // lstur_prefetch_rows: the user ids of the batch the next lstur_prefetch_ids announces
  int prefetch_hits = 0;
  int cur_set = 0;
  bool staged_hit = false;              // this step's staging was prefetched: run_forward skips it
  hipStream_t side_p = nullptr;
  hipEvent_t ev_p0 = nullptr, ev_p1 = nullptr;
  // fused training step (lstur_train_step*): the optimizer is part of the call, so the library schedules it -- the dense
  // sweep of the user table(s) needs only the recurrence's row gradients and runs on a stream of its own under the news
  // encoder's backward (HBM-bound beside MFMA-bound kernels); the weight images of the NEXT step are packed behind the
  // dense Adam instead of in front of the next conv forward
  struct Fused { bool on = false, swept = false, table_done = false; float lr_t = 0.f; int t = 0; } fused;
  hipStream_t side_t = nullptr;
  hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;
  hipEvent_t ev_f0 = nullptr;           // the forward's one fork point (staging prefetch + table sweep)
  bool images_valid = false;            // conv / GRU weight images match dense_param (packed behind the last Adam)
  bool escale_valid = false;            // f16x2: the embedding scale matches the word table (lstur_params_changed clears it)
  int pack_seq = 0;                     // f16x2: packs of the conv weight image so far (rotating maxima, convx.hip)
  int gru_seq = 0;                      // ... and of the recurrent matrix's fp16-pair images (gru4.hip)
  int lazy_t = 0;                       // lazy Adam: iterations applied so far (rows are current for last[row] <= lazy_t)
  int lazy_flushed_t = 0;               // ... and the iteration of the last full flush (the lr_t log is a ring)
  std::vector<ParamSlot> layout;
  size_t dense_total;
  const int32_t* title_table = nullptr;
  int32_t n_docs = 0;
  bool lazy_init_done = false;
  int last_n = 0, last_k1 = 0;
  bool last_usex = false;   // the last forward ran the split-operand conv path (Ex embeddings; the fp32 `emb` layout was not written)
  bool counted = false;   // stage_titles already left the per-block non-pad counts in ws.vscratch
  float* P(const char* name) const {
    for (auto& s : layout)
      if (!strcmp(s.name, name)) return buf.dense_param + s.offset;
    return nullptr;
  }
  float* G(const char* name) const {
    for (auto& s : layout)
      if (!strcmp(s.name, name)) return buf.dense_grad + s.offset;
    return nullptr;
  }
};

thread_local hipEvent_t lstur::g_stop_event = nullptr;
thread_local hipEvent_t lstur::g_start_event = nullptr;
thread_local hipEvent_t lstur::g_stop_pending = nullptr;

namespace {

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(LSTUR_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
  return LSTUR_OK;
}

// HIP-event bracket around one kernel family on the launch stream (only when profiling is enabled)
// attach = true: the launcher below this scope goes through launch_k / last_launch (convx forward + fix-up, kernel
// gradient, gather): the two events ride the family's first and last kernel instead of being recorded around them
struct ProfScope {
  lstur_ctx* c;
  ProfKernel* k = nullptr;
  hipStream_t s;
  bool attached = false;
  ProfScope(lstur_ctx* ctx, const char* name, hipStream_t st, bool attach = false) : c(ctx), s(st) {
    if (!c->prof_on) return;
    if (c->prof_roofline_only && strcmp(name, "gather") && strcmp(name, "conv_fwd") && strcmp(name, "conv_wgrad") &&
        strcmp(name, "conv_wgrad_side") && strcmp(name, "qkv_proj") && strcmp(name, "qkv_wgrad"))
      return;
    k = c->prof_slot(name);
    if (k->used == k->start.size()) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
        k = nullptr;
        return;
      }
      k->start.push_back(a);
      k->stop.push_back(b);
    }
    if (attach && c->opt.prof_attach) {
      arm_bracket(k->start[k->used], k->stop[k->used]);
      attached = true;
    } else {
      (void)hipEventRecord(k->start[k->used], s);
    }
  }
  ~ProfScope() {
    if (k) {
      if (!attached) (void)hipEventRecord(k->stop[k->used], s);
      else if (g_start_event || g_stop_pending || g_stop_event) {   // nothing was launched (empty batch): record plainly
        arm_bracket(nullptr, nullptr);
        g_stop_event = nullptr;
        (void)hipEventRecord(k->start[k->used], s);
        (void)hipEventRecord(k->stop[k->used], s);
      }
      k->used++;
    }
  }
};
#define PROF(name, stmt)            \
  do {                              \
    ProfScope _ps(c, name, s);      \
    stmt;                           \
  } while (0)

// Side streams are shared by ALL contexts of the process on a device and live as long as the process: a stream is a queue,
// the hardware has few of them, and a process that held two contexts (bench.py: the headline model + the end-to-end
// model; the test suite) ran every kernel ~30 % slower once their private streams added up to more than the hardware
// queues.  kind: 0 parameter gradients (high priority), 1 background (staging prefetch, table sweep).  Ordering between
// contexts that share a stream is plain stream order (each context still forks / joins through its own events).
hipStream_t pooled_stream(int kind, bool high_priority) {
  struct Pool { hipStream_t s[2] = {nullptr, nullptr}; };
  static std::mutex mu;
  static std::map<int, Pool> pools;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  Pool& p = pools[dev];
  if (!p.s[kind]) {
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (hipStreamCreateWithPriority(&p.s[kind], hipStreamNonBlocking, high_priority ? hi : lo) != hipSuccess) p.s[kind] = nullptr;
  }
  return p.s[kind];
}

// the side stream of the parameter-gradient GEMMs (nullptr: keep them on the caller's stream)
hipStream_t wgrad_stream(lstur_ctx* c) {
  if (!c->opt.wg_stream) return nullptr;
  if (!c->side_w) {
    // highest priority: its thin kernels are to slip in as soon as a workgroup of the main stream's kernel retires -- at
    // equal priority they waited behind the attention backward's 3520 workgroups and spilled into the conv kernel gradient,
    // which shares a CU with nothing (it then ran ~10 % slower and the step's reduction waited for the stragglers)
    if (!(c->side_w = pooled_stream(0, true)) ||
        hipEventCreateWithFlags(&c->ev_w1, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_w2, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_w3, hipEventDisableTiming) != hipSuccess) {
      c->side_w = nullptr;
      return nullptr;
    }
  }
  return c->side_w;
}

void use_set(lstur_ctx* c, int k) {
  Workspace& w = c->ws;
  c->cur_set = k;
  w.ids = w.set[k].ids; w.e3 = w.set[k].e3; w.vidx = w.set[k].vidx; w.vcount = w.set[k].vcount;
  w.vscratch = w.set[k].vscratch; w.offA = w.set[k].offA; w.offB = w.set[k].offB; w.live = w.set[k].live;
  w.kbits = w.set[k].kbits;
}

void lazy_catchup_rows(lstur_ctx* c, const int32_t* rows, int n, hipStream_t s);
hipStream_t wgrad_stream(lstur_ctx* c);
// the stream a lazily updated table is worked on outside the caller's stream: the parameter-gradient stream (idle under
// the recurrences), else the background stream
hipStream_t lazy_stream(lstur_ctx* c) {
  hipStream_t w = wgrad_stream(c);
  return w ? w : c->side_p;
}
// enqueue the pending prefetch (if any) on the side stream, behind everything `s` holds so far; true: something was launched
// ONE background stream for everything that hides under the recurrences: the next batch's staging, then the zero-gradient
// half of the user-table sweep, later the batch's own table rows (side_p and side_t are the same queue).  Streams are not
// free on this stack: a FIFTH one in use (measured: a second parameter-gradient stream) made every kernel of the step
// ~45 % slower -- beyond the hardware queues streams get multiplexed -- and RCCL brings its own in data-parallel runs.
bool background_stream(lstur_ctx* c) {
  if (c->side_p) return true;
  if (!(c->side_p = pooled_stream(1, false)) ||
      hipEventCreateWithFlags(&c->ev_p0, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_p1, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_t0, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_t1, hipEventDisableTiming) != hipSuccess) {
    c->side_p = c->side_t = nullptr;
    return false;
  }
  c->side_t = c->side_p;   // staging and table sweep share ONE background queue
  return true;
}
bool fork_events(lstur_ctx* c) {
  if (c->ev_f0) return true;
  if (hipEventCreateWithFlags(&c->ev_f0, hipEventDisableTiming) != hipSuccess) c->ev_f0 = nullptr;
  return c->ev_f0 != nullptr;
}
bool prefetch_wanted(const lstur_ctx* c) {
  const Workspace& w = c->ws;
  return c->pend.valid && !c->ready.valid && w.set[1].ids != w.set[0].ids;
}
// `forked`: recorded on the caller's stream behind everything the staging must wait for
bool launch_prefetch(lstur_ctx* c, hipEvent_t forked) {
  auto& q = c->pend;
  const Dims& d = c->d;
  Workspace& w = c->ws;
  if (!prefetch_wanted(c)) return false;
  if (!background_stream(c)) {
    q.valid = false;
    return false;
  }
  const Workspace::StageSet& t = w.set[c->cur_set ^ 1];
  hipStream_t p = c->side_p;
  (void)hipStreamWaitEvent(p, forked, 0);
  const int nW = q.n * d.W, nt = nW + q.n * d.K1, pl = (d.k - 1) / 2;
  launch_expand_count(c->title_table, q.hist, nW, q.cands, q.n * d.K1, d.T, t.ids, t.vscratch, p);
  launch_compact_write_offsets(t.ids, nt * d.T, t.vscratch, t.vidx, t.vcount, d.T, d.Tp, convx_dp(d.D) / 8 * d.np, d.F, t.offA,
                               t.offB, p, t.live, nW * d.T);
  {
    ProfScope _ps(c, "gather", p, true);
    launch_gatherx(c->buf.word_emb, t.ids, t.e3, nt, d.T, d.D, pl, d.k - 1 - pl, c->cfg.dropout,
                   dropout_key(c->cfg.seed, q.step, 0), true, d.np, w.cs, p, c->opt.prefetch_blocks, t.live);
  }
  // the attention passes' dropout decisions for that batch (stream 1 of its step), behind the gather
  q.kbits = launch_attpool_keep_bits(t.live, t.kbits, nt, d.T, d.F, c->cfg.dropout, dropout_key(c->cfg.seed, q.step, 1), p);
  // lazily updated table: the rows that batch will read take their pending zero-gradient updates here, under the
  // recurrence (a row that waited 2 000 iterations is a 2 000-long chain per element: latency-bound work for idle CUs);
  // the forward's own catch-up then finds one iteration left
  // -- on the parameter-gradient stream, idle until BPTT is through and the one the batch's own rows are stepped on (a row
  // of this batch AND the next must take its catch-up and its step in order), beside the gather on this one
  if (d.lazy && q.users) {
    hipStream_t ls = lazy_stream(c);
    if (ls != p) (void)hipStreamWaitEvent(ls, forked, 0);
    lazy_catchup_rows(c, q.users, q.n, ls);
  }
  (void)hipEventRecord(c->ev_p1, p);
  q.launched = true;
  c->ready = q;                      // staged: the next training call may pick it up
  q = lstur_ctx::Pending();
  return true;
}

// C = A . op(B) + bias for the dense layers on the critical path: the bf16x3 kernel where it applies, else the fp32 MFMA one
void dense_gemm(const float* A, int lda, const float* B, int ldb, bool b_nk, float* C, int ldc, const float* bias, int M, int N,
                int K, void* scratch, hipStream_t s) {
  if (gemm3_supported(lda, ldb, M, N, K)) launch_gemm3(A, lda, B, ldb, b_nk, C, ldc, bias, M, N, K, s);
  else launch_gemm(A, lda, 0, B, ldb, b_nk ? 1 : 0, C, ldc, bias, M, N, K, 1.f, scratch, s);
}

thread_local bool g_wgrad_bf16x3 = false;   // (set from the context's options by run_backward)
// G[M,N] = A^T . B (A [K][M], B [K][N]) as a deferred reduction: the parameter gradients behind the recurrence / Dense(U)
void wgrad_gemm(const float* A, int lda, const float* B, int ldb, float* G, int ldg, int M, int N, int K, void* scratch,
                hipStream_t s, ReduceJob* job) {
  // These run on the side stream BESIDE the attention backward and have to fit into what its workgroups leave of a SIMD's
  // registers: the fp32 MFMA kernel (68 VGPRs, 20 KB LDS) does; the bf16x3 kernel with eight k-steps in flight (188
  // VGPRs, 72 KB) waited for whole CUs to drain and ran into the conv kernel gradient (59 us instead of 41 for dgru_u, the
  // last product finished BEHIND the conv kernel gradient), and its thin form (two k-steps in flight, 92 VGPRs) still
  // measures 6-9 us per step behind the fp32 kernel (same box, 1.143 / 1.147 against 1.138 / 1.138 ms).
  // LSTUR_WGRAD_BF16X3=1 selects it for A/B runs.
  if (g_wgrad_bf16x3 && gemm3_tn_supported(lda, ldb, M, N, K)) launch_gemm3_tn(A, lda, B, ldb, G, ldg, M, N, K, 1.f, scratch, s, job);
  else launch_gemm(A, lda, 1, B, ldb, 0, G, ldg, nullptr, M, N, K, 1.f, scratch, s, job);
}

// operand images of the weights the MFMA kernels read (conv: split 16-bit parts or the fp32 pack; GRU: resident strips).
// `fresh`: the caller changed the weights (or this is the first pack): an f16x2 image takes its scale from the weights as
// they are; otherwise from the maximum the previous pack saw, like the dense Adam launch that writes the image in the step
// most one Keras-Adam step moves a parameter, in units of lr_t: |m_t| / sqrt(v_t) <= (1 - b1) / sqrt(1 - b2) / (1 - b1 / sqrt(b2))
// (31.8 at .9 / .999); infinity when the series does not converge (b1 >= sqrt(b2)): packs are then always fresh
float adam_growth_bound(const lstur_cfg& cf) {
  const float q = cf.beta1 / sqrtf(cf.beta2);
  if (!(cf.beta1 >= 0.f && cf.beta2 > 0.f && cf.beta2 < 1.f && q < 1.f)) return INFINITY;
  return (1.0f - cf.beta1) / sqrtf(1.0f - cf.beta2) / (1.0f - q);
}
void pack_weight_images(lstur_ctx* c, hipStream_t sp, bool fresh, float lr_t = 0.f) {
  const Dims& d = c->d;
  Workspace& w = c->ws;
  const bool mhsa = d.enc == LSTUR_ENC_MHSA;
  const float grow = fresh ? 0.f : adam_growth_bound(c->cfg) * fabsf(lr_t);
  if (d.np) launch_convx_pack_w(c->P("conv_w"), d.D, d.k, d.F, w.conv_w3, d.np, w.cs, c->pack_seq++, fresh || !isfinite(grow), grow, sp);
  else if (!mhsa) launch_pack_b(c->P("conv_w"), d.F, d.KD, d.F, w.conv_wp, sp);
  else launch_round_bf16(c->P("qkv_w"), w.qkv_w16, (size_t)d.D * 3 * d.F, sp);   // (D % 4 == 0: mhsa_supported)
  launch_gru_pack(c->P("gru_u"), d.H, w.gru_up, sp, c->opt.gru_f16, c->gru_seq++);   // (a stand-alone pack is always fresh)
}

bool has_user_table(const Dims& d) {
  return d.arch != LSTUR_ARCH_NIGRU && d.arch != LSTUR_ARCH_NIAVG && d.arch != LSTUR_ARCH_NIATT;
}

// ---- lazy dense Adam (large tables; kernels and rationale in misc.hip) ----
void lazy_init(lstur_ctx* c, hipStream_t s);
void lazy_flush_all(lstur_ctx* c, hipStream_t s) {   // every row current for iteration lazy_t
  const Dims& d = c->d;
  const lstur_cfg& cf = c->cfg;
  if (!d.lazy || c->lazy_flushed_t == c->lazy_t) return;
  lazy_init(c, s);
  launch_lazy_flush(c->buf.user_emb, c->buf.user_m, c->buf.user_v, c->ws.lazy_last, c->ws.lazy_lrt, c->lazy_t, d.Nu, d.Ue,
                    cf.beta1, cf.beta2, cf.eps, c->opt.lazy_log - 1, s);
  if (two_tables(d))
    launch_lazy_flush(c->buf.user_emb2, c->buf.user_m2, c->buf.user_v2, c->ws.lazy_last, c->ws.lazy_lrt, c->lazy_t, d.Nu, d.Ue,
                      cf.beta1, cf.beta2, cf.eps, c->opt.lazy_log - 1, s);
  launch_fill_i32(c->ws.lazy_last, c->lazy_t, d.Nu, s);
  c->lazy_flushed_t = c->lazy_t;
}
// the caller is about to apply iteration t: iterations must come one after the other (a restart or a jump flushes and
// re-bases), and no row may fall further behind than the lr_t ring reaches
void lazy_begin_iteration(lstur_ctx* c, int t, hipStream_t s) {
  if (!c->d.lazy) return;
  lazy_init(c, s);
  if (t != c->lazy_t + 1) {
    lazy_flush_all(c, s);
    c->lazy_t = c->lazy_flushed_t = t - 1;
    launch_fill_i32(c->ws.lazy_last, t - 1, c->d.Nu, s);
  } else if (t - c->lazy_flushed_t >= c->opt.lazy_log - 2) {
    lazy_flush_all(c, s);
  }
}
// the rows a forward is about to read, current for every iteration applied so far
void lazy_catchup_rows(lstur_ctx* c, const int32_t* rows, int n, hipStream_t s) {
  const Dims& d = c->d;
  const lstur_cfg& cf = c->cfg;
  if (!d.lazy || c->lazy_t == c->lazy_flushed_t) return;
  const bool two = two_tables(d);
  launch_lazy_catchup(c->buf.user_emb, c->buf.user_m, c->buf.user_v, two ? c->buf.user_emb2 : nullptr,
                      two ? c->buf.user_m2 : nullptr, two ? c->buf.user_v2 : nullptr, rows, n, c->ws.lazy_first,
                      c->ws.lazy_last, c->ws.lazy_lrt, c->lazy_t, d.Ue, cf.beta1, cf.beta2, cf.eps, c->opt.lazy_log - 1, s);
}
// iteration t for the rows with a gradient (gsum / slot from the row combine; second table: combined here)
void lazy_rows_step(lstur_ctx* c, const int32_t* rows, const float* row_grads2, int n_rows, int t, float lr_t, hipStream_t s) {
  const Dims& d = c->d;
  const lstur_cfg& cf = c->cfg;
  launch_lazy_rows_adam(c->buf.user_emb, c->buf.user_m, c->buf.user_v, rows, c->ws.slot, c->ws.gsum, n_rows, d.Ue,
                        c->ws.lazy_last, c->ws.lazy_lrt, t, lr_t, cf.beta1, cf.beta2, cf.eps, c->opt.lazy_log - 1, s);
  if (two_tables(d)) {
    launch_user_grad_combine(rows, row_grads2 ? row_grads2 : c->ws.duser2, n_rows, d.Ue, c->ws.first, c->ws.gsum2, c->ws.slot, s);
    launch_lazy_rows_adam(c->buf.user_emb2, c->buf.user_m2, c->buf.user_v2, rows, c->ws.slot, c->ws.gsum2, n_rows, d.Ue,
                          c->ws.lazy_last, c->ws.lazy_lrt, t, lr_t, cf.beta1, cf.beta2, cf.eps, c->opt.lazy_log - 1, s);
  }
  launch_lazy_rows_mark(rows, c->ws.slot, n_rows, c->ws.lazy_last, c->ws.lazy_lrt, t, lr_t, c->opt.lazy_log - 1, s);
  launch_user_slot_clear(rows, n_rows, c->ws.slot, s);
}

// Keras-Adam sweep of the user table(s) for the rows / row gradients of this batch (duplicates combined in sample order)
void adam_user_tables(lstur_ctx* c, const int32_t* rows, const float* row_grads, int n_rows, float lr_t, bool combine_first,
                      hipStream_t s, const float* row_grads2 = nullptr) {
  const Dims& d = c->d;
  const lstur_cfg& cf = c->cfg;
  if (combine_first) launch_user_grad_combine(rows, row_grads, n_rows, d.Ue, c->ws.first, c->ws.gsum, c->ws.slot, s);
  {
    ProfScope _ps(c, "adam_user_table", s);
    launch_adam_table(c->buf.user_emb, c->buf.user_m, c->buf.user_v, c->ws.slot, c->ws.gsum, d.Nu, d.Ue, lr_t, cf.beta1,
                      cf.beta2, cf.eps, s);
  }
  if (two_tables(d)) {   // the second table: same rows (hence the same slot map), its own gradients
    launch_user_grad_combine(rows, row_grads2 ? row_grads2 : c->ws.duser2, n_rows, d.Ue, c->ws.first, c->ws.gsum2,
                             c->ws.slot, s);
    launch_adam_table(c->buf.user_emb2, c->buf.user_m2, c->buf.user_v2, c->ws.slot, c->ws.gsum2, d.Nu, d.Ue, lr_t,
                      cf.beta1, cf.beta2, cf.eps, s);
  }
  launch_user_slot_clear(rows, n_rows, c->ws.slot, s);
}

// Fused step: the dense sweep of the user table(s) in two halves.  A row outside the batch takes the zero-gradient Adam
// update, which needs nothing from this step but lr_t: that half (all but <= n rows; 3 reads + 3 writes per element) runs
// on a stream of its own under the forward recurrence, where HBM and 240 CUs idle.  (Beside the conv kernels it finds no
// registers, beside the small GEMMs / attention backward it takes the bandwidth they are bound by: both were measured.)
// The batch's own rows are marked first (their table reads of this forward see the old values) and finished behind the
// recurrence's backward from their combined gradients.  Same arithmetic per element as the one-piece sweep.
bool table_stream(lstur_ctx* c) { return background_stream(c); }
// The recurrence beside it is a latency chain (its gate inputs and tape cross HBM every step): a sweep at full bandwidth
// stretched it by most of what the sweep took (176 -> 216 us at 50 k users), and at 256 workgroups by more.  So the sweep
// is launched THIN -- as many workgroups as finish it inside the two recurrences (~14 GB/s each at this load; 64 at 50 k
// users: 234 us, recurrence unchanged) -- unless that takes more than 128, where it runs at full width as before (1 M
// users: the sweep outlasts the recurrences, nothing to hide it under).  after_prefetch: it also queues behind the next
// batch's embedding gather, which keeps the bandwidth it is measured on.
bool early_table_sweep_wanted(const lstur_ctx* c) {
  const auto& f = c->fused;
  return f.on && !f.swept && has_user_table(c->d) && !c->d.lazy;
}
void early_table_sweep(lstur_ctx* c, int n, hipEvent_t forked, bool recurrent, bool after_prefetch) {
  auto& f = c->fused;
  const Dims& d = c->d;
  const lstur_cfg& cf = c->cfg;
  if (!early_table_sweep_wanted(c) || !table_stream(c)) return;
  hipStream_t st = c->side_t;
  (void)hipStreamWaitEvent(st, forked, 0);
  launch_user_rows_mark(c->ws.user_rows, n, c->ws.slot, st);
  if (after_prefetch) (void)hipStreamWaitEvent(st, c->ev_p1, 0);
  const int env_blocks = c->opt.sweep_blocks;
  int sweep_blocks = 0;   // 0: full width
  if (recurrent) {
    const double bytes = (double)d.Nu * d.Ue * 4.0 * 6.0 * (two_tables(d) ? 2.0 : 1.0);
    const double window_s = d.W * 3.4e-6 * 2.0 * 0.8;
    const double need = bytes / (window_s * 14e9);
    if (need <= 128.0) sweep_blocks = need < 32.0 ? 32 : (int)need;
  }
  if (env_blocks >= 0) sweep_blocks = env_blocks;
  {
    ProfScope _ps(c, "adam_user_table", st);
    launch_adam_table_unmarked(c->buf.user_emb, c->buf.user_m, c->buf.user_v, c->ws.slot, d.Nu, d.Ue, f.lr_t, cf.beta1,
                               cf.beta2, cf.eps, sweep_blocks, st);
  }
  if (two_tables(d))
    launch_adam_table_unmarked(c->buf.user_emb2, c->buf.user_m2, c->buf.user_v2, c->ws.slot, d.Nu, d.Ue, f.lr_t, cf.beta1,
                               cf.beta2, cf.eps, sweep_blocks, st);
  f.swept = true;
}
bool finish_table_rows_wanted(lstur_ctx* c) {
  const auto& f = c->fused;
  if (c->d.lazy) return f.on && !f.table_done && table_stream(c);
  return f.on && f.swept && !f.table_done;
}
// behind the row gradients (and their row masks): `forked` was recorded there on the caller's stream
void finish_table_rows(lstur_ctx* c, int n, hipEvent_t forked) {
  auto& f = c->fused;
  const Dims& d = c->d;
  const lstur_cfg& cf = c->cfg;
  hipStream_t st = d.lazy ? lazy_stream(c) : c->side_t;
  const int32_t* rows = c->ws.user_rows;
  (void)hipStreamWaitEvent(st, forked, 0);
  launch_user_grad_combine(rows, c->ws.duser, n, d.Ue, c->ws.first, c->ws.gsum, c->ws.slot, st);
  if (d.lazy) {   // no sweep at all: the batch's rows take iteration t now, every other row when it is next read / updated
    lazy_rows_step(c, rows, nullptr, n, f.t, f.lr_t, st);
    (void)hipEventRecord(c->ev_t1, st);
    f.table_done = true;
    return;
  }
  launch_adam_rows(c->buf.user_emb, c->buf.user_m, c->buf.user_v, rows, c->ws.slot, c->ws.gsum, n, d.Ue, f.lr_t, cf.beta1,
                   cf.beta2, cf.eps, st);
  if (two_tables(d)) {
    launch_user_grad_combine(rows, c->ws.duser2, n, d.Ue, c->ws.first, c->ws.gsum2, c->ws.slot, st);
    launch_adam_rows(c->buf.user_emb2, c->buf.user_m2, c->buf.user_v2, rows, c->ws.slot, c->ws.gsum2, n, d.Ue, f.lr_t,
                     cf.beta1, cf.beta2, cf.eps, st);
  }
  launch_user_slot_clear(rows, n, c->ws.slot, st);
  (void)hipEventRecord(c->ev_t1, st);
  f.table_done = true;
}

// ---- data-parallel step: the dense table sweep's zero-gradient half, one forward later -----------------------------
// lstur_adam_step with exchanged rows (every rank's batch) cannot hide the dense sweep of the user table(s) the way the
// fused step does (the ranks' rows are known only behind the all-reduce): it was 38 us on the critical path at 50 k users.
// A row outside the union of the batches takes the zero-gradient update, which depends on nothing but the row and lr_t --
// so that half waits: the union's rows are updated at once and stay MARKED, the next forward first brings ITS rows up to
// date (launch_defer_rows: marks them too), and the sweep of every unmarked row runs thin on the background stream under
// that forward's recurrence, clearing the marks behind it.  Same operations per element in the same order, hence the same
// tables on every rank (tests/test_gpu_dp_two_ranks.py).  Anything else that reads the tables (evaluation, lstur_flush_tables,
// a step without exchanged rows) runs the pending sweep first, in the open.  LSTUR_DP_DEFER=0: the sweep inside lstur_adam_step.
bool defer_wanted(lstur_ctx* c) {
  return c->opt.dp_defer && !c->d.lazy && has_user_table(c->d) && background_stream(c);
}
void defer_join(lstur_ctx* c, hipStream_t s) {
  if (c->defer.inflight) (void)hipStreamWaitEvent(s, c->ev_t1, 0);
  c->defer.inflight = false;
}
void defer_sweep_launch(lstur_ctx* c, int blocks, hipStream_t st) {   // every unmarked row of the table(s), then the marks
  const Dims& d = c->d;
  const lstur_cfg& cf = c->cfg;
  {
    ProfScope _ps(c, "adam_user_table", st);
    launch_adam_table_unmarked(c->buf.user_emb, c->buf.user_m, c->buf.user_v, c->ws.slot, d.Nu, d.Ue, c->defer.lr_t, cf.beta1,
                               cf.beta2, cf.eps, blocks, st);
  }
  if (two_tables(d))
    launch_adam_table_unmarked(c->buf.user_emb2, c->buf.user_m2, c->buf.user_v2, c->ws.slot, d.Nu, d.Ue, c->defer.lr_t, cf.beta1,
                               cf.beta2, cf.eps, blocks, st);
  launch_user_slot_clear(c->ws.pend_rows, c->defer.n_rows, c->ws.slot, st);
}
void defer_flush(lstur_ctx* c, hipStream_t s) {   // in the open, at full width
  defer_join(c, s);
  if (!c->defer.valid) return;
  defer_sweep_launch(c, 0, s);
  c->defer.valid = false;
}
// this forward's rows first (on the caller's stream, in front of the row gather) ...
void defer_rows_now(lstur_ctx* c, const int32_t* user, int n, hipStream_t s) {
  const Dims& d = c->d;
  const lstur_cfg& cf = c->cfg;
  const bool two = two_tables(d);
  launch_defer_rows(c->buf.user_emb, c->buf.user_m, c->buf.user_v, two ? c->buf.user_emb2 : nullptr, two ? c->buf.user_m2 : nullptr,
                    two ? c->buf.user_v2 : nullptr, user, n, c->ws.slot, d.Ue, c->defer.lr_t, cf.beta1, cf.beta2, cf.eps, s);
}
// ... then everything else under the recurrence (`forked`: recorded on the caller's stream behind the row gather)
void defer_sweep_background(lstur_ctx* c, int n, hipEvent_t forked, bool recurrent, bool after_prefetch) {
  const Dims& d = c->d;
  hipStream_t st = c->side_t;
  (void)hipStreamWaitEvent(st, forked, 0);
  if (after_prefetch) (void)hipStreamWaitEvent(st, c->ev_p1, 0);
  int blocks = 0;   // 0: full width
  if (recurrent) {  // (the thin sweep of early_table_sweep: as many workgroups as finish inside the two recurrences)
    const double bytes = (double)d.Nu * d.Ue * 4.0 * 6.0 * (two_tables(d) ? 2.0 : 1.0);
    const double need = bytes / (d.W * 3.4e-6 * 2.0 * 0.8 * 14e9);
    if (need <= 128.0) blocks = need < 32.0 ? 32 : (int)need;
  }
  defer_sweep_launch(c, blocks, st);
  launch_user_slot_clear(c->ws.user_rows, n, c->ws.slot, st);   // this forward's rows (marked by defer_rows_now)
  (void)hipEventRecord(c->ev_t1, st);
  c->defer.valid = false;
  c->defer.inflight = true;
}

// whole-row dropout probability of the user embedding in training (0: none)
float user_row_drop(const lstur_ctx* c) {
  const int a = c->d.arch;
  if (a == LSTUR_ARCH_DGRU) return 0.5f;
  const bool has_table = a != LSTUR_ARCH_NIGRU && a != LSTUR_ARCH_NIAVG && a != LSTUR_ARCH_NIATT;
  const float k = c->cfg.id_keep;
  return (has_table && k > 0.f && k < 1.f) ? 1.0f - k : 0.f;
}

void lazy_init(lstur_ctx* c, hipStream_t s) {
  if (c->lazy_init_done) return;
  const Dims& d = c->d;
  (void)hipMemsetAsync(c->ws.emb + (size_t)d.nt_max * d.Tp * d.D, 0, kEmbSlack * sizeof(float), s);
  launch_fill_i32(c->ws.slot, -1, d.Nu, s);
  if (d.lazy) {
    launch_fill_i32(c->ws.lazy_last, 0, d.Nu, s);
    (void)hipMemsetAsync(c->ws.lazy_lrt, 0, (size_t)kLazyLog * sizeof(float), s);
  }
  // Ex: the gather skips rows no window covers; the k tail of a window may still touch one (under a zero weight)
  if (c->ws.e3) (void)hipMemsetAsync(c->ws.e3, 0, convx_e_elems(d.nt_max, d.T, d.D, d.k, d.np) * 2, s);
  if (c->ws.set[1].e3 && c->ws.set[1].e3 != c->ws.set[0].e3)
    (void)hipMemsetAsync(c->ws.set[1].e3, 0, convx_e_elems(d.nt_max, d.T, d.D, d.k, d.np) * 2, s);
  if (c->ws.cs) launch_convx_scales_init(c->ws.cs, s);
  c->lazy_init_done = true;
}

// titles already in ws.ids as [n*W history | n*k1 candidates] ; runs the network up to user_vec / doc_vec
bool cand_split_wanted(lstur_ctx* c, int n);
void run_forward(lstur_ctx* c, const int32_t* user, int n, int k1, bool training, uint32_t step, hipStream_t s) {
  const Dims& d = c->d;
  Workspace& w = c->ws;
  const int nW = n * d.W, nt = nW + n * k1;
  const float drop = training ? c->cfg.dropout : 0.f;
  // dropout streams 0 (embeddings), 1 (conv outputs), 2 (dgru / id_keep rows)
  const uint32_t key0 = dropout_key(c->cfg.seed, step, 0), key1 = dropout_key(c->cfg.seed, step, 1);
  lazy_init(c, s);
  const bool mhsa = d.enc == LSTUR_ENC_MHSA;
  const bool usex = d.np != 0;   // split-operand conv kernels (convx.hip); else the exact-fp32 MFMA kernels of conv.hip
  // f16x2: the embeddings' scale follows the word table (bound at create, changed through lstur_params_changed)
  if (d.np == 2 && !c->escale_valid) {
    launch_convx_escale(c->buf.word_emb, (size_t)d.V * d.D, c->cfg.dropout, w.cs, s);
    c->escale_valid = true;
  }
  // operand images of the weights used by MFMA kernels this step
  bool packs_forked = false;
  if (!c->images_valid) {
    // (the Adam entry points leave valid images behind: this runs on a first step, after lstur_params_changed or in eval)
    // the two weight images depend on nothing in the batch: they are packed on the side stream, beside the title staging and
    // the gather, and joined in front of the conv forward; with the staging prefetched there is nothing to hide them under,
    // and two 5-us kernels in line cost less than a fork / join across queues (~10 us each way)
    hipStream_t sp = c->staged_hit ? nullptr : wgrad_stream(c);
    if (sp) {
      (void)hipEventRecord(c->ev_w1, s);
      (void)hipStreamWaitEvent(sp, c->ev_w1, 0);
      packs_forked = true;
    } else {
      sp = s;
    }
    pack_weight_images(c, sp, true);
    if (packs_forked) (void)hipEventRecord(c->ev_w3, sp);
    c->images_valid = true;
  }
#ifdef LSTUR_ABLATE
  // TIMING ONLY (wrong results): what would a recurrence that runs BESIDE the news encoder buy?  LSTUR_PIPE_PROBE bit 0: the
  // forward recurrence starts here on the parameter-gradient stream (stale inputs); bit 1: the history titles' encoder
  // backward runs beside BPTT (run_backward)
  static const int pipe_probe = getenv("LSTUR_PIPE_PROBE") ? atoi(getenv("LSTUR_PIPE_PROBE")) : 0;
  static hipEvent_t px0 = nullptr, px1 = nullptr;
  bool probe_fwd = false;
  if ((pipe_probe & 1) && training && d.arch == LSTUR_ARCH_IGRU && wgrad_stream(c)) {
    if (!px0) { (void)hipEventCreateWithFlags(&px0, hipEventDisableTiming); (void)hipEventCreateWithFlags(&px1, hipEventDisableTiming); }
    hipStream_t sw = wgrad_stream(c);
    (void)hipEventRecord(px0, s);
    (void)hipStreamWaitEvent(sw, px0, 0);
    launch_gru_fwd(w.xw, w.gmask, w.uvec, w.gru_up, w.gru_tape, w.uservec, n, d.W, d.H, sw, c->opt.gru_f16);
    (void)hipEventRecord(px1, sw);
    probe_fwd = true;
  }
#endif
  // news encoder over all titles of the batch (TimeDistributed + K1 candidate calls, paper.py:641-650)
  const int pl = (d.k - 1) / 2;
  if (c->staged_hit) {
    // token ids, compacted list, offsets and Ex of this batch were staged by the prefetch of the previous step
    (void)hipStreamWaitEvent(s, c->ev_p1, 0);
  } else {
    // stable list of the non-pad tokens of this batch (+ the kernel gradient's row offsets), two launches
    if (!c->counted) launch_compact_count(w.ids, nt * d.T, w.vscratch, s);
    c->counted = false;
    // (offA: a token's window offset in the units of its path -- 16-byte pieces of Ex, or floats of the fp32 layout)
    launch_compact_write_offsets(w.ids, nt * d.T, w.vscratch, w.vidx, w.vcount, d.T, d.Tp,
                                 usex ? convx_dp(d.D) / 8 * d.np : d.D, d.F, w.offA, w.offB, s, w.live, nW * d.T);
    if (usex) {   // Embedding + Dropout straight into the pre-split layout of the conv kernels (rows nobody reads skipped)
      ProfScope _ps(c, "gather", s, true);
      launch_gatherx(c->buf.word_emb, w.ids, w.e3, nt, d.T, d.D, pl, d.k - 1 - pl, drop, key0, true, d.np, w.cs, s, 0, w.live);
    } else if (mhsa) {   // bf16 rows for the projections (they rounded the fp32 rows the same way on their way into LDS)
      PROF("gather", launch_gather_bf16(c->buf.word_emb, w.ids, (unsigned short*)w.emb, nt, d.T, d.D, pl, d.k - 1 - pl, drop, key0, s));
    } else {
      PROF("gather", launch_gather(c->buf.word_emb, w.ids, w.emb, nt, d.T, d.D, pl, d.k - 1 - pl, drop, key0, s));
    }
  }
  if (packs_forked) (void)hipStreamWaitEvent(s, c->ev_w3, 0);   // the weight images are ready
  if (mhsa) {   // token stage = multi-head self-attention (build-defined, include/lstur.h: LSTUR_ENC_MHSA)
    PROF("qkv_proj", launch_proj_fwd_bf16e((const unsigned short*)w.emb, w.qkv_w16, c->P("qkv_b"), w.qkv, nt * d.T, 3 * d.F, d.D, s));
    PROF("mhsa_fwd", launch_mhsa_fwd_bf16(w.qkv, w.ids, w.conv, nt, d.T, d.F, d.heads, s));
  } else if (usex) {
    ProfScope _ps(c, "conv_fwd", s, true);
    launch_convx_fwd(w.e3, w.conv_w3, c->P("conv_b"), w.ids, w.conv, nt, d.T, d.D, d.k, d.F, w.vidx, w.vcount, d.np, w.cs,
                     w.fwd2_scratch, s);
  } else {
    PROF("conv_fwd", launch_conv_fwd(w.emb, w.conv_wp, c->P("conv_b"), w.ids, w.conv, nt, d.T, d.D, d.k, d.F, w.vidx,
                                     w.vcount, s));
  }
  PROF("attpool_fwd", launch_attpool_fwd(w.conv, w.ids, c->P("att_w"), c->P("att_b"), w.pooled, w.att_a, w.att_wt, nt, d.T, d.F,
                                         drop, key1, s, w.live, (training && w.kbits_valid) ? w.kbits : nullptr));
  dense_gemm(w.pooled, d.F, c->P("dense_w"), d.U, false, w.docvec, d.U, c->P("dense_b"), nt, d.U, d.F, w.gemm_scratch, s);
  // user encoder (paper.py:584-633)
  if (d.lazy) lazy_catchup_rows(c, user, n, s);   // (large tables: the rows read below, made current)
  bool want_df = false;   // a deferred table sweep (data-parallel step) is pending: this batch's rows now, the rest under the recurrence
  if (c->defer.valid || c->defer.inflight) {
    defer_join(c, s);
    if (c->defer.valid) {
      if (training && defer_wanted(c) && fork_events(c)) {
        defer_rows_now(c, user, n, s);
        want_df = true;
      } else {
        defer_flush(c, s);
      }
    }
  }
  launch_hist_mask(w.ids, w.docvec, w.hmask, w.gmask, nW, d.T, d.U, s, c->buf.user_emb, user, w.user_rows, w.uvec, n,
                   d.Ue, w.user_rows_f);
  // whole-row dropout of the user embedding: dgru's Dropout(0.5, noise_shape=(None, 1)) (paper.py:610), or the id_keep
  // masking of cook.py:139-142 for every arch that reads the table
  const float row_p = training ? user_row_drop(c) : 0.f;
  if (row_p > 0.f) launch_drop_rows(w.uvec, n, d.Ue, row_p, dropout_key(c->cfg.seed, step, 2), s);
  const bool recurrent = d.arch != LSTUR_ARCH_VO && d.arch != LSTUR_ARCH_NIAVG && d.arch != LSTUR_ARCH_NIATT;
  // background work for the recurrence to hide: the next batch's staging, then (fused step) the zero-gradient half of the
  // user-table sweep.  ONE fork point for both side streams, attached to the input projection's completion.
  const bool want_pf = training && recurrent && usex && prefetch_wanted(c);
  const bool want_sw = training && early_table_sweep_wanted(c);
  const bool fork_f = (want_pf || want_sw || want_df) && fork_events(c);
  if (fork_f) arm_stop_event(c->ev_f0);
  dense_gemm(w.docvec, d.U, c->P("gru_w"), 3 * d.H, false, w.xw, 3 * d.H, c->P("gru_b"), nW, 3 * d.H, d.U, w.gemm_scratch, s);
  if (stop_event_pending()) {   // (the fp32 kernel took the launch: record the classic way)
    arm_stop_event(nullptr);
    (void)hipEventRecord(c->ev_f0, s);
  }
  if (fork_f) {
    const bool pf = want_pf && launch_prefetch(c, c->ev_f0);
    early_table_sweep(c, n, c->ev_f0, recurrent, pf);
    if (want_df) defer_sweep_background(c, n, c->ev_f0, recurrent, pf);
  }
  if (d.arch == LSTUR_ARCH_VO) {          // the user embedding alone (paper.py:622-623)
    (void)hipMemcpyAsync(w.uservec, w.uvec, (size_t)n * d.U * 4, hipMemcpyDeviceToDevice, s);
  } else if (d.arch == LSTUR_ARCH_NIAVG) {   // masked mean of the history vectors (paper.py:629-630)
    launch_masked_mean(w.docvec, w.gmask, w.uservec, nullptr, nullptr, n, d.W, d.U, s);
  } else if (d.arch == LSTUR_ARCH_NIATT) {   // additive attention over the history vectors (paper.py:207-209): the title
    // kernels with "tokens" = history slots, "filters" = U, no dropout; Masking = any(x != 0), like the title path
    launch_attpool_fwd(w.docvec, nullptr, c->P("uatt_w"), c->P("uatt_b"), w.uservec, w.ua_a, w.ua_w, n, d.W, d.U, 0.f, 0u, s);
  } else {
    // igru / nigru: the last state IS the user vector (H == U): the recurrence writes it in place
    const bool direct = d.arch == LSTUR_ARCH_IGRU || d.arch == LSTUR_ARCH_NIGRU;
    const bool init_user = d.arch == LSTUR_ARCH_IGRU || two_tables(d);   // initial_state = user_vec
    // igru / nigru with the dot scorer: the recurrence's kernel scores its samples itself (GruScore); the candidate titles'
    // backward then forks behind IT (run_listwise_score keeps the bookkeeping)
    GruScore sc;
    const bool fuse = c->sfuse.want && direct && d.head == LSTUR_HEAD_SOFTMAX && d.score == LSTUR_SCORE_DOT && k1 == d.K1 &&
                      d.H == d.U && gru_score_fusable(d.H, k1);
    if (fuse) {
      const bool bw = c->sfuse.backward;
      sc.d = w.docvec + (size_t)nW * d.U;
      sc.logits = w.logits; sc.probs = w.probs; sc.loss_i = w.loss_i; sc.correct_i = w.correct_i;
      sc.du = bw ? w.du : nullptr;
      sc.dd = bw ? w.ddocvec + (size_t)nW * d.U : nullptr;
      sc.K1 = k1;
      sc.inv_n = c->sfuse.inv_n;
      const bool fork_c = bw && cand_split_wanted(c, n);
      if (fork_c) arm_stop_event(c->ev_c0);
      c->cand_forked = fork_c;
    }
#ifdef LSTUR_ABLATE
    if (probe_fwd) (void)hipStreamWaitEvent(s, px1, 0);
    else
#endif
    PROF("gru_fwd", launch_gru_fwd(w.xw, w.gmask, init_user ? w.uvec : nullptr, w.gru_up,
                                   w.gru_tape, direct ? w.uservec : w.hlast, n, d.W, d.H, s, c->opt.gru_f16, fuse ? &sc : nullptr));
    if (fuse) {
      if (stop_event_pending()) {   // (nothing took the armed event: record the classic way)
        arm_stop_event(nullptr);
        (void)hipEventRecord(c->ev_c0, s);
      }
      c->sfuse.done = true;
    }
    if (d.arch == LSTUR_ARCH_PGRU) {        // GRU(U)(x) + user_vec (paper.py:624-626)
      launch_add2(w.hlast, w.uvec, w.uservec, (size_t)n * d.U, s);
    } else if (!direct) {
      const float* side = w.uvec;
      if (two_tables(d)) {     // the concatenated vector comes from the SECOND table (paper.py:617-619, 345-347)
        launch_gather_rows(c->buf.user_emb2, w.user_rows, w.uvec2, n, d.Ue, s);
        if (row_p > 0.f) launch_drop_rows(w.uvec2, n, d.Ue, row_p, dropout_key(c->cfg.seed, step, 2), s);
        side = w.uvec2;
      }
      launch_concat2(w.hlast, d.H, side, d.Ue, w.cat, n, s);
      if (d.arch == LSTUR_ARCH_GRU || d.arch == LSTUR_ARCH_IIGRU)
        launch_gemm(w.cat, d.H + d.Ue, 0, c->P("con_w"), d.U, 0, w.uservec, d.U, c->P("con_b"), n, d.U, d.H + d.Ue, 1.f,
                    w.gemm_scratch, s);
      else   // hgru (U/2 + U/2), ngru / dgru / iicgru (U + U): the concatenation is the user vector
        (void)hipMemcpyAsync(w.uservec, w.cat, (size_t)n * d.Uu * 4, hipMemcpyDeviceToDevice, s);
    }
  }
  c->last_n = n;
  c->last_k1 = k1;
  c->last_usex = usex;
}

// stats (may be NULL): the batch means of loss_i / correct_i ride the pass's single reduction launch
int run_backward(lstur_ctx* c, int n, uint32_t step, hipStream_t s, float* stats) {
  const Dims& d = c->d;
  Workspace& w = c->ws;
  const int nW = n * d.W, nt = nW + n * d.K1;
  const float drop = c->cfg.dropout;
  const uint32_t key1 = dropout_key(c->cfg.seed, step, 1);
  g_wgrad_bf16x3 = c->opt.wgrad_bf16x3;
  // f16x2 kernel gradient: the attention backward leaves max |dC| of every title in gparts[title] (title path; the
  // streaming kernels: one entry per workgroup), the kernel-gradient launches scale by the maximum over their titles
  float* gp = d.np == 2 ? w.gparts : nullptr;
  const bool gp_titles = attpool_title_path(d.T, d.F);
  // title-resident attention backward: one partial-sum row / one maximum per kAttBwdTitlesPerWg titles (attpool.hip)
  // (two titles while the launch is a few rounds of workgroups, four from 8 192 titles on: measured 0.906 / 0.908 / 0.910 ms
  //  per step with 2 / 3 / 4 titles at B = 64 against 0.919 with one; 4.19 / 4.19 ms with 4 / 8 at B = 512 against 4.30)
  const int tpb = gp_titles ? (nt >= 8192 ? 4 : 2) : 1;
  // user encoder head
  const float* dhl = w.du;
  // LSTUR-con's Dense over [h | user_vec]: its parameter gradients feed nothing but the optimizer -- they go to the side stream
  // together with the candidate titles' encoder backward (the fork behind the scorer), off the path to the recurrence
  const bool con_side = (d.arch == LSTUR_ARCH_GRU || d.arch == LSTUR_ARCH_IIGRU) && c->cand_forked && wgrad_stream(c) &&
                        wgrad_stream(c) != s && !c->cfg.freeze_encoder;
  if (d.arch == LSTUR_ARCH_GRU || d.arch == LSTUR_ARCH_IIGRU) {
    if (!con_side) {
      launch_gemm(w.cat, d.H + d.Ue, 1, w.du, d.U, 0, c->G("con_w"), d.U, nullptr, d.H + d.Ue, d.U, n, 1.f,
                  w.gemm_scratch, s);
      launch_colsum2(w.du, n, d.U, d.U, c->G("con_b"), 1.f, w.colsum_scratch, s);
    }
    launch_gemm(w.du, d.U, 0, c->P("con_w"), d.U, 1, w.dcat, d.H + d.Ue, nullptr, n, d.H + d.Ue, d.U, 1.f,
                w.gemm_scratch, s);
    launch_split2(w.dcat, d.H, d.Ue, w.dhlast, d.arch == LSTUR_ARCH_IIGRU ? w.duser2 : w.duser, n, s);
    dhl = w.dhlast;
  } else if (d.arch == LSTUR_ARCH_IICGRU) {          // u = [h | user_vec2], no Dense between (paper.py:343-348)
    launch_split2(w.du, d.H, d.Ue, w.dhlast, w.duser2, n, s);
    dhl = w.dhlast;
  } else if (d.arch == LSTUR_ARCH_HGRU || d.arch == LSTUR_ARCH_NGRU || d.arch == LSTUR_ARCH_DGRU) {
    launch_split2(w.du, d.H, d.Ue, w.dhlast, w.duser, n, s);
    dhl = w.dhlast;
  } else if (d.arch == LSTUR_ARCH_PGRU || d.arch == LSTUR_ARCH_VO) {   // u = h + user_vec / u = user_vec
    (void)hipMemcpyAsync(w.duser, w.du, (size_t)n * d.U * 4, hipMemcpyDeviceToDevice, s);
  }
  const bool has_gru = d.arch != LSTUR_ARCH_VO && d.arch != LSTUR_ARCH_NIAVG && d.arch != LSTUR_ARCH_NIATT;
  ReduceJob jobs[kMaxReduceJobs];   // (make_dims bounds the count: 21 + kernel)
  int nj = 0;
  hipStream_t sw = wgrad_stream(c);
  if (!sw) sw = s;
  bool forked = false;
  auto join = [&]() {     // the step's single reduction (on s) reads what the side stream produced
    if (forked) {
      (void)hipEventRecord(c->ev_w3, sw);
      (void)hipStreamWaitEvent(s, c->ev_w3, 0);
    }
  };
  auto finish = [&]() -> int {
    if (stats) {
      jobs[nj++] = ReduceJob{w.loss_i, stats, 1, 1, 1, n, 1, 1.0f / (float)n};
      jobs[nj++] = ReduceJob{w.correct_i, stats + 1, 1, 1, 1, n, 1, 1.0f / (float)n};
      jobs[nj++] = ReduceJob{w.loss_i, stats + 2, 1, 1, 1, n, 1, 1.0f / (float)n, 1.0f};      // running sums for the
      jobs[nj++] = ReduceJob{w.correct_i, stats + 3, 1, 1, 1, n, 1, 1.0f / (float)n, 1.0f};   // host's epoch means
    }
    // data-parallel step (lstur_dp_attach): this rank's (row, gradient) pairs go into the bucket's tail in the same launch
    if (c->dp_tail && has_user_table(d) && c->buf.user_emb)
      nj += dp_stage_jobs(w.user_rows_f, w.duser, two_tables(d) ? w.duser2 : nullptr, n, d.B, d.Ue, c->dp_tail, c->dp_rank,
                          c->dp_world, &jobs[nj]);
    join();
    if (!launch_multi_reduce(jobs, nj, s)) return fail(LSTUR_ERR_STATE, "backward: too many deferred reductions");
    return LSTUR_OK;
  };
  // the candidate titles' encoder backward, under the recurrence's backward (cand_split_wanted)
  const bool split = c->cand_forked && sw != s && !c->cfg.freeze_encoder;
  c->cand_forked = false;
  const int ncand = nt - nW;
#ifdef LSTUR_ABLATE
  const bool probe_bwd = split && getenv("LSTUR_PIPE_PROBE") && (atoi(getenv("LSTUR_PIPE_PROBE")) & 2);
#endif
  // kernel-gradient slabs: [0] the side launch's slabs reduced ahead of time | the main launch's | the side launch's
  const size_t slab_f = d.np ? convx_wgrad_slab_floats(d.D, d.k, d.F) : 0;
  const int main_slabs = split ? convx_wgrad_slabs(nW, d.T, d.D, d.k, d.F) : 0;
  const int side_base = 1 + main_slabs;
  if (split) {
    (void)hipStreamWaitEvent(sw, c->ev_c0, 0);
    forked = true;
    if (con_side) {   // (scratch of their own: the main stream's products use gemm_scratch / colsum_scratch meanwhile)
      launch_gemm(w.cat, d.H + d.Ue, 1, w.du, d.U, 0, c->G("con_w"), d.U, nullptr, d.H + d.Ue, d.U, n, 1.f, w.gemm_scratch_k[3], sw);
      launch_colsum2(w.du, n, d.U, d.U, c->G("con_b"), 1.f, w.colsum_scratch_k[1], sw);
    }
    dense_gemm(w.ddocvec + (size_t)nW * d.U, d.U, c->P("dense_w"), d.U, true, w.dpooled + (size_t)nW * d.F, d.F, nullptr, ncand,
               d.F, d.U, nullptr, sw);
    const int off_cb = 0, off_aw = (int)(c->G("att_w") - c->G("conv_b")), off_ab = (int)(c->G("att_b") - c->G("conv_b"));
    const int ldp = off_ab + 1;
    int cand_slabs = 0;
    {   // (families of their own in the profile: these launches run beside the recurrence, off the critical path)
      ProfScope _ps(c, "attpool_bwd_side", sw);
      launch_attpool_bwd(w.conv, w.ids, c->P("att_w"), w.att_a, w.att_wt, w.dpooled, w.dconv, w.parts, w.att_ds, nt, d.T, d.F,
                         drop, key1, sw, ldp, off_aw, off_cb, off_ab, 1, w.live, w.kbits_valid ? w.kbits : nullptr, nW, ncand,
                         gp, tpb, cdiv(nW, tpb));
    }
    {
      ProfScope _ps(c, "conv_wgrad_side", sw, true);
      (void)launch_convx_wgrad(w.e3, w.dconv, c->G("conv_w"), w.wgrad_scratch, nt, d.T, d.D, d.k, d.F, w.vcount, sw, nullptr, w.offA,
                               w.offB, d.np, w.cs, gp ? gp + cdiv(nW, tpb) : nullptr, cdiv(ncand, tpb), 1.0f, 2, side_base, ncand,
                               &cand_slabs);
    }
#ifdef LSTUR_ABLATE
    if (getenv("LSTUR_PIPE_PROBE") && (atoi(getenv("LSTUR_PIPE_PROBE")) & 2)) {   // TIMING ONLY: stale dgates
      dense_gemm(w.dgates, 3 * d.H, c->P("gru_w"), 3 * d.H, true, w.ddocvec, d.U, nullptr, nW, d.U, 3 * d.H, w.gemm_scratch_k[0], sw);
      dense_gemm(w.ddocvec, d.U, c->P("dense_w"), d.U, true, w.dpooled, d.F, nullptr, nW, d.F, d.U, w.gemm_scratch_k[0], sw);
      launch_attpool_bwd(w.conv, w.ids, c->P("att_w"), w.att_a, w.att_wt, w.dpooled, w.dconv, w.parts, w.att_ds, nt, d.T, d.F,
                         drop, key1, sw, ldp, off_aw, off_cb, off_ab, 1, w.live, w.kbits_valid ? w.kbits : nullptr, 0, nW, gp);
      ProfScope _ps(c, "conv_wgrad", sw, true);
      nj += launch_convx_wgrad(w.e3, w.dconv, c->G("conv_w"), w.wgrad_scratch, nt, d.T, d.D, d.k, d.F, w.vcount, sw, &jobs[nj], w.offA,
                               w.offB, d.np, w.cs, gp, nW, 1.0f, 1, 1, nW);
    }
#endif
    // the side launch's slabs -> slab 0 (off the critical path), then the join event
    ReduceJob pre{(const float*)w.wgrad_scratch + (size_t)side_base * slab_f, (float*)w.wgrad_scratch, d.k * convx_dp(d.D), d.F, d.F,
                  cand_slabs, slab_f, 1.0f, 0.f};
    (void)launch_multi_reduce(&pre, 1, sw);
    (void)hipEventRecord(c->ev_c1, sw);
  }
  // ONE event behind the recurrence serves both side streams (forks are merged wherever the dependencies allow), attached
  // to the recurrence kernel's own completion when nothing else sits between it and the fork
  const bool fork_t = finish_table_rows_wanted(c);
  hipEvent_t fe = sw != s ? c->ev_w1 : c->ev_t0;   // (ev_t0 exists whenever fork_t does)
  const bool fork_b = (has_gru && sw != s) || fork_t;
  bool fe_recorded = false;
  if (has_gru) {
    float *hprev = w.hprev, *rh = w.rh;   // fast path: both already sit row-major in the tape (no copies)
    gru_tape_rows(w.gru_tape, n, d.W, d.H, &hprev, &rh);
    const bool attach = fork_b && !(user_row_drop(c) > 0.f) && !(c->prof_on && !c->prof_roofline_only);   // (a bracketed launch records behind it anyway)
    if (attach) arm_stop_event(fe);
    PROF("gru_bwd", launch_gru_bwd(dhl, w.gmask, w.gru_tape, w.gru_up, w.dgates, rh, hprev,
                                   (d.arch == LSTUR_ARCH_IGRU || two_tables(d)) ? w.duser : w.dh0, n, d.W, d.H, s));
    fe_recorded = attach && !stop_event_pending();
    arm_stop_event(nullptr);
  }
  {   // the row gradients of the user table(s) pass the forward's row mask and scale (dgru / id_keep)
    const float row_p = user_row_drop(c);
    if (row_p > 0.f) {
      launch_drop_rows(w.duser, n, d.Ue, row_p, dropout_key(c->cfg.seed, step, 2), s);
      if (two_tables(d)) launch_drop_rows(w.duser2, n, d.Ue, row_p, dropout_key(c->cfg.seed, step, 2), s);
    }
  }
  if (fork_b && !fe_recorded) (void)hipEventRecord(fe, s);
  if (fork_t) finish_table_rows(c, n, fe);   // fused step: the batch's own rows of the user table(s)
  if (has_gru) {
    float *hprev = w.hprev, *rh = w.rh;
    gru_tape_rows(w.gru_tape, n, d.W, d.H, &hprev, &rh);
    // GRU parameter gradients (their split-K / column-sum partials are reduced by ONE launch at the end of the pass), on
    // the side stream: nothing but that reduction reads them
    hipStream_t sg = sw;
#ifdef LSTUR_ABLATE
    if (probe_bwd) sg = s;
#endif
    if (sg != s) {
      (void)hipStreamWaitEvent(sg, fe, 0);
      forked = true;
    }
    wgrad_gemm(w.docvec, d.U, w.dgates, 3 * d.H, c->G("gru_w"), 3 * d.H, d.U, 3 * d.H, nW, w.gemm_scratch_k[0], sg, &jobs[nj++]);
    launch_colsum2(w.dgates, nW, 3 * d.H, 3 * d.H, c->G("gru_b"), 1.f, w.colsum_scratch_k[0], sg, &jobs[nj++]);
    wgrad_gemm(hprev, d.H, w.dgates, 3 * d.H, c->G("gru_u"), 3 * d.H, d.H, 2 * d.H, nW, w.gemm_scratch_k[1], sg, &jobs[nj++]);
    wgrad_gemm(rh, d.H, w.dgates + 2 * d.H, 3 * d.H, c->G("gru_u") + 2 * d.H, 3 * d.H, d.H, d.H, nW, w.gemm_scratch_k[2], sg,
               &jobs[nj++]);
  } else {   // vo / niavg: no GRU in the graph -- its bucket slots get an exact zero gradient
    (void)hipMemsetAsync(c->G("gru_w"), 0, (size_t)(c->G("gru_b") + 3 * d.H - c->G("gru_w")) * sizeof(float), s);
  }
  if (c->cfg.freeze_encoder)   // loaded, non-trainable news encoder: nothing behind the doc vectors is trained
    return finish();
  // d(history doc vectors) = dgates . gru_w^T, masked by the history mask (paper.py:644-645)
  bool w2_recorded = false;
  if (has_gru) {
    if (sw != s) arm_stop_event(c->ev_w2);   // the Dense(U) parameter gradients fork behind this product
#ifdef LSTUR_ABLATE
    if (!probe_bwd)
#endif
    dense_gemm(w.dgates, 3 * d.H, c->P("gru_w"), 3 * d.H, true, w.ddocvec, d.U, nullptr, nW, d.U, 3 * d.H, w.gemm_scratch, s);
    w2_recorded = sw != s && !stop_event_pending();
    arm_stop_event(nullptr);
  }
  else if (d.arch == LSTUR_ARCH_NIAVG)   // dx[b,w] = mask[b,w] * du[b] / (count_b + eps)
    launch_masked_mean(nullptr, w.gmask, nullptr, w.du, w.ddocvec, n, d.W, d.U, s);
  else if (d.arch == LSTUR_ARCH_NIATT) {   // backward of the attention over the history (no ReLU behind it: relu = 0)
    launch_attpool_bwd(w.docvec, nullptr, c->P("uatt_w"), w.ua_a, w.ua_w, w.du, w.ddocvec, w.ua_parts, w.ua_ds, n, d.W, d.U,
                       0.f, 0u, s, 0, 0, 0, 0, 0);
    // partial sums per row: datt_w[U] | (column sums of dx: unused)[U] | datt_b
    const int np_ = attpool_bwd_parts(n);
    launch_colsum2(w.ua_parts, np_, d.U, 2 * d.U + 1, c->G("uatt_w"), 1.f, w.colsum_scratch, s);
    launch_colsum2(w.ua_parts + 2 * d.U, np_, 1, 2 * d.U + 1, c->G("uatt_b"), 1.f, w.colsum_scratch, s);
  }
  else                                   // vo: the history never reaches the loss
    (void)hipMemsetAsync(w.ddocvec, 0, (size_t)nW * d.U * sizeof(float), s);
  // (the history mask needs no pass here: a padded slot is a masked GRU step, its dgates row -- hence its ddocvec
  //  row -- is exactly 0, cf. paper.py:644-645)
  // news encoder backward (candidate rows of ddocvec were written by the scorer)
  if (sw != s) {          // Dense(U) parameter gradients: beside d(pooled) -> attention backward -> conv kernel gradient
    if (!w2_recorded) (void)hipEventRecord(c->ev_w2, s);
    (void)hipStreamWaitEvent(sw, c->ev_w2, 0);
    forked = true;
  }
  {
    hipStream_t sg = sw;
#ifdef LSTUR_ABLATE
    if (probe_bwd) sg = s;
#endif
    wgrad_gemm(w.pooled, d.F, w.ddocvec, d.U, c->G("dense_w"), d.U, d.F, d.U, nt, w.gemm_scratch_k[3], sg, &jobs[nj++]);
    launch_colsum2(w.ddocvec, nt, d.U, d.U, c->G("dense_b"), 1.f, w.colsum_scratch_k[1], sg, &jobs[nj++]);
  }
#ifdef LSTUR_ABLATE
  if (!probe_bwd)
#endif
  dense_gemm(w.ddocvec, d.U, c->P("dense_w"), d.U, true, w.dpooled, d.F, nullptr, split ? nW : nt, d.F, d.U, w.gemm_scratch, s);
  // partial sums laid out like the bucket slice conv_b .. att_b -> one reduction straight into dense_grad
  if (d.enc == LSTUR_ENC_MHSA) {
    // partial sums: datt_w | datt_b like the bucket slice att_w .. att_b, the (unused) column sums of dconv behind them
    const int off_aw = 0, off_ab = (int)(c->G("att_b") - c->G("att_w")), off_cb = off_ab + (int)kParamAlign;
    const int ldp = off_cb + (int)((d.F + kParamAlign - 1) / kParamAlign * kParamAlign);
    PROF("attpool_bwd", launch_attpool_bwd(w.conv, w.ids, c->P("att_w"), w.att_a, w.att_wt, w.dpooled, w.dconv, w.parts,
                                           w.att_ds, nt, d.T, d.F, drop, key1, s, ldp, off_aw, off_cb, off_ab, 0, w.live,
                                           w.kbits_valid ? w.kbits : nullptr, 0, nt, nullptr, tpb, 0));
    launch_colsum2(w.parts, gp_titles ? cdiv(nt, tpb) : attpool_bwd_parts(nt), off_ab + 1, ldp, c->G("att_w"), 1.f,
                   w.colsum_scratch_k[2], s, &jobs[nj++]);
    PROF("mhsa_bwd", launch_mhsa_bwd_bf16(w.qkv, w.dconv, w.ids, w.dqkv, nt, d.T, d.F, d.heads, s));
    // weight AND bias gradient in one product: the bucket keeps qkv_b right behind qkv_w (D * 3F is a multiple of the
    // slot alignment), so row D of [E | 1]^T . dQKV lands on it
    PROF("qkv_wgrad", launch_proj_wgrad_bf16e((const unsigned short*)w.emb, w.dqkv, c->G("qkv_w"), w.mh_scratch, d.D, 3 * d.F, nt * d.T, s,
                                             &jobs[nj++]));
    return finish();
  }
  const int off_cb = 0, off_aw = (int)(c->G("att_w") - c->G("conv_b")), off_ab = (int)(c->G("att_b") - c->G("conv_b"));
  const int ldp = off_ab + 1;
#ifdef LSTUR_ABLATE
  if (!probe_bwd)
#endif
  PROF("attpool_bwd", launch_attpool_bwd(w.conv, w.ids, c->P("att_w"), w.att_a, w.att_wt, w.dpooled, w.dconv, w.parts,
                                         w.att_ds, nt, d.T, d.F, drop, key1, s, ldp, off_aw, off_cb, off_ab, 1, w.live,
                                         w.kbits_valid ? w.kbits : nullptr, 0, split ? nW : nt, gp, tpb, 0));
  // rows of partial sums the launch(es) left: the main launch's, then the candidate titles' launch's
  const int att_rows = !gp_titles ? attpool_bwd_parts(nt) : split ? cdiv(nW, tpb) + cdiv(ncand, tpb) : cdiv(nt, tpb);
  if (split) (void)hipStreamWaitEvent(s, c->ev_c1, 0);   // the candidate titles' partial sums, gradient rows and slabs
  // (on the side stream this one found no free CU beside the conv kernel gradient and finished behind it)
  launch_colsum2(w.parts, att_rows, ldp, ldp, c->G("conv_b"), 1.f, w.colsum_scratch_k[2], s, &jobs[nj++]);
  if (d.word_trainable) {   // conv input gradient -> word-table gradient (embedding dropout mask = stream 0 of the step)
    (void)hipMemsetAsync(c->buf.word_grad, 0, (size_t)d.V * d.D * sizeof(float), s);
    PROF("conv_dgrad", launch_conv_dgrad_scatter(w.dconv, c->P("conv_w"), w.ids, c->buf.word_grad, w.dgrad_scratch, nt, d.T,
                                                  d.D, d.k, d.F, drop, dropout_key(c->cfg.seed, step, 0), s));
  }
#ifdef LSTUR_ABLATE
  if (probe_bwd) return finish();   // (the kernel gradient ran on the side stream)
#endif
  if (d.np) {   // split-operand contraction (convx.hip), like the forward; one reduce job per tap
    ProfScope _ps(c, "conv_wgrad", s, true);
    nj += launch_convx_wgrad(w.e3, w.dconv, c->G("conv_w"), w.wgrad_scratch, nt, d.T, d.D, d.k, d.F, w.vcount, s, &jobs[nj],
                             w.offA, w.offB, d.np, w.cs, gp, !gp_titles ? attpool_bwd_parts(nt) : cdiv(split ? nW : nt, tpb), 1.0f,
                             split ? 1 : 0, split ? 1 : 0, split ? nW : 0);
  } else
    PROF("conv_wgrad", launch_conv_wgrad(w.emb, w.dconv, c->G("conv_w"), w.wgrad_scratch, nt, d.T, d.D, d.k, d.F,
                                         w.vidx, w.vcount, s, w.offA, w.offB,
                                         w.emb + (size_t)d.nt_max * d.Tp * d.D /* zero slack */, &jobs[nj++]));
  return finish();
}

int stage_titles(lstur_ctx* c, const int32_t* hist, const int32_t* cands, int n, int k1, bool ids_form, hipStream_t s) {
  const Dims& d = c->d;
  Workspace& w = c->ws;
  const size_t nW = (size_t)n * d.W;
  if (ids_form) {
    if (!c->title_table) return fail(LSTUR_ERR_STATE, "no title table bound (lstur_set_title_table)");
    launch_expand_count(c->title_table, hist, (int)nW, cands, n * k1, d.T, w.ids, w.vscratch, s);
    c->counted = true;
  } else {
    (void)hipMemcpyAsync(w.ids, hist, nW * d.T * 4, hipMemcpyDeviceToDevice, s);
    (void)hipMemcpyAsync(w.ids + nW * d.T, cands, (size_t)n * k1 * d.T * 4, hipMemcpyDeviceToDevice, s);
  }
  return LSTUR_OK;
}

// The candidate titles' share of the news-encoder backward (d(pooled) -> attention backward -> conv kernel gradient, 9 % of
// the titles at 5 candidates against 50 history slots) needs nothing from the recurrence: d(candidate doc vectors) comes
// out of the scorer.  It runs on the parameter-gradient stream UNDER the recurrence's backward (240 idle CUs) instead of
// behind it; the history titles follow on the caller's stream as before.  Same kernels on title / token sub-ranges (the
// compaction counts the tokens in front of the first candidate title), the kernel gradient's slabs of both launches are
// reduced together in fixed order.  LSTUR_CAND_SPLIT=0: one launch each over all titles.
bool cand_split_wanted(lstur_ctx* c, int n) {
  const Dims& d = c->d;
  const bool has_gru = d.arch != LSTUR_ARCH_VO && d.arch != LSTUR_ARCH_NIAVG && d.arch != LSTUR_ARCH_NIATT;
  if (!c->opt.cand_split || !has_gru || !d.np) return false;
  if (d.enc != LSTUR_ENC_CNNATT || d.word_trainable || c->cfg.freeze_encoder) return false;
  if (d.head != LSTUR_HEAD_SOFTMAX || d.score != LSTUR_SCORE_DOT || !attpool_title_path(d.T, d.F)) return false;
  if (!gemm3_supported(d.U, d.U, n * d.K1, d.F, d.U) || !wgrad_stream(c)) return false;
  // the recurrence's backward must leave most of the chip idle (one workgroup per 4 samples): at B = 512 it holds 128 CUs
  // and the side launches only get in its way (same box: 6.05 against 5.94 ms per step)
  if (cdiv(n, 4) > 32) return false;
  if (!c->ev_c0 && (hipEventCreateWithFlags(&c->ev_c0, hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&c->ev_c1, hipEventDisableTiming) != hipSuccess)) {
    c->ev_c0 = c->ev_c1 = nullptr;
    return false;
  }
  return true;
}

// Listwise score model on (uservec [n, Uu], candidate doc vectors [n*k1, U]) (paper.py:443-464).
//   scores != NULL: test model, sigmoid(score) for k1 == 1 (paper.py:661-665)
//   else: logits / probs / loss (+ `backward`: du, d(candidate doc vectors) and the score model's parameter gradients)
void run_listwise_score(lstur_ctx* c, int n, int k1, float inv_n, bool backward, float* scores, hipStream_t s) {
  const Dims& d = c->d;
  Workspace& w = c->ws;
  const int nW = n * d.W, rows = n * k1;
  const float* dvec = w.docvec + (size_t)nW * d.U;
  float* ddvec = w.ddocvec + (size_t)nW * d.U;
  if (c->sfuse.done && !scores) {   // the forward recurrence's kernel scored its samples (run_forward)
    c->sfuse.done = false;
    return;
  }
  if (d.score == LSTUR_SCORE_DOT) {
    if (scores) launch_predict_score(w.uservec, dvec, scores, n, d.U, s);
    else {
      const bool fork_c = backward && k1 == d.K1 && cand_split_wanted(c, n);   // the candidates' encoder backward forks here
      if (fork_c) arm_stop_event(c->ev_c0);
      launch_score(w.uservec, dvec, w.logits, w.probs, w.loss_i, w.correct_i, backward ? w.du : nullptr,
                   backward ? ddvec : nullptr, n, k1, d.U, inv_n, s);
      if (fork_c && stop_event_pending()) {   // (the general scorer took the launch: record the classic way)
        arm_stop_event(nullptr);
        (void)hipEventRecord(c->ev_c0, s);
      }
      c->cand_forked = fork_c;
    }
    return;
  }
  if (d.score == LSTUR_SCORE_DDOT) {   // tanh-Dense on both sides, then dot (paper.py:452-455)
    float *uh = w.lw_c, *dh = w.lw_a, *duh = w.lw_d, *ddh = w.lw_b;   // [n,U], [rows,U], [n,U], [rows,U]
    launch_gemm(w.uservec, d.Uu, 0, c->P("sdu_w"), d.U, 0, uh, d.U, c->P("sdu_b"), n, d.U, d.Uu, 1.f, w.gemm_scratch, s);
    launch_tanh_inplace(uh, (size_t)n * d.U, s);
    launch_gemm(dvec, d.U, 0, c->P("sdd_w"), d.U, 0, dh, d.U, c->P("sdd_b"), rows, d.U, d.U, 1.f, w.gemm_scratch, s);
    launch_tanh_inplace(dh, (size_t)rows * d.U, s);
    if (scores) {
      launch_predict_score(uh, dh, scores, n, d.U, s);
      return;
    }
    launch_score(uh, dh, w.logits, w.probs, w.loss_i, w.correct_i, backward ? duh : nullptr, backward ? ddh : nullptr, n,
                 k1, d.U, inv_n, s);
    if (!backward) return;
    launch_tanh_bwd(duh, uh, (size_t)n * d.U, s);
    launch_tanh_bwd(ddh, dh, (size_t)rows * d.U, s);
    launch_gemm(w.uservec, d.Uu, 1, duh, d.U, 0, c->G("sdu_w"), d.U, nullptr, d.Uu, d.U, n, 1.f, w.gemm_scratch, s);
    launch_colsum2(duh, n, d.U, d.U, c->G("sdu_b"), 1.f, w.colsum_scratch, s);
    launch_gemm(dvec, d.U, 1, ddh, d.U, 0, c->G("sdd_w"), d.U, nullptr, d.U, d.U, rows, 1.f, w.gemm_scratch, s);
    launch_colsum2(ddh, rows, d.U, d.U, c->G("sdd_b"), 1.f, w.colsum_scratch, s);
    launch_gemm(duh, d.U, 0, c->P("sdu_w"), d.U, 1, w.du, d.Uu, nullptr, n, d.Uu, d.U, 1.f, w.gemm_scratch, s);
    launch_gemm(ddh, d.U, 0, c->P("sdd_w"), d.U, 1, ddvec, d.U, nullptr, rows, d.U, d.U, 1.f, w.gemm_scratch, s);
    return;
  }
  // dnn: concat -> Dense(U, relu) -> Dense(1) per candidate (paper.py:448-451)
  const int wc = d.Uu + d.U;
  float *cat = w.lw_a, *dcat = w.lw_b, *pre = w.lw_c, *dhid = w.lw_d;
  launch_concat_rep(w.uservec, d.Uu, k1, dvec, d.U, cat, rows, s);
  launch_gemm(cat, wc, 0, c->P("sn1_w"), d.U, 0, pre, d.U, c->P("sn1_b"), rows, d.U, wc, 1.f, w.gemm_scratch, s);
  // raw score = relu(pre) . sn2_w + sn2_b (lw_s); the kernel's sigmoid output is the test model's score
  launch_point_mlp_out(pre, c->P("sn2_w"), c->P("sn2_b"), nullptr, w.lw_s, scores ? scores : w.lw_ds, nullptr, nullptr,
                       nullptr, nullptr, rows, d.U, 1.f, 1, 0.f, s);
  if (scores) return;
  // softmax + categorical cross-entropy over the k1 scores of a sample: the dot scorer's kernel with u = 1, U = 1
  launch_fill_f32(w.lw_ones, 1.0f, n, s);
  launch_score(w.lw_ones, w.lw_s, w.logits, w.probs, w.loss_i, w.correct_i, backward ? w.sds : nullptr,
               backward ? w.lw_ds : nullptr, n, k1, 1, inv_n, s);
  if (!backward) return;
  launch_point_w2_grad(pre, w.lw_ds, c->G("sn2_w"), c->G("sn2_b"), rows, d.U, s);
  launch_mlp_dhid(pre, w.lw_ds, c->P("sn2_w"), dhid, rows, d.U, s);
  launch_gemm(cat, wc, 1, dhid, d.U, 0, c->G("sn1_w"), d.U, nullptr, wc, d.U, rows, 1.f, w.gemm_scratch, s);
  launch_colsum2(dhid, rows, d.U, d.U, c->G("sn1_b"), 1.f, w.colsum_scratch, s);
  launch_gemm(dhid, d.U, 0, c->P("sn1_w"), d.U, 1, dcat, wc, nullptr, rows, wc, d.U, 1.f, w.gemm_scratch, s);
  launch_split_sum(dcat, d.Uu, d.U, k1, w.du, ddvec, n, s);
}

int train_impl(lstur_ctx* c, const int32_t* user, const int32_t* hist, const int32_t* cands, int n, uint32_t step,
               float grad_scale, float* stats, hipStream_t s, bool ids_form, bool backward, bool training) {
  if (!c || !user || !hist || !cands) return fail(LSTUR_ERR_ARG, "NULL argument");
  const Dims& d = c->d;
  if (d.head != LSTUR_HEAD_SOFTMAX) return fail(LSTUR_ERR_STATE, "context was created with a pointwise head");
  if (n < 1 || n > d.B) return fail(LSTUR_ERR_ARG, "n must be in [1, max_batch]");
  // was this batch staged by a prefetch during the previous step?  (same arrays, same size, same step)
  auto& q = c->ready;
  c->staged_hit = backward && training && ids_form && q.valid && q.hist == hist && q.cands == cands && q.n == n &&
                  q.step == step;
  c->ws.kbits_valid = c->staged_hit && q.kbits;
  if (backward && training) q = lstur_ctx::Pending();   // consumed, or stale: either way the other set is free again
  if (c->staged_hit) c->prefetch_hits++;
  int rc = LSTUR_OK;
  if (c->staged_hit) use_set(c, c->cur_set ^ 1);
  else rc = stage_titles(c, hist, cands, n, d.K1, ids_form, s);
  if (rc) return rc;
  Workspace& w = c->ws;
  c->sfuse.want = c->opt.score_fuse;
  c->sfuse.backward = backward;
  c->sfuse.inv_n = grad_scale / (float)n;
  c->sfuse.done = false;
  PROF("forward_total", run_forward(c, user, n, d.K1, training, step, s));
  c->sfuse.want = false;
  c->staged_hit = false;
  run_listwise_score(c, n, d.K1, grad_scale / (float)n, backward, nullptr, s);
  if (stats && !backward) launch_mean2(w.loss_i, w.correct_i, n, stats, s);
  if (backward) {
    ProfScope _ps(c, "backward_total", s);
    if ((rc = run_backward(c, n, step, s, stats)) != LSTUR_OK) return rc;
  }
  return check_launch(backward ? "forward_backward" : "evaluate");
}

// scorer of the pointwise heads on (uservec [n,U], doc vectors of the n candidates); fills logits/probs (+ loss,
// + gradients wrt both inputs and the head's parameters when `backward`)
void run_point_head(lstur_ctx* c, const float* labels, int n, float scale, bool with_loss, bool backward, hipStream_t s) {
  const Dims& d = c->d;
  Workspace& w = c->ws;
  const float* dvec = w.docvec + (size_t)n * d.W * d.U;
  float* ddvec = w.ddocvec + (size_t)n * d.W * d.U;
  const float gain = c->cfg.gain;
  const int K = c->cfg.loss_negatives;
  if (d.head == LSTUR_HEAD_POINT_DOT) {
    launch_point_dot(w.uservec, dvec, labels, w.logits, w.probs, w.loss_i, w.correct_i, backward ? w.du : nullptr,
                     backward ? ddvec : nullptr, n, d.U, gain, K, scale, s);
    return;
  }
  const int wc = d.Uu + d.U;
  launch_concat2(w.uservec, d.Uu, dvec, d.U, w.scat, n, s);
  launch_gemm(w.scat, wc, 0, c->P("score_w1"), d.U, 0, w.spre, d.U, c->P("score_b1"), n, d.U, wc, 1.f, w.gemm_scratch, s);
  launch_point_mlp_out(w.spre, c->P("score_w2"), c->P("score_b2"), labels, w.logits, w.probs,
                       with_loss ? w.loss_i : nullptr, w.correct_i, backward ? w.sds : nullptr,
                       backward ? w.sdhid : nullptr, n, d.U, gain, K, scale, s);
  if (!backward) return;
  launch_point_w2_grad(w.spre, w.sds, c->G("score_w2"), c->G("score_b2"), n, d.U, s);
  launch_gemm(w.scat, wc, 1, w.sdhid, d.U, 0, c->G("score_w1"), d.U, nullptr, wc, d.U, n, 1.f, w.gemm_scratch, s);
  launch_colsum2(w.sdhid, n, d.U, d.U, c->G("score_b1"), 1.f, w.colsum_scratch, s);
  launch_gemm(w.sdhid, d.U, 0, c->P("score_w1"), d.U, 1, w.sdcat, wc, nullptr, n, wc, d.U, 1.f, w.gemm_scratch, s);
  launch_split2(w.sdcat, d.Uu, d.U, w.du, ddvec, n, s);
}

int point_impl(lstur_ctx* c, const int32_t* user, const int32_t* hist, const int32_t* cand, const float* labels, int n,
               uint32_t step, float grad_scale, float* stats, hipStream_t s, bool ids_form, bool backward,
               bool training) {
  if (!c || !user || !hist || !cand || !labels) return fail(LSTUR_ERR_ARG, "NULL argument");
  const Dims& d = c->d;
  if (d.head == LSTUR_HEAD_SOFTMAX) return fail(LSTUR_ERR_STATE, "context was created with the listwise head");
  if (n < 1 || n > d.B) return fail(LSTUR_ERR_ARG, "n must be in [1, max_batch]");
  c->ws.kbits_valid = false;
  int rc = stage_titles(c, hist, cand, n, 1, ids_form, s);
  if (rc) return rc;
  Workspace& w = c->ws;
  PROF("forward_total", run_forward(c, user, n, 1, training, step, s));
  run_point_head(c, labels, n, grad_scale / (float)n, true, backward, s);
  if (stats && !backward) launch_mean2(w.loss_i, w.correct_i, n, stats, s);
  if (backward) {
    ProfScope _ps(c, "backward_total", s);
    if ((rc = run_backward(c, n, step, s, stats)) != LSTUR_OK) return rc;
  }
  return check_launch(backward ? "point_forward_backward" : "point_evaluate");
}

int predict_impl(lstur_ctx* c, const int32_t* user, const int32_t* hist, const int32_t* cand, int n, float* scores,
                 hipStream_t s, bool ids_form) {
  if (!c || !user || !hist || !cand || !scores) return fail(LSTUR_ERR_ARG, "NULL argument");
  const Dims& d = c->d;
  int cap = d.nt_max / (d.W + 1);
  if (n < 1 || n > cap) return fail(LSTUR_ERR_ARG, "n exceeds max_predict_rows");
  c->ws.kbits_valid = false;
  int rc = stage_titles(c, hist, cand, n, 1, ids_form, s);
  if (rc) return rc;
  run_forward(c, user, n, 1, false, 0, s);
  if (d.head == LSTUR_HEAD_POINT_MLP) {   // model.predict of the pointwise model = the scorer's sigmoid output
    Workspace& w = c->ws;
    launch_concat2(w.uservec, d.Uu, w.docvec + (size_t)n * d.W * d.U, d.U, w.scat, n, s);
    launch_gemm(w.scat, d.Uu + d.U, 0, c->P("score_w1"), d.U, 0, w.spre, d.U, c->P("score_b1"), n, d.U, d.Uu + d.U, 1.f,
                w.gemm_scratch, s);
    launch_point_mlp_out(w.spre, c->P("score_w2"), c->P("score_b2"), nullptr, w.logits, scores, nullptr, nullptr,
                         nullptr, nullptr, n, d.U, 1.f, 1, 0.f, s);
  } else {
    run_listwise_score(c, n, 1, 0.f, false, scores, s);
  }
  return check_launch("predict");
}

}  // namespace

// =================================================================================================
extern "C" {

const char* lstur_last_error(void) { return g_err.c_str(); }
const char* lstur_version(void) { return "lstur-hip 0.3 (gfx950)"; }
int lstur_abi_version(void) { return LSTUR_ABI_VERSION; }

int lstur_query_sizes(const lstur_cfg* cfg, lstur_sizes* out) {
  Dims d;
  std::string why;
  if (!out) return fail(LSTUR_ERR_ARG, "out is NULL");
  if (!make_dims(cfg, &d, &why)) return fail(LSTUR_ERR_ARG, why);
  size_t total;
  param_layout(d, &total);
  Workspace w;
  carve(d, nullptr, &w);
  size_t ws_bytes = w.bytes;
  out->word_emb = (size_t)d.V * d.D;
  out->dense = total;
  out->user_emb = (size_t)d.Nu * d.Ue;
  out->workspace_bytes = ws_bytes;
  out->user_emb_dim = d.Ue;
  out->gru_dim = d.H;
  out->max_predict_rows = d.nt_max / (d.W + 1);
  return LSTUR_OK;
}

int lstur_param(const lstur_cfg* cfg, const char* name, size_t* offset, size_t* count) {
  Dims d;
  std::string why;
  if (!name) return fail(LSTUR_ERR_ARG, "name is NULL");
  if (!make_dims(cfg, &d, &why)) return fail(LSTUR_ERR_ARG, why);
  size_t total;
  for (auto& s : param_layout(d, &total))
    if (!strcmp(s.name, name)) {
      if (offset) *offset = s.offset;
      if (count) *count = s.count;
      return LSTUR_OK;
    }
  return fail(LSTUR_ERR_ARG, std::string("unknown parameter ") + name);
}

const char* lstur_param_name(const lstur_cfg* cfg, int i) {
  Dims d;
  std::string why;
  if (!make_dims(cfg, &d, &why)) return nullptr;
  size_t total;
  auto v = param_layout(d, &total);
  return (i >= 0 && i < (int)v.size()) ? v[i].name : nullptr;
}

int lstur_create(const lstur_cfg* cfg, const lstur_buffers* buf, lstur_ctx** out) {
  Dims d;
  std::string why;
  if (!buf || !out) return fail(LSTUR_ERR_ARG, "NULL argument");
  if (!make_dims(cfg, &d, &why)) return fail(LSTUR_ERR_ARG, why);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(LSTUR_ERR_NOGPU, "no HIP device visible: the LSTUR path has no CPU fallback");
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
    return fail(LSTUR_ERR_HIP, "hipGetDeviceProperties failed");
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(LSTUR_ERR_NOGPU, std::string("kernels are built for gfx950 only, device is ") + prop.gcnArchName);
  if (!buf->word_emb || !buf->dense_param || !buf->dense_grad || !buf->dense_m || !buf->dense_v || !buf->user_emb ||
      !buf->user_m || !buf->user_v || !buf->workspace)
    return fail(LSTUR_ERR_ARG, "every buffer must be bound");
  if (d.word_trainable && (!buf->word_grad || !buf->word_m || !buf->word_v))
    return fail(LSTUR_ERR_ARG, "word_trainable needs word_grad / word_m / word_v");
  if (two_tables(d) && (!buf->user_emb2 || !buf->user_m2 || !buf->user_v2))
    return fail(LSTUR_ERR_ARG, "iigru / iicgru need user_emb2 / user_m2 / user_v2");
  if (((uintptr_t)buf->workspace & 255) || ((uintptr_t)buf->word_emb & 15) || ((uintptr_t)buf->dense_param & 15) ||
      ((uintptr_t)buf->dense_grad & 15) || ((uintptr_t)buf->user_emb & 15))
    return fail(LSTUR_ERR_ARG, "buffers must be 16-byte (workspace 256-byte) aligned");
  lstur_ctx* c = new lstur_ctx();
  c->opt = Opts::read();
  c->cfg = *cfg;
  c->d = d;
  c->buf = *buf;
  c->layout = param_layout(d, &c->dense_total);
  if (d.enc == LSTUR_ENC_MHSA && c->G("qkv_b") != c->G("qkv_w") + (size_t)d.D * 3 * d.F) {
    delete c;
    return fail(LSTUR_ERR_ARG, "mhsa: qkv_b must follow qkv_w in the bucket");
  }
  carve(d, buf->workspace, &c->ws);
  size_t need = c->ws.bytes;
  if (need > buf->workspace_bytes) {
    delete c;
    return fail(LSTUR_ERR_ARG, "workspace too small (see lstur_query_sizes)");
  }
  *out = c;
  return LSTUR_OK;
}

int lstur_destroy(lstur_ctx* ctx) {
  if (!ctx) return LSTUR_OK;
  // (pooled streams outlive the context; everything this context enqueued on them is drained before its events go)
  if (ctx->side_p) {
    (void)hipStreamSynchronize(ctx->side_p);
    if (ctx->side_t && ctx->side_t != ctx->side_p) (void)hipStreamSynchronize(ctx->side_t);
    (void)hipEventDestroy(ctx->ev_p0);
    (void)hipEventDestroy(ctx->ev_p1);
    (void)hipEventDestroy(ctx->ev_t0);
    (void)hipEventDestroy(ctx->ev_t1);
  }
  if (ctx->ev_f0) (void)hipEventDestroy(ctx->ev_f0);
  if (ctx->side_w) {
    (void)hipStreamSynchronize(ctx->side_w);
    (void)hipEventDestroy(ctx->ev_w1);
    (void)hipEventDestroy(ctx->ev_w2);
    (void)hipEventDestroy(ctx->ev_w3);
  }
  if (ctx->ev_c0) (void)hipEventDestroy(ctx->ev_c0);
  if (ctx->ev_c1) (void)hipEventDestroy(ctx->ev_c1);
  for (auto& k : ctx->prof)
    for (size_t i = 0; i < k.start.size(); ++i) {
      (void)hipEventDestroy(k.start[i]);
      (void)hipEventDestroy(k.stop[i]);
    }
  delete ctx;
  return LSTUR_OK;
}

int lstur_set_title_table(lstur_ctx* ctx, const int32_t* titles, int32_t n_docs) {
  if (!ctx || !titles || n_docs < 1) return fail(LSTUR_ERR_ARG, "bad title table");
  ctx->title_table = titles;
  ctx->n_docs = n_docs;
  return LSTUR_OK;
}

int lstur_forward_backward(lstur_ctx* ctx, const int32_t* user, const int32_t* hist, const int32_t* cands, int32_t n,
                           uint32_t step, float grad_scale, float* stats, lstur_stream stream) {
  return train_impl(ctx, user, hist, cands, n, step, grad_scale, stats, (hipStream_t)stream, false, true, true);
}
int lstur_forward_backward_ids(lstur_ctx* ctx, const int32_t* user, const int32_t* hist_ids, const int32_t* cand_ids,
                               int32_t n, uint32_t step, float grad_scale, float* stats, lstur_stream stream) {
  return train_impl(ctx, user, hist_ids, cand_ids, n, step, grad_scale, stats, (hipStream_t)stream, true, true, true);
}
int lstur_prefetch_ids(lstur_ctx* ctx, const int32_t* hist_ids, const int32_t* cand_ids, int32_t n, uint32_t step) {
  if (!ctx || !hist_ids || !cand_ids) return fail(LSTUR_ERR_ARG, "prefetch: NULL argument");
  if (n < 1 || n > ctx->d.B) return fail(LSTUR_ERR_ARG, "prefetch: n must be in [1, max_batch]");
  if (!ctx->title_table) return fail(LSTUR_ERR_STATE, "no title table bound (lstur_set_title_table)");
  ctx->pend = lstur_ctx::Pending();      // (replaces an announcement that was never staged; a staged batch stays `ready`)
  if (ctx->ws.set[1].ids == ctx->ws.set[0].ids) return LSTUR_OK;      // this configuration stages inline: a no-op
  ctx->pend.hist = hist_ids;
  ctx->pend.cands = cand_ids;
  ctx->pend.users = ctx->next_users;     // (lstur_prefetch_rows, same batch)
  ctx->next_users = nullptr;
  ctx->pend.n = n;
  ctx->pend.step = step;
  ctx->pend.valid = true;
  return LSTUR_OK;
}

int lstur_prefetch_rows(lstur_ctx* ctx, const int32_t* user) {
  if (!ctx) return fail(LSTUR_ERR_ARG, "ctx is NULL");
  ctx->next_users = user;
  return LSTUR_OK;
}

int32_t lstur_prefetch_hits(lstur_ctx* ctx) { return ctx ? ctx->prefetch_hits : -1; }

int lstur_evaluate(lstur_ctx* ctx, const int32_t* user, const int32_t* hist, const int32_t* cands, int32_t n,
                   float* stats, lstur_stream stream) {
  return train_impl(ctx, user, hist, cands, n, 0, 1.f, stats, (hipStream_t)stream, false, false, false);
}

int lstur_point_forward_backward(lstur_ctx* ctx, const int32_t* user, const int32_t* hist, const int32_t* cand,
                                 const float* labels, int32_t n, uint32_t step, float grad_scale, float* stats,
                                 lstur_stream stream) {
  return point_impl(ctx, user, hist, cand, labels, n, step, grad_scale, stats, (hipStream_t)stream, false, true, true);
}
int lstur_point_forward_backward_ids(lstur_ctx* ctx, const int32_t* user, const int32_t* hist_ids,
                                     const int32_t* cand_ids, const float* labels, int32_t n, uint32_t step,
                                     float grad_scale, float* stats, lstur_stream stream) {
  return point_impl(ctx, user, hist_ids, cand_ids, labels, n, step, grad_scale, stats, (hipStream_t)stream, true, true,
                    true);
}
int lstur_point_evaluate(lstur_ctx* ctx, const int32_t* user, const int32_t* hist, const int32_t* cand,
                         const float* labels, int32_t n, float* stats, lstur_stream stream) {
  return point_impl(ctx, user, hist, cand, labels, n, 0, 1.f, stats, (hipStream_t)stream, false, false, false);
}

int lstur_user_grad(lstur_ctx* ctx, const int32_t** rows, const float** grads, int32_t* n) {
  if (!ctx) return fail(LSTUR_ERR_ARG, "ctx is NULL");
  if (rows) *rows = ctx->ws.user_rows;
  if (grads) *grads = ctx->ws.duser;
  if (n) *n = ctx->last_n;
  return LSTUR_OK;
}

size_t lstur_dp_tail_floats(const lstur_cfg* cfg, int32_t world) {
  Dims d;
  std::string why;
  if (world < 1 || !make_dims(cfg, &d, &why)) return 0;
  return (size_t)world * d.B * (1 + (two_tables(d) ? 2 : 1) * d.Ue);   // iigru / iicgru: both tables' row gradients
}
int lstur_dp_stage(lstur_ctx* ctx, float* tail, int32_t rank, int32_t world, lstur_stream stream) {
  if (!ctx || !tail || world < 1 || rank < 0 || rank >= world) return fail(LSTUR_ERR_ARG, "dp stage: bad argument");
  const Dims& d = ctx->d;
  if ((size_t)world * d.B > (size_t)d.rows_cap) return fail(LSTUR_ERR_ARG, "dp stage: world size above 8");
  if (d.Nu >= (1 << 24)) return fail(LSTUR_ERR_ARG, "dp stage: row numbers above 2^24 are not exact as float");
  launch_dp_stage(ctx->ws.user_rows, ctx->ws.duser, two_tables(d) ? ctx->ws.duser2 : nullptr, ctx->last_n, d.B, d.Ue, tail,
                  rank, world, (hipStream_t)stream);
  return check_launch("dp_stage");
}
int lstur_dp_attach(lstur_ctx* ctx, float* tail, int32_t rank, int32_t world) {
  if (!ctx) return fail(LSTUR_ERR_ARG, "dp attach: ctx is NULL");
  if (!tail) {
    ctx->dp_tail = nullptr;
    return LSTUR_OK;
  }
  const Dims& d = ctx->d;
  if (world < 1 || rank < 0 || rank >= world) return fail(LSTUR_ERR_ARG, "dp attach: bad rank / world");
  if ((size_t)world * d.B > (size_t)d.rows_cap) return fail(LSTUR_ERR_ARG, "dp attach: world size above 8");
  if (d.Nu >= (1 << 24)) return fail(LSTUR_ERR_ARG, "dp attach: row numbers above 2^24 are not exact as float");
  ctx->dp_tail = tail;
  ctx->dp_rank = rank;
  ctx->dp_world = world;
  return LSTUR_OK;
}
int lstur_dp_unpack(lstur_ctx* ctx, const float* tail, int32_t world, int32_t n, const int32_t** rows, const float** grads,
                    lstur_stream stream) {
  if (!ctx || !tail || !rows || !grads || world < 1 || n < 0) return fail(LSTUR_ERR_ARG, "dp unpack: bad argument");
  const Dims& d = ctx->d;
  if (n > d.B || (size_t)world * n > (size_t)d.rows_cap) return fail(LSTUR_ERR_ARG, "dp unpack: too many rows");
  launch_dp_unpack(tail, n, d.B, d.Ue, world, ctx->ws.dp_rows, ctx->ws.dp_grads, two_tables(d) ? ctx->ws.dp_grads2 : nullptr,
                   (hipStream_t)stream);
  *rows = ctx->ws.dp_rows;
  *grads = ctx->ws.dp_grads;
  return check_launch("dp_unpack");
}

// Keras Adam over everything trainable.  rows == NULL: this process's own rows of the last step.  skip_table: the fused
// step already swept the user table(s) on its side stream (joined here).  Leaves valid weight images for the next forward.
// rows_f (lstur_adam_step_dp, full batches): the reduced tail read in place -- row numbers as floats (the first launch leaves
// them as int32 in ws.dp_rows = `rows`), row_grads / grads2_in point into the tail
static int adam_impl(lstur_ctx* ctx, float lr, int32_t t, const int32_t* rows, const float* row_grads, int32_t n_rows,
                     hipStream_t s, bool skip_table, const float* rows_f = nullptr, const float* grads2_in = nullptr) {
  const Dims& d = ctx->d;
  const lstur_cfg& c = ctx->cfg;
  const float* row_grads2 = nullptr;   // second table (iigru / iicgru): own rows -> ws.duser2, exchanged rows -> ws.dp_grads2
  const bool exchanged = rows != nullptr;   // the data-parallel step: rows and gradients of every rank's batch
  if (rows_f) row_grads2 = grads2_in;
  else if (rows && two_tables(d)) {
    if (rows != ctx->ws.dp_rows || row_grads != ctx->ws.dp_grads)
      return fail(LSTUR_ERR_ARG, "iigru: external user rows must come from lstur_dp_unpack (it carries both tables' gradients)");
    row_grads2 = ctx->ws.dp_grads2;
  }
  if (!rows) {
    rows = ctx->ws.user_rows;
    row_grads = ctx->ws.duser;
    n_rows = ctx->last_n;
  }
  if (n_rows < 0 || n_rows > d.rows_cap) return fail(LSTUR_ERR_ARG, "too many sparse rows for one adam step");
  const float lr_t = lr * (sqrtf(1.0f - powf(c.beta2, (float)t)) / (1.0f - powf(c.beta1, (float)t)));
  defer_flush(ctx, s);   // (a deferred sweep no forward picked up, or still in flight: the slot map must be clear)
  lazy_init(ctx, s);
  // (a prefetch-time catch-up of the next batch's rows ran on the parameter-gradient stream: the backward joined it)
  lazy_begin_iteration(ctx, t, s);   // (large tables; a no-op when the fused step already did it)
  const bool no_user = !has_user_table(d);
  // The next forward's weight images ride the dense Adam launch when they exist (packed whole once) and both fast kernels
  // are in use: two 5-us launches less between the step's reduction and the next conv forward.
  bool packed = false;
  ImagePlan pl;
  {
    const int kq = gru4_supported(d.H) ? gru4_pack_kq(d.H) : 0;
    const float grow = adam_growth_bound(c) * fabsf(lr_t);
    if (ctx->images_valid && ctx->opt.adam_pack && d.np && kq > 0 && ctx->ws.gru_up && isfinite(grow)) {
      int Kp = 0, Np = 0;
      convx_w_dims(d.D, d.k, d.F, d.np, &Kp, &Np);
      pl.conv_off = (size_t)(ctx->P("conv_w") - ctx->buf.dense_param);
      pl.conv_cnt = (size_t)d.k * d.D * d.F;
      pl.D = d.D; pl.Dp = convx_dp(d.D); pl.F = d.F; pl.Np = Np;
      pl.w3 = ctx->ws.conv_w3;
      pl.w3_part = (size_t)Kp * Np;
      pl.np = d.np; pl.cs = ctx->ws.cs; pl.seq = ctx->pack_seq++; pl.grow = grow;
      pl.gru_off = (size_t)(ctx->P("gru_u") - ctx->buf.dense_param);
      pl.H = d.H; pl.KQ = kq;
      pl.gimg = ctx->ws.gru_up;
      pl.gru_f16 = ctx->opt.gru_f16 ? 1 : 0; pl.gru_seq = ctx->gru_seq++; pl.ggrow = grow;
      packed = true;
    }
  }
  if (no_user || skip_table) {
    if (packed)
      launch_adam_flat_pack(ctx->buf.dense_param, ctx->buf.dense_grad, ctx->buf.dense_m, ctx->buf.dense_v, ctx->dense_total,
                            lr_t, c.beta1, c.beta2, c.eps, pl, s);
    else
      launch_adam_flat(ctx->buf.dense_param, ctx->buf.dense_grad, ctx->buf.dense_m, ctx->buf.dense_v, ctx->dense_total,
                       lr_t, c.beta1, c.beta2, c.eps, s);
  } else {   // (the split / data-parallel sequence: + the duplicate-row combine of the row gradients in the same launch)
    launch_adam_flat_combine(ctx->buf.dense_param, ctx->buf.dense_grad, ctx->buf.dense_m, ctx->buf.dense_v,
                             ctx->dense_total, lr_t, c.beta1, c.beta2, c.eps, rows, row_grads, n_rows, d.Ue,
                             ctx->ws.gsum, ctx->ws.slot, s, packed ? &pl : nullptr, rows_f, rows_f ? ctx->ws.dp_rows : nullptr);
  }
  // the next forward's weight images, behind the dense update (in front of the next conv forward they would be a bubble:
  // with the staging prefetched nothing else is left to hide them under)
  if (!packed) pack_weight_images(ctx, s, !ctx->images_valid, lr_t);
  ctx->images_valid = true;
  if (d.word_trainable)   // dense Keras-Adam sweep of the word table (TF-1 densifies the IndexedSlices gradient)
    launch_adam_flat(ctx->buf.word_emb, ctx->buf.word_grad, ctx->buf.word_m, ctx->buf.word_v, (size_t)d.V * d.D, lr_t,
                     c.beta1, c.beta2, c.eps, s);
  if (!no_user && !skip_table) {   // nigru / niavg: the user embedding is not part of the graph (paper.py:627-630)
    if (d.lazy) lazy_rows_step(ctx, rows, row_grads2, n_rows, t, lr_t, s);
    else if (exchanged && n_rows > 0 && defer_wanted(ctx)) {
      // the union's rows now (combined gradients: gsum / slot from the launch above), every other row one forward later
      launch_adam_rows(ctx->buf.user_emb, ctx->buf.user_m, ctx->buf.user_v, rows, ctx->ws.slot, ctx->ws.gsum, n_rows, d.Ue, lr_t,
                       c.beta1, c.beta2, c.eps, s);
      if (two_tables(d)) {
        launch_user_grad_combine(rows, row_grads2 ? row_grads2 : ctx->ws.duser2, n_rows, d.Ue, ctx->ws.first, ctx->ws.gsum2,
                                 ctx->ws.slot, s);
        launch_adam_rows(ctx->buf.user_emb2, ctx->buf.user_m2, ctx->buf.user_v2, rows, ctx->ws.slot, ctx->ws.gsum2, n_rows, d.Ue,
                         lr_t, c.beta1, c.beta2, c.eps, s);
      }
      (void)hipMemcpyAsync(ctx->ws.pend_rows, rows, (size_t)n_rows * sizeof(int32_t), hipMemcpyDeviceToDevice, s);
      ctx->defer.valid = true;
      ctx->defer.lr_t = lr_t;
      ctx->defer.n_rows = n_rows;
    } else adam_user_tables(ctx, rows, row_grads, n_rows, lr_t, false, s, row_grads2);
  }
  if (skip_table) (void)hipStreamWaitEvent(s, ctx->ev_t1, 0);
  if (d.lazy) ctx->lazy_t = t;
  return check_launch("adam_step");
}

int lstur_adam_step(lstur_ctx* ctx, float lr, int32_t t, const int32_t* rows, const float* row_grads, int32_t n_rows,
                    lstur_stream stream) {
  if (!ctx || t < 1) return fail(LSTUR_ERR_ARG, "bad adam step");
  return adam_impl(ctx, lr, t, rows, row_grads, n_rows, (hipStream_t)stream, false);
}

int lstur_adam_step_dp(lstur_ctx* ctx, float lr, int32_t t, const float* tail, int32_t world, int32_t n, lstur_stream stream) {
  if (!ctx || t < 1 || !tail || world < 1 || n < 0) return fail(LSTUR_ERR_ARG, "bad data-parallel adam step");
  const Dims& d = ctx->d;
  if (n > d.B || (size_t)world * d.B > (size_t)d.rows_cap) return fail(LSTUR_ERR_ARG, "adam step (dp): too many rows");
  const size_t WB = (size_t)world * d.B;
  if (n == d.B && has_user_table(d))   // full batches: the reduced tail IS the row list and the row-gradient matrix
    return adam_impl(ctx, lr, t, ctx->ws.dp_rows, tail + WB, world * n, (hipStream_t)stream, false, tail,
                     two_tables(d) ? tail + WB * (1 + d.Ue) : nullptr);
  launch_dp_unpack(tail, n, d.B, d.Ue, world, ctx->ws.dp_rows, ctx->ws.dp_grads, two_tables(d) ? ctx->ws.dp_grads2 : nullptr,
                   (hipStream_t)stream);
  return adam_impl(ctx, lr, t, ctx->ws.dp_rows, ctx->ws.dp_grads, world * n, (hipStream_t)stream, false);
}

int lstur_flush_tables(lstur_ctx* ctx, lstur_stream stream) {
  if (!ctx) return fail(LSTUR_ERR_ARG, "ctx is NULL");
  lazy_flush_all(ctx, (hipStream_t)stream);
  defer_flush(ctx, (hipStream_t)stream);
  return check_launch("flush_tables");
}
int lstur_lazy_adam(lstur_ctx* ctx) { return (ctx && ctx->d.lazy) ? 1 : 0; }

int lstur_params_changed(lstur_ctx* ctx) {
  if (!ctx) return fail(LSTUR_ERR_ARG, "ctx is NULL");
  ctx->images_valid = false;
  ctx->escale_valid = false;            // (f16x2: the embeddings' scale follows the word table)
  ctx->ready = lstur_ctx::Pending();    // a staged batch holds embeddings of the old word table
  return LSTUR_OK;
}

// forward + backward + Adam of ONE process's step in one call (no gradient exchange in between)
static int train_step_impl(lstur_ctx* ctx, const int32_t* user, const int32_t* hist, const int32_t* cands,
                           const float* labels, int32_t n, uint32_t step, float lr, int32_t t, float* stats, hipStream_t s,
                           bool ids_form) {
  if (!ctx || t < 1) return fail(LSTUR_ERR_ARG, "bad train step");
  const lstur_cfg& c = ctx->cfg;
  ctx->fused.on = ctx->opt.fused_adam;
  ctx->fused.swept = ctx->fused.table_done = false;
  ctx->fused.lr_t = lr * (sqrtf(1.0f - powf(c.beta2, (float)t)) / (1.0f - powf(c.beta1, (float)t)));
  ctx->fused.t = t;
  defer_flush(ctx, s);
  lazy_begin_iteration(ctx, t, s);
  int rc;
  if (ctx->d.head == LSTUR_HEAD_SOFTMAX)
    rc = train_impl(ctx, user, hist, cands, n, step, 1.f, stats, s, ids_form, true, true);
  else
    rc = point_impl(ctx, user, hist, cands, labels, n, step, 1.f, stats, s, ids_form, true, true);
  const bool done = ctx->fused.table_done;
  if (ctx->fused.swept && !done) {   // (not reached: every path that sweeps runs the backward; keep the table whole anyway)
    ctx->fused.on = false;
    return fail(LSTUR_ERR_STATE, "fused step: the user-table sweep was left half done");
  }
  ctx->fused.on = false;
  ctx->fused.swept = ctx->fused.table_done = false;
  if (rc) {
    if (done) (void)hipStreamWaitEvent(s, ctx->ev_t1, 0);
    return rc;
  }
  return adam_impl(ctx, lr, t, nullptr, nullptr, 0, s, done);
}
int lstur_train_step(lstur_ctx* ctx, const int32_t* user, const int32_t* hist, const int32_t* cands, const float* labels,
                     int32_t n, uint32_t step, float lr, int32_t t, float* stats, lstur_stream stream) {
  return train_step_impl(ctx, user, hist, cands, labels, n, step, lr, t, stats, (hipStream_t)stream, false);
}
int lstur_train_step_ids(lstur_ctx* ctx, const int32_t* user, const int32_t* hist_ids, const int32_t* cand_ids,
                         const float* labels, int32_t n, uint32_t step, float lr, int32_t t, float* stats,
                         lstur_stream stream) {
  return train_step_impl(ctx, user, hist_ids, cand_ids, labels, n, step, lr, t, stats, (hipStream_t)stream, true);
}

int lstur_predict(lstur_ctx* ctx, const int32_t* user, const int32_t* hist, const int32_t* cand, int32_t n,
                  float* scores, lstur_stream stream) {
  return predict_impl(ctx, user, hist, cand, n, scores, (hipStream_t)stream, false);
}
int lstur_predict_ids(lstur_ctx* ctx, const int32_t* user, const int32_t* hist_ids, const int32_t* cand_ids, int32_t n,
                      float* scores, lstur_stream stream) {
  return predict_impl(ctx, user, hist_ids, cand_ids, n, scores, (hipStream_t)stream, true);
}

int lstur_profile(lstur_ctx* ctx, int enable) {
  if (!ctx) return fail(LSTUR_ERR_ARG, "ctx is NULL");
  ctx->prof_on = enable != 0;
  ctx->prof_roofline_only = enable == 2;
  for (auto& k : ctx->prof) k.used = 0;
  return LSTUR_OK;
}

int lstur_profile_read(lstur_ctx* ctx, int index, const char** name, float* total_ms, int32_t* launches) {
  if (!ctx || index < 0) return fail(LSTUR_ERR_ARG, "bad profile index");
  if (index >= (int)ctx->prof.size()) return 1;  // past the end
  ProfKernel& k = ctx->prof[index];
  float tot = 0.f;
  for (size_t i = 0; i < k.used; ++i) {
    float ms = 0.f;
    if (hipEventSynchronize(k.stop[i]) != hipSuccess || hipEventElapsedTime(&ms, k.start[i], k.stop[i]) != hipSuccess)
      return fail(LSTUR_ERR_HIP, "hipEventElapsedTime failed");
    tot += ms;
  }
  if (name) *name = k.name;
  if (total_ms) *total_ms = tot;
  if (launches) *launches = (int32_t)k.used;
  return LSTUR_OK;
}

int lstur_conv_scheme(lstur_ctx* ctx) {
  if (!ctx) return fail(LSTUR_ERR_ARG, "ctx is NULL");
  if (ctx->d.enc == LSTUR_ENC_MHSA) return 0;
  return ctx->d.np == 2 ? 3 : ctx->d.np == 3 ? 1 : 2;
}

int lstur_peek(lstur_ctx* ctx, const char* name, const float** ptr, size_t* count) {
  if (!ctx || !name || !ptr) return fail(LSTUR_ERR_ARG, "NULL argument");
  const Dims& d = ctx->d;
  const Workspace& w = ctx->ws;
  const size_t n = ctx->last_n, k1 = ctx->last_k1, nt = n * (d.W + k1);
  struct { const char* nm; const float* p; size_t cnt; } tab[] = {
      {"doc_vec", w.docvec, nt * d.U},      {"user_vec", w.uservec, n * d.U},
      {"logits", w.logits, n * k1},         {"probs", w.probs, n * k1},
      {"conv_out", w.conv, nt * d.T * d.F}, {"att_w", w.att_wt, nt * d.T},
      {"emb", w.emb, nt * d.Tp * d.D},      {"pooled", w.pooled, nt * d.F},
      {"loss_i", w.loss_i, n},              {"user_grad", w.duser, n * d.Ue},
      {"dconv", w.dconv, nt * d.T * d.F},
      {"duser2", w.duser2, two_tables(d) ? n * d.Ue : 0},
      {"h_last", (d.arch == LSTUR_ARCH_IGRU || d.arch == LSTUR_ARCH_NIGRU) ? w.uservec : w.hlast, n * d.H},
      {"dgates", w.dgates, n * d.W * 3 * d.H}, {"ddoc_vec", w.ddocvec, nt * d.U},
  };
  if (!strcmp(name, "emb") && ctx->last_usex)
    return fail(LSTUR_ERR_STATE, "emb: the split-operand conv path keeps the embeddings pre-split (Ex); run with LSTUR_CONV_F32=1");
  if (!strcmp(name, "emb") && d.enc == LSTUR_ENC_MHSA)
    return fail(LSTUR_ERR_STATE, "emb: the mhsa encoder keeps the gathered embeddings as bf16 rows");
  for (auto& e : tab)
    if (!strcmp(e.nm, name)) {
      *ptr = e.p;
      if (count) *count = e.cnt;
      return LSTUR_OK;
    }
  return fail(LSTUR_ERR_ARG, std::string("unknown intermediate ") + name);
}

// ---- per-kernel entry points ---------------------------------------------------------------------
int lstur_gather_f32(const float* table, const int32_t* ids, float* out, int32_t n_titles, int32_t title_len,
                     int32_t dim, int32_t pad_left, int32_t pad_right, float drop, uint32_t key, lstur_stream stream) {
  if (!table || !ids || !out || dim % 4 || n_titles < 0) return fail(LSTUR_ERR_ARG, "gather: bad argument (dim % 4)");
  launch_gather(table, ids, out, n_titles, title_len, dim, pad_left, pad_right, drop, key, (hipStream_t)stream);
  return check_launch("gather");
}
int lstur_valid_tokens(const int32_t* ids, int32_t n_tokens, int32_t* vidx, int32_t* count, int32_t* scratch,
                       lstur_stream stream) {
  if (!ids || !vidx || !count || !scratch) return fail(LSTUR_ERR_ARG, "valid_tokens: NULL");
  launch_compact_valid(ids, n_tokens, vidx, count, scratch, (hipStream_t)stream);
  return check_launch("valid_tokens");
}
size_t lstur_valid_tokens_scratch(int32_t n_tokens) { return compact_scratch_ints(n_tokens) * sizeof(int32_t); }
int lstur_conv1d_relu_mask_fwd(const float* emb_padded, const float* w_packed, const float* bias, const int32_t* ids,
                               float* out, int32_t n_titles, int32_t title_len, int32_t dim, int32_t ksize,
                               int32_t filters, const int32_t* vidx, const int32_t* vcount, lstur_stream stream) {
  if (!emb_padded || !w_packed || !bias || !ids || !out || dim % 4) return fail(LSTUR_ERR_ARG, "conv fwd: bad argument");
  launch_conv_fwd(emb_padded, w_packed, bias, ids, out, n_titles, title_len, dim, ksize, filters, vidx, vcount,
                  (hipStream_t)stream);
  return check_launch("conv_fwd");
}
size_t lstur_conv1d_wgrad_scratch(int32_t dim, int32_t ksize, int32_t filters) {
  return conv_wgrad_scratch_bytes(dim, ksize, filters);
}
int lstur_conv1d_wgrad(const float* emb_padded, const float* dout, float* dw, void* scratch, int32_t n_titles,
                       int32_t title_len, int32_t dim, int32_t ksize, int32_t filters, const int32_t* vidx,
                       const int32_t* vcount, lstur_stream stream) {
  if (!emb_padded || !dout || !dw || !scratch || dim % 4) return fail(LSTUR_ERR_ARG, "conv wgrad: bad argument");
  launch_conv_wgrad(emb_padded, dout, dw, scratch, n_titles, title_len, dim, ksize, filters, vidx, vcount,
                    (hipStream_t)stream);
  return check_launch("conv_wgrad");
}
size_t lstur_pack_b_floats(int32_t K, int32_t N) { return pack_b_floats(K, N); }
int lstur_pack_b(const float* B, int32_t ldb, int32_t K, int32_t N, float* packed, lstur_stream stream) {
  if (!B || !packed) return fail(LSTUR_ERR_ARG, "pack_b: NULL");
  launch_pack_b(B, ldb, K, N, packed, (hipStream_t)stream);
  return check_launch("pack_b");
}
int lstur_pack_bt(const float* Bt, int32_t ldb, int32_t K, int32_t N, float* packed, lstur_stream stream) {
  if (!Bt || !packed) return fail(LSTUR_ERR_ARG, "pack_bt: NULL");
  launch_pack_b_t(Bt, ldb, K, N, packed, (hipStream_t)stream);
  return check_launch("pack_bt");
}
int lstur_attpool_fwd(const float* conv_out, const int32_t* ids, const float* att_w, const float* att_b, float* pooled,
                      float* att_tanh, float* att_weight, int32_t n_titles, int32_t title_len, int32_t filters,
                      float drop, uint32_t key, lstur_stream stream) {
  if (!conv_out || !att_w || !att_b || !pooled || !att_tanh || !att_weight) return fail(LSTUR_ERR_ARG, "attpool: NULL");
  launch_attpool_fwd(conv_out, ids, att_w, att_b, pooled, att_tanh, att_weight, n_titles, title_len, filters, drop, key,
                     (hipStream_t)stream);
  return check_launch("attpool_fwd");
}
// ---- split-operand conv family (convx.hip): parts = 2 (f16x2, with a scale block) or 3 (bf16x3, scales may be NULL) ----
static int convx_args_ok(int32_t dim, int32_t ksize, int32_t filters, int32_t parts, const float* scales, const char* who) {
  if (parts != 2 && parts != 3) return fail(LSTUR_ERR_ARG, std::string(who) + ": parts must be 2 (f16x2) or 3 (bf16x3)");
  if (!convx_supported(dim, ksize, filters))
    return fail(LSTUR_ERR_ARG, std::string(who) + ": dim % 4 == 0, filters <= 448, ksize <= 8");
  if (parts == 2 && !scales) return fail(LSTUR_ERR_ARG, std::string(who) + ": parts == 2 needs the scale block");
  return LSTUR_OK;
}
size_t lstur_convx_scales_bytes(void) { return kCsWords * sizeof(float); }
int lstur_convx_scales_init(float* scales, lstur_stream stream) {
  if (!scales) return fail(LSTUR_ERR_ARG, "convx_scales_init: NULL");
  launch_convx_scales_init(scales, (hipStream_t)stream);
  return check_launch("convx_scales_init");
}
int lstur_convx_embedding_scale(const float* table, size_t count, float drop, float* scales, lstur_stream stream) {
  if (!table || !scales || !(drop >= 0.f && drop < 1.f)) return fail(LSTUR_ERR_ARG, "convx_embedding_scale: bad argument");
  launch_convx_escale(table, count, drop, scales, (hipStream_t)stream);
  return check_launch("convx_embedding_scale");
}
size_t lstur_convx_e_elems(int32_t n_titles, int32_t title_len, int32_t dim, int32_t ksize, int32_t parts) {
  return (parts == 2 || parts == 3) ? convx_e_elems(n_titles, title_len, dim, ksize, parts) : 0;
}
int lstur_convx_gather(const float* table, const int32_t* ids, uint16_t* ex, int32_t n_titles, int32_t title_len, int32_t dim,
                       int32_t pad_left, int32_t pad_right, float drop, uint32_t key, int32_t skip_unread, int32_t parts,
                       const float* scales, const uint32_t* live, lstur_stream stream) {
  if (!table || !ids || !ex || dim % 4 || n_titles < 0 || (parts != 2 && parts != 3) || (parts == 2 && !scales))
    return fail(LSTUR_ERR_ARG, "convx_gather: bad argument (dim % 4, parts 2 | 3, scales)");
  if (live && title_len > 32) return fail(LSTUR_ERR_ARG, "convx_gather: the non-pad masks hold 32 tokens per title");
  launch_gatherx(table, ids, ex, n_titles, title_len, dim, pad_left, pad_right, drop, key, skip_unread != 0, parts, scales,
                 (hipStream_t)stream, 0, live);
  return check_launch("convx_gather");
}
size_t lstur_convx_w_elems(int32_t dim, int32_t ksize, int32_t filters, int32_t parts) {
  return ((parts == 2 || parts == 3) && convx_supported(dim, ksize, filters)) ? convx_w_elems(dim, ksize, filters, parts) : 0;
}
int lstur_convx_pack_w(const float* w, int32_t dim, int32_t ksize, int32_t filters, uint16_t* wx, int32_t parts, float* scales,
                       lstur_stream stream) {
  if (!w || !wx) return fail(LSTUR_ERR_ARG, "convx_pack_w: NULL");
  if (int rc = convx_args_ok(dim, ksize, filters, parts, scales, "convx_pack_w")) return rc;
  launch_convx_pack_w(w, dim, ksize, filters, wx, parts, scales, 0, true, 0.f, (hipStream_t)stream);
  return check_launch("convx_pack_w");
}
int lstur_convx_fwd(const uint16_t* ex, const uint16_t* wx, const float* bias, const int32_t* ids, float* out, int32_t n_titles,
                    int32_t title_len, int32_t dim, int32_t ksize, int32_t filters, const int32_t* vidx, const int32_t* vcount,
                    int32_t parts, const float* scales, void* scratch, lstur_stream stream) {
  if (!ex || !wx || !bias || !ids || !out) return fail(LSTUR_ERR_ARG, "convx_fwd: NULL");
  if (int rc = convx_args_ok(dim, ksize, filters, parts, scales, "convx_fwd")) return rc;
  if ((vidx == nullptr) != (vcount == nullptr)) return fail(LSTUR_ERR_ARG, "convx_fwd: vidx and vcount go together");
  if (convx_e_elems(n_titles, title_len, dim, ksize, parts) * 2 >= ((size_t)1 << 32) ||
      (size_t)n_titles * title_len * filters * 4 >= ((size_t)1 << 32))
    return fail(LSTUR_ERR_ARG, "convx_fwd: more than 4 GiB of embeddings / outputs per call");
  launch_convx_fwd(ex, wx, bias, ids, out, n_titles, title_len, dim, ksize, filters, vidx, vcount, parts, scales, scratch,
                   (hipStream_t)stream);
  return check_launch("convx_fwd");
}
size_t lstur_convx_fwd_scratch(void) { return convx_fwd_scratch_bytes(); }
size_t lstur_convx_wgrad_scratch(int32_t n_titles, int32_t title_len, int32_t dim, int32_t ksize, int32_t filters) {
  return convx_supported(dim, ksize, filters) ? convx_wgrad_scratch_bytes(n_titles, title_len, dim, ksize, filters) : 0;
}
int lstur_convx_wgrad(const uint16_t* ex, const float* dout, float* dw, void* scratch, int32_t* offsets, int32_t n_titles,
                      int32_t title_len, int32_t dim, int32_t ksize, int32_t filters, const int32_t* vidx,
                      const int32_t* vcount, int32_t parts, float* scales, lstur_stream stream) {
  if (!ex || !dout || !dw || !scratch || !offsets) return fail(LSTUR_ERR_ARG, "convx_wgrad: NULL");
  if (int rc = convx_args_ok(dim, ksize, filters, parts, scales, "convx_wgrad")) return rc;
  if ((vidx == nullptr) != (vcount == nullptr)) return fail(LSTUR_ERR_ARG, "convx_wgrad: vidx and vcount go together");
  const int M = n_titles * title_len;
  if (convx_e_elems(n_titles, title_len, dim, ksize, parts) * 2 >= ((size_t)1 << 32) || (size_t)M * filters * 4 >= ((size_t)1 << 32))
    return fail(LSTUR_ERR_ARG, "convx_wgrad: more than 4 GiB of embeddings / gradients per call");
  hipStream_t s = (hipStream_t)stream;
  launch_token_offsets(vidx, vcount, M, title_len, title_len + ksize - 1, convx_dp(dim) / 8 * parts, filters, offsets, offsets + M, s);
  if (parts == 2) {   // the stand-alone call takes max |dout| over the rows it reads (in the step the attention backward leaves it)
    (void)hipMemsetAsync(scales + kCsGmax, 0, sizeof(float), s);
    launch_absmax_rows(dout, M, filters, vidx, vcount, (uint32_t*)scales + kCsGmax, s);   // (bits of a non-negative float)
  }
  ReduceJob jobs[kMaxReduceJobs];
  const int nj = launch_convx_wgrad(ex, dout, dw, scratch, n_titles, title_len, dim, ksize, filters, vcount, s, jobs, offsets,
                                    offsets + M, parts, scales, parts == 2 ? scales + kCsGmax : nullptr, 1);
  if (M > 0 && !launch_multi_reduce(jobs, nj, s)) return fail(LSTUR_ERR_STATE, "convx_wgrad: too many reductions");
  return check_launch("convx_wgrad");
}
int lstur_mhsa_fwd(const float* qkv, const int32_t* ids, float* out, int32_t n_titles, int32_t title_len,
                   int32_t filters, int32_t heads, lstur_stream stream) {
  if (!qkv || !ids || !out) return fail(LSTUR_ERR_ARG, "mhsa_fwd: NULL");
  if (n_titles < 0 || !mhsa_supported(title_len, filters, heads))
    return fail(LSTUR_ERR_ARG, "mhsa: title_len <= 32, filters <= 512, filters / heads in {16, 32, 64}");
  launch_mhsa_fwd(qkv, ids, out, n_titles, title_len, filters, heads, (hipStream_t)stream);
  return check_launch("mhsa_fwd");
}
int lstur_mhsa_bwd(const float* qkv, const float* dout, const int32_t* ids, float* dqkv, int32_t n_titles,
                   int32_t title_len, int32_t filters, int32_t heads, lstur_stream stream) {
  if (!qkv || !dout || !ids || !dqkv) return fail(LSTUR_ERR_ARG, "mhsa_bwd: NULL");
  if (n_titles < 0 || !mhsa_supported(title_len, filters, heads))
    return fail(LSTUR_ERR_ARG, "mhsa: title_len <= 32, filters <= 512, filters / heads in {16, 32, 64}");
  launch_mhsa_bwd(qkv, dout, ids, dqkv, n_titles, title_len, filters, heads, (hipStream_t)stream);
  return check_launch("mhsa_bwd");
}
int lstur_proj_fwd_bf16(const float* E, const float* W, const float* bias, uint16_t* out, int32_t M, int32_t N, int32_t K,
                        lstur_stream stream) {
  if (!E || !W || !out) return fail(LSTUR_ERR_ARG, "proj_fwd_bf16: NULL");
  if (M < 0 || N < 8 || (N % 8) || K < 4 || (K % 4)) return fail(LSTUR_ERR_ARG, "proj_fwd_bf16: N % 8 == 0, K % 4 == 0");
  launch_proj_fwd_bf16(E, W, bias, out, M, N, K, (hipStream_t)stream);
  return check_launch("proj_fwd_bf16");
}
size_t lstur_proj_wgrad_scratch(int32_t D, int32_t N, int32_t K) { return proj_wgrad_scratch_bytes(D, N, K); }
int lstur_proj_wgrad_bf16(const float* E, const uint16_t* dout, float* dw, void* scratch, int32_t D, int32_t N, int32_t K,
                          lstur_stream stream) {
  if (!E || !dout || !dw || !scratch) return fail(LSTUR_ERR_ARG, "proj_wgrad_bf16: NULL");
  if (D < 4 || (D % 4) || D > 316 || N < 8 || (N % 8) || K < 1)
    return fail(LSTUR_ERR_ARG, "proj_wgrad_bf16: D % 4 == 0, D <= 316, N % 8 == 0");
  ReduceJob job;
  launch_proj_wgrad_bf16(E, dout, dw, scratch, D, N, K, (hipStream_t)stream, &job);
  launch_multi_reduce(&job, 1, (hipStream_t)stream);
  return check_launch("proj_wgrad_bf16");
}
int32_t lstur_attpool_bwd_parts(int32_t n_titles) { return attpool_bwd_parts(n_titles); }
int lstur_attpool_bwd(const float* conv_out, const int32_t* ids, const float* att_w, const float* att_tanh,
                      const float* att_weight, const float* dpooled, float* dconv, float* parts, float* row_scratch,
                      int32_t n_titles, int32_t title_len, int32_t filters, float drop, uint32_t key,
                      lstur_stream stream) {
  if (!conv_out || !att_w || !att_tanh || !att_weight || !dpooled || !dconv || !parts || !row_scratch ||
      filters > 1024)
    return fail(LSTUR_ERR_ARG, "attpool bwd: bad argument");
  launch_attpool_bwd(conv_out, ids, att_w, att_tanh, att_weight, dpooled, dconv, parts, row_scratch, n_titles,
                     title_len, filters, drop, key, (hipStream_t)stream);
  return check_launch("attpool_bwd");
}
int lstur_colsum(const float* x, int32_t rows, int32_t cols, int32_t ld, float* out, float scale, lstur_stream stream) {
  if (!x || !out) return fail(LSTUR_ERR_ARG, "colsum: NULL");
  launch_colsum(x, rows, cols, ld, out, scale, (hipStream_t)stream);
  return check_launch("colsum");
}
size_t lstur_gemm_scratch(int32_t M, int32_t N, int32_t K) { return gemm_scratch_bytes(M, N, K); }
int lstur_gemm_bf16x3(const float* A, int32_t lda, const float* B, int32_t ldb, int32_t b_nk, float* C, int32_t ldc,
                      const float* bias, int32_t M, int32_t N, int32_t K, lstur_stream stream) {
  if (!A || !B || !C) return fail(LSTUR_ERR_ARG, "NULL argument");
  if (!gemm3_supported(lda, ldb, M, N, K)) return fail(LSTUR_ERR_ARG, "gemm_bf16x3: K, lda, ldb must be positive multiples of 4");
  launch_gemm3(A, lda, B, ldb, b_nk != 0, C, ldc, bias, M, N, K, (hipStream_t)stream);
  return check_launch("gemm_bf16x3");
}
size_t lstur_gemm_bf16x3_tn_scratch(int32_t M, int32_t N, int32_t K) { return gemm3_tn_scratch_bytes(M, N, K); }
int lstur_gemm_bf16x3_tn(const float* A, int32_t lda, const float* B, int32_t ldb, float* C, int32_t ldc, int32_t M, int32_t N,
                         int32_t K, void* scratch, lstur_stream stream) {
  if (!A || !B || !C || !scratch) return fail(LSTUR_ERR_ARG, "NULL argument");
  if (!gemm3_tn_supported(lda, ldb, M, N, K)) return fail(LSTUR_ERR_ARG, "gemm_bf16x3_tn: lda, ldb must be multiples of 4");
  ReduceJob job;
  launch_gemm3_tn(A, lda, B, ldb, C, ldc, M, N, K, 1.f, scratch, (hipStream_t)stream, &job);
  launch_multi_reduce(&job, 1, (hipStream_t)stream);
  return check_launch("gemm_bf16x3_tn");
}
int lstur_gemm_f32(const float* A, int32_t lda, int32_t ta, const float* B, int32_t ldb, int32_t tb, float* C,
                   int32_t ldc, const float* bias, int32_t M, int32_t N, int32_t K, float alpha, void* scratch,
                   lstur_stream stream) {
  if (!A || !B || !C) return fail(LSTUR_ERR_ARG, "gemm: NULL");
  launch_gemm(A, lda, ta, B, ldb, tb, C, ldc, bias, M, N, K, alpha, scratch, (hipStream_t)stream);
  return check_launch("gemm");
}
size_t lstur_gru_pack_floats(int32_t H) { return gru_pack_floats(H); }
int lstur_gru_pack_u(const float* gru_u, int32_t H, float* packed, lstur_stream stream) {
  if (!gru_u || !packed || H < 1 || H > 256) return fail(LSTUR_ERR_ARG, "gru pack: bad argument");
  launch_gru_pack(gru_u, H, packed, (hipStream_t)stream, gru4_f16_env(), 0);
  return check_launch("gru_pack");
}
size_t lstur_gru_tape_floats(int32_t n, int32_t W, int32_t H) { return gru_tape_floats(n, W, H); }
int lstur_gru_fwd(const float* xw, const uint8_t* gmask, const float* h0, const float* packed, float* tape,
                  float* hlast, int32_t n, int32_t W, int32_t H, lstur_stream stream) {
  if (!xw || !gmask || !packed || !tape || !hlast || n < 0 || W < 1 || H < 1 || H > 256 ||
      (size_t)n * W * 3 * H >= ((size_t)1 << 30))
    return fail(LSTUR_ERR_ARG, "gru fwd: bad argument");
  launch_gru_fwd(xw, gmask, h0, packed, tape, hlast, n, W, H, (hipStream_t)stream, gru4_f16_env());
  return check_launch("gru_fwd");
}
int lstur_gru_bwd(const float* dh_last, const uint8_t* gmask, float* tape, const float* packed, float* dgates,
                  float* rh, float* hprev, float* dh0, int32_t n, int32_t W, int32_t H, lstur_stream stream) {
  if (!dh_last || !gmask || !tape || !packed || !dgates || !rh || !hprev || !dh0 || n < 0 || W < 1 || H < 1 || H > 256 ||
      (size_t)n * W * 3 * H >= ((size_t)1 << 30))
    return fail(LSTUR_ERR_ARG, "gru bwd: bad argument");
  launch_gru_bwd(dh_last, gmask, tape, packed, dgates, rh, hprev, dh0, n, W, H, (hipStream_t)stream);
  return check_launch("gru_bwd");
}
int lstur_score_softmax_ce(const float* u, const float* d, float* logits, float* probs, float* loss_i, float* correct_i,
                           float* du, float* dd, int32_t n, int32_t K1, int32_t U, float inv_n, lstur_stream stream) {
  if (!u || !d || !logits || !probs || !loss_i || !correct_i || K1 > 64) return fail(LSTUR_ERR_ARG, "score: bad argument");
  launch_score(u, d, logits, probs, loss_i, correct_i, du, dd, n, K1, U, inv_n, (hipStream_t)stream);
  return check_launch("score");
}
int lstur_adam_flat(float* p, const float* g, float* m, float* v, size_t count, float lr_t, float beta1, float beta2,
                    float eps, lstur_stream stream) {
  if (!p || !g || !m || !v) return fail(LSTUR_ERR_ARG, "adam: NULL");
  launch_adam_flat(p, g, m, v, count, lr_t, beta1, beta2, eps, (hipStream_t)stream);
  return check_launch("adam_flat");
}

}  // extern "C"
