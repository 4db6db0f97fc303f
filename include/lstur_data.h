/*
 * lstur_data.h -- C-ABI of liblstur_data.so: native loaders and sample feeds for the LSTUR path (host C++).
 *
 * Replaces the Python generators on the training hot loop of the reference
 * (task/paper.py:528-542 `Seq2VecPaperSoftmaxId.train_gen`, task/seq2vec.py:182-193 `Seq2Vec.train`):
 * per positive click one sample (user, last-W clicked doc ids, [pos, K negatives]); a pool of >= 100*B samples
 * is shuffled before every batch and its first B samples are emitted.  Output is in doc-id form (the title table
 * is resident on the GPU).  The NumPy legacy global RNG is reproduced bit for bit: MT19937 seeded like
 * np.random.seed(int), `np.random.choice(neg, K)` = K masked-rejection draws of randint(0, len(neg)),
 * `np.random.shuffle(list)` = Fisher-Yates from the back with random_interval -- so for the same seed the feed
 * emits exactly the batches the reference's generators emit.
 *
 * Plain pointers and sizes; 0 on success, negative on error (lstur_data_last_error()).  Host memory only.
 */
#ifndef LSTUR_DATA_H
#define LSTUR_DATA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lstur_data lstur_data;

const char* lstur_data_last_error(void);

/* Training histories in CSR form.  User u (= row number of ClickData.tsv, paper.py:531) owns training impressions
 * imp_off[u] .. imp_off[u+1]; impression i has positives pos[pos_off[i] .. pos_off[i+1]) and negatives
 * neg[neg_off[i] .. neg_off[i+1]) (doc ids).  The arrays are copied.
 * rank/world: data-parallel shard, this feed walks users rank, rank+world, ...  (world = 1: all users). */
int lstur_data_create(int32_t n_users, const int64_t* imp_off, const int64_t* pos_off, const int32_t* pos,
                      const int64_t* neg_off, const int32_t* neg, int32_t window, int32_t negatives,
                      int32_t batch, int32_t rank, int32_t world, lstur_data** out);
int lstur_data_destroy(lstur_data* d);

/* np.random.seed(seed) for this feed's private generator (also restarts the walk and empties the pool). */
int lstur_data_seed(lstur_data* d, uint32_t seed);
/* Pointwise tasks: a user with more than max_impression training impressions trains on
 * random.sample(impressions, max_impression) (task/paper.py:42-43, 271-272: Python's `random`, not NumPy's).  The feed
 * restates CPython 3's sample() on the same MT19937 core; the host hands over the generator: the 624 state words and the
 * position of random.getstate()[1].  0 turns the sub-sampling off.  Restarts every walk. */
int lstur_data_pyrandom(lstur_data* d, const uint32_t* mt624, int32_t pos, int32_t max_impression);

/* Next pool-shuffled batch: user int32 [B], hist int32 [B, W] (front-padded with 0, newest last),
 * cands int32 [B, K+1] (candidate 0 = the clicked doc). */
int lstur_data_next_train(lstur_data* d, int32_t* user, int32_t* hist, int32_t* cands);

/* ---- loaders (mnexp_amd/csrc/lstur_load.cpp): the three files of the path, parsed natively -----------------------------
 * DocMeta.tsv (Seq2Vec._load_docs, task/seq2vec.py:85-110; document.parse_document / pad_document(1, T), document.py:23-54):
 * column [1] = doc index, [4] = title ids ("#N#"-separated sentences of space-separated ids: the first one is kept,
 * truncated / zero-padded to title_len); a later line with the same index wins; doc 0 = all-zero pad title.
 * Where the reference raises (too few columns, an empty token) the call fails with the line number in the message. */
typedef struct lstur_docs lstur_docs;
int lstur_docs_load(const char* path, int32_t title_len, lstur_docs** out);
int32_t lstur_docs_count(const lstur_docs* d);              /* max index + 1                           */
const int32_t* lstur_docs_titles(const lstur_docs* d);      /* int32 [doc_count, title_len]             */
const uint8_t* lstur_docs_present(const lstur_docs* d);     /* uint8 [doc_count]: the index occurs (or 0) */
int lstur_docs_destroy(lstur_docs* d);

/* ClickData.tsv (Seq2VecPaper._load_data + Impression, task/paper.py:7-35): column [2] = training-period history, [3] =
 * validation-period history; impressions joined by "#N#", fields by "#TAB#": positives, negatives, optional time
 * "%m/%d/%Y %I:%M:%S %p"; impressions that start with "#TAB#" (no positives) are dropped; user id = line number.
 * present / doc_count (may be NULL / 0): every doc id must be a known document (the task's load-time check). */
typedef struct lstur_clicks lstur_clicks;
int lstur_clicks_load(const char* path, const uint8_t* present, int32_t doc_count, lstur_clicks** out);
int32_t lstur_clicks_users(const lstur_clicks* c);
int32_t lstur_clicks_timed(const lstur_clicks* c);          /* 1: every impression carries a time stamp */
/* which = 0 training / 1 validation period, CSR: user u owns impressions imp_off[u] .. imp_off[u+1]; impression i has
 * positives pos[pos_off[i] .. pos_off[i+1]), negatives neg[neg_off[i] .. neg_off[i+1]), time[i] (seconds since
 * 1970-01-01, naive; INT64_MIN: none).  The pointers stay valid until lstur_clicks_destroy. */
int lstur_clicks_arrays(const lstur_clicks* c, int32_t which, const int64_t** imp_off, const int64_t** pos_off,
                        const int32_t** pos, const int64_t** neg_off, const int32_t** neg, const int64_t** time,
                        int64_t* n_imp, int64_t* n_pos, int64_t* n_neg);
int lstur_clicks_destroy(lstur_clicks* c);

/* Vocab.tsv (utils.load_textual_embedding, utils.py:34-52): rows "word \t index \t v0 v1 ..."; rows whose vector does not
 * have exactly dim fields are skipped; matrix float32 [rows, dim] with rows = max index + 1, missing rows (row 0) zero.
 * Free with lstur_vocab_free. */
int lstur_vocab_load(const char* path, int32_t dim, float** matrix, int32_t* rows);
void lstur_vocab_free(float* matrix);

/* ---- the feed over a loaded ClickData: training, validation and test generators of the task family -------------------
 * mode 0: listwise samples of Seq2VecPaperSoftmax[Id] (task/paper.py:385-441, 528-582): cands = [pos, K negatives];
 * mode 1: pointwise rows of Seq2VecPaper / ...Dot / ...Id (task/paper.py:38-97, 266-328): a positive row (label 1), then K
 *         rows of sampled negatives (label 0); cands has ONE column (users with more than max_impression training
 *         impressions are sub-sampled with Python's random.sample in the reference: lstur_data_pyrandom);
 * days > 0: the time-aware click window of the `...Days` tasks (task/paper.py:669-792): an entry is visible for `days` days
 *           (fractions allowed, as timedelta(days=...) allows them).
 * The arrays are copied: the lstur_clicks handle may be destroyed afterwards. */
int lstur_data_create_from_clicks(const lstur_clicks* c, int32_t window, int32_t negatives, int32_t batch, int32_t rank,
                                  int32_t world, int32_t mode, double days, lstur_data** out);
/* lstur_data_next_train with the label column (pointwise rows; label may be NULL) */
int lstur_data_next_train_rows(lstur_data* d, int32_t* user, int32_t* hist, int32_t* cands, int32_t* label);
/* Seq2Vec.valid (task/seq2vec.py:195-200): the next B records of valid_gen, unshuffled.  Walks ALL users (no shard):
 * history = every training click, a validation impression's positives enter it after the impression. */
int lstur_data_next_valid(lstur_data* d, int32_t* user, int32_t* hist, int32_t* cands, int32_t* label);
/* test_gen (task/paper.py:79-97, 420-441, 563-582): one validation impression at a time -- all positives (label 1), then
 * all negatives (label 0), one candidate per row, the history frozen in front of the impression.
 * lstur_data_test_rows: rows of the NEXT impression (0: the pass is over); lstur_data_test_next: writes them (user ==
 * NULL: skips the impression -- data-parallel evaluation takes every world-th one); lstur_data_test_reset: from the top. */
int lstur_data_test_reset(lstur_data* d);
int64_t lstur_data_test_rows(lstur_data* d);
int lstur_data_test_next(lstur_data* d, int32_t* user, int32_t* hist, int32_t* cand, int32_t* label);

/* Synthetic DocMeta.tsv / ClickData.tsv of a workload's shape in the reference's formats (mnexp_amd/csrc/lstur_synth.cpp;
 * distributions of SURVEY.md 8d / mnexp_amd/synth.py, its own generator): bench.py's end-to-end legs at 50 k .. 1 M users.
 * Workload, not fixture: nothing is pinned on these bytes. */
int lstur_synth_write(const char* docmeta_path, const char* clickdata_path, int32_t n_docs, int32_t vocab,
                      int32_t title_len, int32_t n_users, double lam_train, double lam_valid, uint64_t seed);

/* Next n raw (unshuffled) samples of the underlying generator, same layout (for parity tests). */
int lstur_data_next_samples(lstur_data* d, int32_t n, int32_t* user, int32_t* hist, int32_t* cands);

/* Four draws of np.random.randint(0, 1 << 30) from the feed's generator state (parity tests). */
int lstur_data_peek_rng(lstur_data* d, int32_t* out4);

#ifdef __cplusplus
}
#endif
#endif /* LSTUR_DATA_H */
