/*
 * lstur_data.h -- C-ABI of liblstur_data.so: native training-sample feed for the LSTUR path (host C++).
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

/* Next pool-shuffled batch: user int32 [B], hist int32 [B, W] (front-padded with 0, newest last),
 * cands int32 [B, K+1] (candidate 0 = the clicked doc). */
int lstur_data_next_train(lstur_data* d, int32_t* user, int32_t* hist, int32_t* cands);

/* Next n raw (unshuffled) samples of the underlying generator, same layout (for parity tests). */
int lstur_data_next_samples(lstur_data* d, int32_t n, int32_t* user, int32_t* hist, int32_t* cands);

/* Four draws of np.random.randint(0, 1 << 30) from the feed's generator state (parity tests). */
int lstur_data_peek_rng(lstur_data* d, int32_t* out4);

#ifdef __cplusplus
}
#endif
#endif /* LSTUR_DATA_H */
