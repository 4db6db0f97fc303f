// Native training-sample feed (host C++) -- see include/lstur_data.h.
//
// Mirrors, bit for bit, what the reference's Python does on the training hot loop:
//   Seq2VecPaperSoftmaxId.train_gen   task/paper.py:528-542   (walk users / impressions / positives,
//                                                              Window of the last W clicks, K negatives)
//   Seq2Vec.train                     task/seq2vec.py:182-193 (pool of >= 100*B samples, shuffle, emit B)
// with NumPy's legacy RandomState algorithms restated from their published definitions:
//   MT19937 (init_genrand / genrand_int32), randint via masked rejection on 32-bit draws,
//   shuffle of a Python list = Fisher-Yates from the back with random_interval.
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/lstur_data.h"

namespace {

thread_local std::string g_err;
int fail(const char* m) {
  g_err = m;
  return -1;
}

// ---- MT19937, NumPy legacy seeding for an integer seed ---------------------------------------------
struct MT19937 {
  uint32_t key[624];
  int pos;
  void seed(uint32_t s) {
    key[0] = s;
    for (int i = 1; i < 624; ++i) key[i] = 1812433253u * (key[i - 1] ^ (key[i - 1] >> 30)) + (uint32_t)i;
    pos = 624;
  }
  void gen() {
    const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MAG = 0x9908b0dfu;
    int kk;
    uint32_t y;
    for (kk = 0; kk < 624 - 397; ++kk) {
      y = (key[kk] & UPPER) | (key[kk + 1] & LOWER);
      key[kk] = key[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? MAG : 0u);
    }
    for (; kk < 623; ++kk) {
      y = (key[kk] & UPPER) | (key[kk + 1] & LOWER);
      key[kk] = key[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? MAG : 0u);
    }
    y = (key[623] & UPPER) | (key[0] & LOWER);
    key[623] = key[396] ^ (y >> 1) ^ ((y & 1u) ? MAG : 0u);
    pos = 0;
  }
  uint32_t next32() {
    if (pos == 624) gen();
    uint32_t y = key[pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
  }
  // smallest 2^k - 1 >= v
  static uint32_t mask_of(uint32_t v) {
    v |= v >> 1;
    v |= v >> 2;
    v |= v >> 4;
    v |= v >> 8;
    v |= v >> 16;
    return v;
  }
  // uniform integer in [0, max] by masked rejection (randint(0, max+1) and random_interval(max) coincide
  // for max < 2^32: no draw at all when max == 0)
  uint32_t bounded(uint32_t max) {
    if (max == 0) return 0;
    const uint32_t mask = mask_of(max);
    uint32_t v;
    while ((v = (next32() & mask)) > max) {
    }
    return v;
  }
};

}  // namespace

struct lstur_data {
  int32_t n_users, W, K, B, rank, world;
  std::vector<int64_t> imp_off, pos_off, neg_off;
  std::vector<int32_t> pos, neg;
  MT19937 rng;
  // generator position (train_gen is an endless loop over the users of this shard)
  int64_t user, imp;        // current user, current impression (absolute index)
  int64_t pidx;             // next positive inside the impression
  std::vector<int32_t> window;  // last W clicks of the current user, newest last
  int64_t count;            // pushes so far for the current user
  bool user_open;
  // sample arena + pool of slot ids (pool order = the Python list's order)
  int rec;                  // ints per sample: user | hist[W] | cands[K+1]
  std::vector<int32_t> arena;
  std::vector<int32_t> free_slots, pool;

  void restart() {
    user = rank;
    user_open = false;
    pool.clear();
    free_slots.clear();
    const int cap = 100 * B + B + 8;
    arena.assign((size_t)cap * rec, 0);
    for (int i = cap - 1; i >= 0; --i) free_slots.push_back(i);
  }
  bool any_train_user() const {
    for (int64_t u = rank; u < n_users; u += world)
      if (imp_off[u + 1] > imp_off[u]) return true;
    return false;
  }
  void open_user() {
    window.assign(W, 0);
    count = 0;
    imp = imp_off[user];
    pidx = pos_off[imp];
    user_open = true;
  }
  // next sample of train_gen into dst (rec ints)
  void next_sample(int32_t* dst) {
    for (;;) {
      if (!user_open) {
        while (imp_off[user + 1] == imp_off[user]) {   // `if ih:` -- users without training history are skipped
          user += world;
          if (user >= n_users) user = rank;
        }
        open_user();
      }
      if (imp >= imp_off[user + 1]) {   // user exhausted -> next user of the shard (wrapping: `while True`)
        user += world;
        if (user >= n_users) user = rank;
        user_open = false;
        continue;
      }
      if (pidx >= pos_off[imp + 1]) {   // impression exhausted
        ++imp;
        if (imp < imp_off[user + 1]) pidx = pos_off[imp];
        continue;
      }
      const int32_t p = pos[pidx++];
      bool emit = count > 0;
      if (emit) {
        dst[0] = (int32_t)user;
        memcpy(dst + 1, window.data(), (size_t)W * sizeof(int32_t));
        int32_t* c = dst + 1 + W;
        c[0] = p;
        const int64_t n0 = neg_off[imp], nn = neg_off[imp + 1] - n0;
        for (int k = 0; k < K; ++k) c[1 + k] = neg[n0 + rng.bounded((uint32_t)(nn - 1))];   // np.random.choice(neg, K)
      }
      // ch.push(pos)
      memmove(window.data(), window.data() + 1, (size_t)(W - 1) * sizeof(int32_t));
      window[W - 1] = p;
      ++count;
      if (emit) return;
    }
  }
};

extern "C" {

const char* lstur_data_last_error(void) { return g_err.c_str(); }

int lstur_data_create(int32_t n_users, const int64_t* imp_off, const int64_t* pos_off, const int32_t* pos,
                      const int64_t* neg_off, const int32_t* neg, int32_t window, int32_t negatives, int32_t batch,
                      int32_t rank, int32_t world, lstur_data** out) {
  if (!imp_off || !pos_off || !pos || !neg_off || !neg || !out) return fail("NULL argument");
  if (n_users < 1 || window < 1 || negatives < 0 || batch < 1 || world < 1 || rank < 0 || rank >= world)
    return fail("bad dimensions");
  lstur_data* d = new lstur_data();
  d->n_users = n_users;
  d->W = window;
  d->K = negatives;
  d->B = batch;
  d->rank = rank;
  d->world = world;
  d->imp_off.assign(imp_off, imp_off + n_users + 1);
  const int64_t n_imp = imp_off[n_users];
  d->pos_off.assign(pos_off, pos_off + n_imp + 1);
  d->neg_off.assign(neg_off, neg_off + n_imp + 1);
  d->pos.assign(pos, pos + pos_off[n_imp]);
  d->neg.assign(neg, neg + neg_off[n_imp]);
  for (int64_t i = 0; i < n_imp; ++i)
    if (d->neg_off[i + 1] == d->neg_off[i] || d->pos_off[i + 1] == d->pos_off[i]) {
      delete d;
      return fail("every impression needs at least one positive and one negative");
    }
  if (rank >= n_users || !d->any_train_user()) {
    delete d;
    return fail("this shard has no user with a training history");
  }
  d->rec = 1 + window + negatives + 1;
  d->rng.seed(0);
  d->restart();
  *out = d;
  return 0;
}

int lstur_data_destroy(lstur_data* d) {
  delete d;
  return 0;
}

int lstur_data_seed(lstur_data* d, uint32_t seed) {
  if (!d) return fail("NULL feed");
  d->rng.seed(seed);
  d->restart();
  return 0;
}

int lstur_data_next_samples(lstur_data* d, int32_t n, int32_t* user, int32_t* hist, int32_t* cands) {
  if (!d || !user || !hist || !cands || n < 0) return fail("bad argument");
  std::vector<int32_t> tmp(d->rec);
  for (int i = 0; i < n; ++i) {
    d->next_sample(tmp.data());
    user[i] = tmp[0];
    memcpy(hist + (size_t)i * d->W, tmp.data() + 1, (size_t)d->W * 4);
    memcpy(cands + (size_t)i * (d->K + 1), tmp.data() + 1 + d->W, (size_t)(d->K + 1) * 4);
  }
  return 0;
}

int lstur_data_next_train(lstur_data* d, int32_t* user, int32_t* hist, int32_t* cands) {
  if (!d || !user || !hist || !cands) return fail("bad argument");
  const size_t size = (size_t)100 * d->B;
  // pool.append(next(gen)) until len(pool) >= 100*B
  do {
    const int32_t slot = d->free_slots.back();
    d->free_slots.pop_back();
    d->next_sample(d->arena.data() + (size_t)slot * d->rec);
    d->pool.push_back(slot);
  } while (d->pool.size() < size);
  // np.random.shuffle(pool): for i = n-1 .. 1: j = random_interval(i); swap
  for (size_t i = d->pool.size() - 1; i >= 1; --i) {
    const uint32_t j = d->rng.bounded((uint32_t)i);
    const int32_t t = d->pool[i];
    d->pool[i] = d->pool[j];
    d->pool[j] = t;
  }
  for (int b = 0; b < d->B; ++b) {
    const int32_t slot = d->pool[b];
    const int32_t* r = d->arena.data() + (size_t)slot * d->rec;
    user[b] = r[0];
    memcpy(hist + (size_t)b * d->W, r + 1, (size_t)d->W * 4);
    memcpy(cands + (size_t)b * (d->K + 1), r + 1 + d->W, (size_t)(d->K + 1) * 4);
    d->free_slots.push_back(slot);
  }
  d->pool.erase(d->pool.begin(), d->pool.begin() + d->B);
  return 0;
}

int lstur_data_peek_rng(lstur_data* d, int32_t* out4) {
  if (!d || !out4) return fail("bad argument");
  MT19937 copy = d->rng;
  for (int i = 0; i < 4; ++i) out4[i] = (int32_t)copy.bounded((1u << 30) - 1u);
  return 0;
}

}  // extern "C"
