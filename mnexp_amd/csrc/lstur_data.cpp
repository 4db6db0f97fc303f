// Native training-sample feed (host C++) -- see include/lstur_data.h.
//
// Mirrors, bit for bit, what the reference's Python does on the training hot loop:
//   Seq2VecPaperSoftmaxId.train_gen   task/paper.py:528-542   (walk users / impressions / positives,
//                                                              Window of the last W clicks, K negatives)
//   Seq2Vec.train                     task/seq2vec.py:182-193 (pool of >= 100*B samples, shuffle, emit B)
// with NumPy's legacy RandomState algorithms restated from their published definitions:
//   MT19937 (init_genrand / genrand_int32), randint via masked rejection on 32-bit draws,
//   shuffle of a Python list = Fisher-Yates from the back with random_interval.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <cmath>
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

int lstur_data_fail(const std::string& m) {   // (shared with lstur_load.cpp)
  g_err = m;
  return -1;
}

// impressions of one period in CSR form (+ time stamps, seconds since 1970, for the `...Days` tasks)
struct Clicks {
  std::vector<int64_t> imp_off, pos_off, neg_off, time;
  std::vector<int32_t> pos, neg;
  int64_t n_imp() const { return (int64_t)pos_off.size() - 1; }
};

// Seq2Vec.Window (task/seq2vec.py:17-53) and the time-aware window of the `...Days` tasks (task/paper.py:669-696, 763-792):
// the last W clicked docs, front-padded with 0, newest last; with days > 0 an entry is visible until `days` after its click
// and reads as the pad doc 0 IN PLACE afterwards
struct ClickWindow {
  int W = 0;
  bool timed = false;               // the `...Days` window
  int64_t delta = 0;                // seconds an entry stays visible
  int64_t pushes = 0;
  std::vector<int32_t> ids;
  std::vector<int64_t> exp;
  static constexpr int64_t kZero = 946684800;   // datetime(2000, 1, 1): the expiry of the initial pad entries (paper.py:671)
  void reset() {
    ids.assign(W, 0);
    exp.assign(W, kZero);
    pushes = 0;
  }
  void push(int32_t doc, int64_t t) {
    memmove(ids.data(), ids.data() + 1, (size_t)(W - 1) * sizeof(int32_t));
    ids[W - 1] = doc;
    if (timed) {
      memmove(exp.data(), exp.data() + 1, (size_t)(W - 1) * sizeof(int64_t));
      exp[W - 1] = t + delta;
    }
    ++pushes;
  }
  void get(int64_t t, int32_t* out) const {
    if (!timed) {
      memcpy(out, ids.data(), (size_t)W * sizeof(int32_t));
      return;
    }
    for (int i = 0; i < W; ++i) out[i] = exp[i] >= t ? ids[i] : 0;
  }
  int64_t count(int64_t t) const {      // plain: pushes so far (Window.count); days: visible entries (count(time))
    if (!timed) return pushes;
    int64_t n = 0;
    for (int i = 0; i < W; ++i) n += exp[i] >= t;
    return n;
  }
};

struct lstur_data {
  int32_t n_users, W, K, B, rank, world;
  int32_t mode = 0;         // 0: listwise samples (user | hist | pos, K negatives); 1: pointwise rows (user | hist | cand | label)
  Clicks tr, va;
  MT19937 rng;
  // pointwise tasks: a user with more than max_imp training impressions trains on random.sample(impressions, max_imp)
  // (task/paper.py:42-43, 271-272) -- CPython's generator (the same MT19937 core, state handed over by the host:
  // lstur_data_pyrandom) and CPython 3's sample(): `chosen[i]` = impression i of the open user is trainable
  MT19937 pyrng;
  int32_t max_imp = 0;
  std::vector<uint8_t> chosen;
  int C = 0, rec = 0;       // candidates per record; ints per record: user | hist[W] | cands[C] | label
  // ---- train_gen: an endless loop over the users of this shard
  int64_t user, imp, pidx;
  ClickWindow win;
  bool user_open;
  // pointwise: the K negative rows that follow a positive row.  The reference's generator yields the positive row FIRST and
  // draws the negatives when it is resumed (`for neg in impression.negative_samples(K)` behind the yield, paper.py:52-57):
  // a pool shuffle between the two consumes the generator state in between, so the draw is deferred the same way
  struct Pending {
    std::vector<int32_t> rows;      // K records (heads filled, candidates drawn on demand)
    size_t at = 0;
    const Clicks* src = nullptr;    // != NULL: the negatives of impression `imp` are still to be drawn
    int64_t imp = 0;
    bool empty() const { return at >= rows.size(); }
    void clear() {
      rows.clear();
      at = 0;
      src = nullptr;
    }
  };
  Pending pending;
  // sample arena + pool of slot ids (pool order = the Python list's order)
  std::vector<int32_t> arena;
  std::vector<int32_t> free_slots, pool;
  // ---- valid_gen: an endless loop over ALL users with a training and a validation history
  int64_t v_user = 0, v_imp = 0, v_pidx = 0;
  bool v_open = false;
  ClickWindow v_win;
  Pending v_pending;
  // ---- test_gen: one pass over the same users, one impression at a time
  int64_t t_user = 0, t_imp = 0;
  bool t_open = false;
  ClickWindow t_win;

  void restart() {
    user = rank;
    user_open = false;
    pool.clear();
    free_slots.clear();
    pending.clear();
    const int cap = 100 * B + B + 8;
    arena.assign((size_t)cap * rec, 0);
    for (int i = cap - 1; i >= 0; --i) free_slots.push_back(i);
    v_user = 0;
    v_open = false;
    v_pending.clear();
    t_user = 0;
    t_open = false;
  }
  bool any_train_user() const {
    for (int64_t u = rank; u < n_users; u += world)
      if (tr.imp_off[u + 1] > tr.imp_off[u]) return true;
    return false;
  }
  bool any_valid_user() const {
    for (int64_t u = 0; u < n_users; ++u)
      if (tr.imp_off[u + 1] > tr.imp_off[u] && va.imp_off[u + 1] > va.imp_off[u]) return true;
    return false;
  }
  // random.Random._randbelow_with_getrandbits(n): k = n.bit_length() bits per draw (genrand_uint32 >> (32 - k)), rejected
  // while >= n
  uint32_t py_randbelow(uint32_t n) {
    int k = 0;
    for (uint32_t v = n; v; v >>= 1) ++k;
    uint32_t r;
    do r = pyrng.next32() >> (32 - k);
    while (r >= n);
    return r;
  }
  // random.sample(range(n), k) as CPython 3.10 draws it (Lib/random.py: pool selection for small n, a set of chosen indices
  // otherwise); only the membership matters here (the reference builds a set of the sampled impressions)
  void py_sample(int64_t n, int64_t k) {
    chosen.assign((size_t)n, 0);
    double setsize = 21;
    if (k > 5) setsize += std::pow(4.0, std::ceil(std::log((double)k * 3) / std::log(4.0)));
    if ((double)n <= setsize) {
      std::vector<int64_t> pool((size_t)n);
      for (int64_t i = 0; i < n; ++i) pool[i] = i;
      for (int64_t i = 0; i < k; ++i) {
        const uint32_t j = py_randbelow((uint32_t)(n - i));
        chosen[(size_t)pool[j]] = 1;
        pool[j] = pool[n - i - 1];
      }
    } else {
      for (int64_t i = 0; i < k; ++i) {
        uint32_t j = py_randbelow((uint32_t)n);
        while (chosen[j]) j = py_randbelow((uint32_t)n);
        chosen[j] = 1;
      }
    }
  }
  int64_t time_of(const Clicks& c, int64_t i) const { return win.timed ? c.time[i] : 0; }
  // one record: user | window as seen at `t` | cands | label
  void fill_head(int32_t* dst, int64_t u, const ClickWindow& w, int64_t t) const {
    dst[0] = (int32_t)u;
    w.get(t, dst + 1);
  }
  // np.random.choice(neg, K) of impression i: K masked-rejection draws of randint(0, len(neg))
  void draw_negatives(const Clicks& c, int64_t i, int32_t* out) {
    const int64_t n0 = c.neg_off[i], nn = c.neg_off[i + 1] - n0;
    for (int k = 0; k < K; ++k) out[k] = c.neg[n0 + rng.bounded((uint32_t)(nn - 1))];
  }
  // the record(s) of one positive click: listwise one record; pointwise the positive row into dst and K negative rows
  // into `queue` (they are emitted next, task/paper.py:52-57)
  void emit_click(int32_t* dst, int64_t u, const ClickWindow& w, const Clicks& c, int64_t i, int32_t p, Pending* queue) {
    const int64_t t = time_of(c, i);
    fill_head(dst, u, w, t);
    int32_t* cd = dst + 1 + W;
    if (mode == 0) {
      cd[0] = p;
      draw_negatives(c, i, cd + 1);
      cd[C] = 1;
      return;
    }
    cd[0] = p;
    cd[1] = 1;
    queue->rows.assign((size_t)K * rec, 0);
    queue->at = 0;
    queue->src = &c;
    queue->imp = i;
    for (int k = 0; k < K; ++k) memcpy(queue->rows.data() + (size_t)k * rec, dst, (size_t)(1 + W) * sizeof(int32_t));
  }
  // the next queued negative row (draws the impression's K negatives when the first one is asked for)
  void pop_pending(Pending* q, int32_t* dst) {
    if (q->src) {
      std::vector<int32_t> negs(K);
      draw_negatives(*q->src, q->imp, negs.data());
      for (int k = 0; k < K; ++k) {
        q->rows[(size_t)k * rec + 1 + W] = negs[k];
        q->rows[(size_t)k * rec + 2 + W] = 0;
      }
      q->src = nullptr;
    }
    memcpy(dst, q->rows.data() + q->at, (size_t)rec * sizeof(int32_t));
    q->at += rec;
  }
  // next record of train_gen into dst (rec ints)   task/paper.py:38-60, 385-399, 528-542, 698-716
  void next_sample(int32_t* dst) {
    if (!pending.empty()) return pop_pending(&pending, dst);
    for (;;) {
      if (!user_open) {
        while (tr.imp_off[user + 1] == tr.imp_off[user]) {   // `if ih:` -- users without training history are skipped
          user += world;
          if (user >= n_users) user = rank;
        }
        win.reset();
        imp = tr.imp_off[user];
        pidx = tr.pos_off[imp];
        user_open = true;
        const int64_t n_imp = tr.imp_off[user + 1] - tr.imp_off[user];
        if (mode == 1 && max_imp > 0 && n_imp > max_imp) py_sample(n_imp, max_imp);
        else chosen.clear();
      }
      if (imp >= tr.imp_off[user + 1]) {   // user exhausted -> next user of the shard (wrapping: `while True`)
        user += world;
        if (user >= n_users) user = rank;
        user_open = false;
        continue;
      }
      if (pidx >= tr.pos_off[imp + 1]) {   // impression exhausted
        ++imp;
        if (imp < tr.imp_off[user + 1]) pidx = tr.pos_off[imp];
        continue;
      }
      const int32_t p = tr.pos[pidx++];
      const int64_t t = time_of(tr, imp);
      const bool emit = win.count(t) > 0 && (chosen.empty() || chosen[(size_t)(imp - tr.imp_off[user])]);
      if (emit) emit_click(dst, user, win, tr, imp, p, &pending);
      win.push(p, t);                      // ch.push(pos): the history excludes the current click
      if (emit) return;
    }
  }
  // window of a user after all training clicks (paper.py: `for impression in ih1: for pos in impression.pos: ch.push`)
  void after_training(ClickWindow* w, int64_t u) const {
    w->reset();
    for (int64_t i = tr.imp_off[u]; i < tr.imp_off[u + 1]; ++i)
      for (int64_t q = tr.pos_off[i]; q < tr.pos_off[i + 1]; ++q) w->push(tr.pos[q], time_of(tr, i));
  }
  bool both(int64_t u) const { return tr.imp_off[u + 1] > tr.imp_off[u] && va.imp_off[u + 1] > va.imp_off[u]; }
  // next record of valid_gen   task/paper.py:62-77, 401-418, 544-562, 718-735
  void next_valid(int32_t* dst) {
    if (!v_pending.empty()) return pop_pending(&v_pending, dst);
    for (;;) {
      if (!v_open) {
        while (!both(v_user)) v_user = (v_user + 1) % n_users;
        after_training(&v_win, v_user);
        v_imp = va.imp_off[v_user];
        v_pidx = va.pos_off[v_imp];
        v_open = true;
      }
      if (v_imp >= va.imp_off[v_user + 1]) {
        v_user = (v_user + 1) % n_users;
        v_open = false;
        continue;
      }
      if (v_pidx >= va.pos_off[v_imp + 1]) {   // impression done: its positives enter the history, then the next one
        for (int64_t q = va.pos_off[v_imp]; q < va.pos_off[v_imp + 1]; ++q) v_win.push(va.pos[q], time_of(va, v_imp));
        ++v_imp;
        if (v_imp < va.imp_off[v_user + 1]) v_pidx = va.pos_off[v_imp];
        continue;
      }
      const int32_t p = va.pos[v_pidx++];
      emit_click(dst, v_user, v_win, va, v_imp, p, &v_pending);
      return;
    }
  }
  // test_gen (task/paper.py:79-97, 420-441, 563-582, 737-750): rows of the next validation impression, or 0 at the end
  bool test_seek() {
    for (;;) {
      if (t_user >= n_users) return false;
      if (!t_open) {
        if (!both(t_user)) {
          ++t_user;
          continue;
        }
        after_training(&t_win, t_user);
        t_imp = va.imp_off[t_user];
        t_open = true;
      }
      if (t_imp >= va.imp_off[t_user + 1]) {
        ++t_user;
        t_open = false;
        continue;
      }
      return true;
    }
  }
  int64_t test_rows() {
    if (!test_seek()) return 0;
    return (va.pos_off[t_imp + 1] - va.pos_off[t_imp]) + (va.neg_off[t_imp + 1] - va.neg_off[t_imp]);
  }
  void test_take(int32_t* u, int32_t* hist, int32_t* cand, int32_t* label) {   // (test_seek() returned true)
    const int64_t t = time_of(va, t_imp);
    int64_t r = 0;
    for (int pass = 0; pass < 2 && u; ++pass) {
      const std::vector<int32_t>& src = pass ? va.neg : va.pos;
      const int64_t a = pass ? va.neg_off[t_imp] : va.pos_off[t_imp], b = pass ? va.neg_off[t_imp + 1] : va.pos_off[t_imp + 1];
      for (int64_t q = a; q < b; ++q, ++r) {
        u[r] = (int32_t)t_user;
        t_win.get(t, hist + r * W);
        cand[r] = src[q];
        label[r] = pass ? 0 : 1;
      }
    }
    for (int64_t q = va.pos_off[t_imp]; q < va.pos_off[t_imp + 1]; ++q) t_win.push(va.pos[q], t);
    ++t_imp;
  }
};

namespace {

int check_clicks(const Clicks& c, int32_t n_users, const char* what) {
  if ((int64_t)c.imp_off.size() != (int64_t)n_users + 1) return fail("impression offsets do not match the user count");
  for (int64_t i = 0; i < c.n_imp(); ++i)
    if (c.neg_off[i + 1] == c.neg_off[i] || c.pos_off[i + 1] == c.pos_off[i]) {
      g_err = std::string(what) + ": every impression needs at least one positive and one negative";
      return -1;
    }
  return 0;
}

int finish_create(lstur_data* d, lstur_data** out) {
  if (check_clicks(d->tr, d->n_users, "training history") || check_clicks(d->va, d->n_users, "validation history")) {
    delete d;
    return -1;
  }
  if (d->rank >= d->n_users || !d->any_train_user()) {
    delete d;
    return fail("this shard has no user with a training history");
  }
  d->C = d->mode == 0 ? d->K + 1 : 1;
  d->rec = 1 + d->W + d->C + 1;
  d->win.W = d->v_win.W = d->t_win.W = d->W;
  d->v_win.delta = d->t_win.delta = d->win.delta;
  d->v_win.timed = d->t_win.timed = d->win.timed;
  d->rng.seed(0);
  d->restart();
  *out = d;
  return 0;
}

// B records -> the caller's column arrays
void scatter(const lstur_data* d, const int32_t* r, int b, int32_t* user, int32_t* hist, int32_t* cands, int32_t* label) {
  user[b] = r[0];
  memcpy(hist + (size_t)b * d->W, r + 1, (size_t)d->W * 4);
  memcpy(cands + (size_t)b * d->C, r + 1 + d->W, (size_t)d->C * 4);
  if (label) label[b] = r[1 + d->W + d->C];
}

}  // namespace

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
  d->tr.imp_off.assign(imp_off, imp_off + n_users + 1);
  const int64_t n_imp = imp_off[n_users];
  d->tr.pos_off.assign(pos_off, pos_off + n_imp + 1);
  d->tr.neg_off.assign(neg_off, neg_off + n_imp + 1);
  d->tr.pos.assign(pos, pos + pos_off[n_imp]);
  d->tr.neg.assign(neg, neg + neg_off[n_imp]);
  d->va.imp_off.assign((size_t)n_users + 1, 0);   // no validation history: lstur_data_next_valid / _test report it
  d->va.pos_off.assign(1, 0);
  d->va.neg_off.assign(1, 0);
  return finish_create(d, out);
}

// everything a loaded ClickData.tsv holds (lstur_clicks_load): training AND validation histories (+ time stamps).
// mode 0: listwise samples (Seq2VecPaperSoftmax[Id]); 1: pointwise rows (Seq2VecPaper / ...Dot / ...Id: a positive row,
// then K rows of sampled negatives).  days > 0: the time-aware window of the `...Days` tasks (needs time stamps).
int lstur_data_create_from_clicks(const lstur_clicks* c, int32_t window, int32_t negatives, int32_t batch, int32_t rank,
                                  int32_t world, int32_t mode, double days, lstur_data** out) {
  if (!c || !out) return fail("NULL argument");
  const int32_t n_users = lstur_clicks_users(c);
  if (n_users < 1 || window < 1 || negatives < 0 || batch < 1 || world < 1 || rank < 0 || rank >= world || mode < 0 || mode > 1 ||
      !(days >= 0))
    return fail("bad dimensions");
  if (days > 0 && !lstur_clicks_timed(c)) return fail("the `...Days` tasks need a time stamp in every impression");
  lstur_data* d = new lstur_data();
  d->n_users = n_users;
  d->W = window;
  d->K = negatives;
  d->B = batch;
  d->rank = rank;
  d->world = world;
  d->mode = mode;
  // timedelta(days=days) has microsecond resolution, the time stamps whole seconds: click + delta >= t  <=>  t - click <= floor(delta)
  d->win.timed = days > 0;
  d->win.delta = (int64_t)floor(llround(days * 86400e6) / 1e6);
  for (int which = 0; which < 2; ++which) {
    const int64_t *io, *po, *no, *tm;
    const int32_t *p, *n;
    int64_t ni, np_, nn;
    lstur_clicks_arrays(c, which, &io, &po, &p, &no, &n, &tm, &ni, &np_, &nn);
    Clicks& s = which ? d->va : d->tr;
    s.imp_off.assign(io, io + n_users + 1);
    s.pos_off.assign(po, po + ni + 1);
    s.neg_off.assign(no, no + ni + 1);
    s.time.assign(tm, tm + ni);
    s.pos.assign(p, p + np_);
    s.neg.assign(n, n + nn);
  }
  return finish_create(d, out);
}

int lstur_data_destroy(lstur_data* d) {
  delete d;
  return 0;
}

int lstur_data_pyrandom(lstur_data* d, const uint32_t* mt624, int32_t pos, int32_t max_impression) {
  if (!d || !mt624 || pos < 0 || pos > 624 || max_impression < 0) return fail("pyrandom: bad argument");
  memcpy(d->pyrng.key, mt624, sizeof(d->pyrng.key));
  d->pyrng.pos = pos;
  d->max_imp = max_impression;
  d->restart();
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
    scatter(d, tmp.data(), i, user, hist, cands, nullptr);
  }
  return 0;
}

// pool.append(next(gen)) until len(pool) >= 100*B; np.random.shuffle(pool); emit pool[:B]   (task/seq2vec.py:182-193)
int lstur_data_next_train_rows(lstur_data* d, int32_t* user, int32_t* hist, int32_t* cands, int32_t* label) {
  if (!d || !user || !hist || !cands) return fail("bad argument");
  const size_t size = (size_t)100 * d->B;
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
    scatter(d, d->arena.data() + (size_t)slot * d->rec, b, user, hist, cands, label);
    d->free_slots.push_back(slot);
  }
  d->pool.erase(d->pool.begin(), d->pool.begin() + d->B);
  return 0;
}
int lstur_data_next_train(lstur_data* d, int32_t* user, int32_t* hist, int32_t* cands) {
  return lstur_data_next_train_rows(d, user, hist, cands, nullptr);
}

// the next B records of valid_gen, unshuffled (task/seq2vec.py:195-200)
int lstur_data_next_valid(lstur_data* d, int32_t* user, int32_t* hist, int32_t* cands, int32_t* label) {
  if (!d || !user || !hist || !cands) return fail("bad argument");
  if (!d->any_valid_user()) return fail("no user has both a training and a validation history");
  std::vector<int32_t> tmp(d->rec);
  for (int b = 0; b < d->B; ++b) {
    d->next_valid(tmp.data());
    scatter(d, tmp.data(), b, user, hist, cands, label);
  }
  return 0;
}

int lstur_data_test_reset(lstur_data* d) {
  if (!d) return fail("NULL feed");
  d->t_user = 0;
  d->t_open = false;
  return 0;
}
int64_t lstur_data_test_rows(lstur_data* d) { return d ? d->test_rows() : -1; }
int lstur_data_test_next(lstur_data* d, int32_t* user, int32_t* hist, int32_t* cand, int32_t* label) {
  if (!d) return fail("NULL feed");
  if (!d->test_seek()) return fail("test_next: past the last validation impression");
  if (user && (!hist || !cand || !label)) return fail("bad argument");
  d->test_take(user, hist, cand, label);    // user == NULL: skip the impression (its positives still enter the history)
  return 0;
}

int lstur_data_peek_rng(lstur_data* d, int32_t* out4) {
  if (!d || !out4) return fail("bad argument");
  MT19937 copy = d->rng;
  for (int i = 0; i < 4; ++i) out4[i] = (int32_t)copy.bounded((1u << 30) - 1u);
  return 0;
}

}  // extern "C"
