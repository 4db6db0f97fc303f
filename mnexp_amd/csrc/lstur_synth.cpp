// Synthetic DocMeta.tsv / ClickData.tsv writer for bench.py's end-to-end legs and the scale tests (host C++;
// include/lstur_data.h: lstur_synth_write).  The files follow the reference's on-disk formats (README.md:11-23; parsing
// sites task/seq2vec.py:85-110, task/paper.py:10-33) and the distributions of SURVEY.md section 8d / mnexp_amd/synth.py:
// a latent topic per document (32 topics), title length ~ U{5 .. T + 5}, 70 % of a title's tokens from the topic's slice of the
// vocabulary and 30 % Zipf(1.1); per user a Dirichlet(0.3) topic mix, Poisson(lam) impressions (>= 1; 10 % of the users have
// no validation history), 5 .. 20 Zipf-drawn documents per impression clicked with probability sigmoid(3 (32 mix[z] - 1)),
// at least one positive and one negative each, increasing time stamps.  synth.py writes the same thing with NumPy's
// generator (the small fixture sets); this writer exists because a 1 M-user set takes Python loops an hour.  The two do NOT
// produce the same bytes (different generators): nothing is pinned on this output, it is workload, not fixture.
#include <math.h>
#include <stdint.h>
#include <stdio.h>

#include <random>
#include <string>
#include <vector>

#include "../../include/lstur_data.h"

int lstur_data_fail(const std::string& m);

namespace {

constexpr int kTopics = 32;

// Walker alias table over weights 1 / (i + 1)^a, i < n, in a shuffled order
struct Alias {
  std::vector<float> prob;
  std::vector<uint32_t> alias, perm;
  void build(uint32_t n, double a, std::mt19937_64& g, bool shuffle) {
    std::vector<double> p(n);
    double sum = 0;
    for (uint32_t i = 0; i < n; ++i) sum += p[i] = 1.0 / pow((double)i + 1.0, a);
    perm.resize(n);
    for (uint32_t i = 0; i < n; ++i) perm[i] = i;
    if (shuffle)
      for (uint32_t i = n - 1; i > 0; --i) std::swap(perm[i], perm[g() % (i + 1)]);
    prob.assign(n, 0.f);
    alias.assign(n, 0);
    std::vector<uint32_t> small, large;
    for (uint32_t i = 0; i < n; ++i) {
      p[i] = p[i] / sum * n;
      (p[i] < 1.0 ? small : large).push_back(i);
    }
    while (!small.empty() && !large.empty()) {
      const uint32_t s = small.back(), l = large.back();
      small.pop_back();
      prob[s] = (float)p[s];
      alias[s] = l;
      p[l] = p[l] + p[s] - 1.0;
      if (p[l] < 1.0) {
        large.pop_back();
        small.push_back(l);
      }
    }
    for (uint32_t i : large) prob[i] = 1.f;
    for (uint32_t i : small) prob[i] = 1.f;
  }
  uint32_t draw(std::mt19937_64& g) const {
    const uint64_t r = g();
    const uint32_t i = (uint32_t)((r >> 32) % prob.size());
    const float u = (float)(r & 0xFFFFFF) / 16777216.0f;
    return perm[u < prob[i] ? i : alias[i]];
  }
};

struct Out {
  FILE* f;
  std::string buf;
  void num(int64_t v) { buf += std::to_string(v); }
  void flush() {
    if (buf.size() > (1u << 20)) {
      fwrite(buf.data(), 1, buf.size(), f);
      buf.clear();
    }
  }
  void close() {
    fwrite(buf.data(), 1, buf.size(), f);
    fclose(f);
  }
};

}  // namespace

extern "C" int lstur_synth_write(const char* docmeta_path, const char* clickdata_path, int32_t n_docs, int32_t vocab,
                                 int32_t title_len, int32_t n_users, double lam_train, double lam_valid, uint64_t seed) {
  if (!docmeta_path || !clickdata_path || n_docs < 2 || vocab < 2 * kTopics || title_len < 1 || n_users < 1)
    return lstur_data_fail("synth_write: bad argument");
  std::mt19937_64 g(seed * 0x9E3779B97F4A7C15ull + 1);
  // ---- documents
  std::vector<uint8_t> topic(n_docs + 1);
  Alias words;
  words.build((uint32_t)vocab - 1, 1.1, g, false);
  Out d{fopen(docmeta_path, "wb"), {}};
  if (!d.f) return lstur_data_fail(std::string("cannot write ") + docmeta_path);
  const uint32_t slice = (uint32_t)std::max(1, (vocab - 1) / kTopics);
  std::string title;
  for (int32_t i = 1; i <= n_docs; ++i) {
    topic[i] = (uint8_t)(g() % kTopics);
    const int n = 5 + (int)(g() % (uint32_t)(title_len + 1));
    title.clear();
    for (int t = 0; t < n; ++t) {
      const uint32_t lo = 1 + topic[i] * slice;
      const uint32_t id = (g() % 10 < 7) ? std::min<uint32_t>((uint32_t)vocab - 1, lo + (uint32_t)(g() % slice)) : 1 + words.draw(g);
      if (t) title += ' ';
      title += std::to_string(id);
    }
    d.buf += "id";
    d.num(i);
    d.buf += '\t';
    d.num(i);
    d.buf += "\tnews\tnewsus\t";
    d.buf += title;
    d.buf += '\t';
    d.buf += title;
    d.buf += '\n';
    d.flush();
  }
  d.close();
  // ---- users
  Alias docs;
  docs.build((uint32_t)n_docs, 1.1, g, true);
  Out c{fopen(clickdata_path, "wb"), {}};
  if (!c.f) return lstur_data_fail(std::string("cannot write ") + clickdata_path);
  std::gamma_distribution<double> gamma(0.3, 1.0);
  std::poisson_distribution<int> ptrain(lam_train), pvalid(lam_valid);
  std::uniform_real_distribution<double> unif(0.0, 1.0);
  double mix[kTopics];
  std::vector<int32_t> shown, pos, neg;
  std::vector<double> pc;
  char ts[64];
  for (int32_t u = 0; u < n_users; ++u) {
    double s = 0;
    for (int k = 0; k < kTopics; ++k) s += mix[k] = gamma(g) + 1e-300;
    for (int k = 0; k < kTopics; ++k) mix[k] /= s;
    int64_t stamp = (int64_t)(g() % 86400);
    const int n_tr = std::max(1, ptrain(g));
    const int n_va = unif(g) < 0.1 ? 0 : std::max(1, pvalid(g));
    c.buf += 'u';
    c.num(u);
    c.buf += "\tt\t";
    for (int period = 0; period < 2; ++period) {
      const int n_imp = period ? n_va : n_tr;
      for (int i = 0; i < n_imp; ++i) {
        const int n = 5 + (int)(g() % 16);
        shown.clear();
        pc.clear();
        pos.clear();
        neg.clear();
        int best = 0, worst = 0;
        for (int k = 0; k < n; ++k) {
          const int32_t doc = 1 + (int32_t)docs.draw(g);
          const double p = 1.0 / (1.0 + exp(-3.0 * (mix[topic[doc]] * kTopics - 1.0)));
          shown.push_back(doc);
          pc.push_back(p);
          if (p > pc[best]) best = k;
          if (p < pc[worst]) worst = k;
        }
        std::vector<char> click(n);
        int nc = 0;
        for (int k = 0; k < n; ++k) nc += click[k] = unif(g) < pc[k];
        if (nc == 0) click[best] = 1, nc = 1;
        if (nc == n) click[worst == best ? (best + 1) % n : worst] = 0;
        for (int k = 0; k < n; ++k) (click[k] ? pos : neg).push_back(shown[k]);
        if (i) c.buf += "#N#";
        for (size_t k = 0; k < pos.size(); ++k) {
          if (k) c.buf += ' ';
          c.num(pos[k]);
        }
        c.buf += "#TAB#";
        for (size_t k = 0; k < neg.size(); ++k) {
          if (k) c.buf += ' ';
          c.num(neg[k]);
        }
        stamp += 60 + (int64_t)(g() % 7140);
        const int day = 1 + (int)((stamp / 86400) % 28), hh = (int)((stamp / 3600) % 24);
        snprintf(ts, sizeof(ts), "#TAB#01/%02d/2019 %02d:%02d:%02d %s#TAB#x", day, hh % 12 ? hh % 12 : 12, (int)((stamp / 60) % 60),
                 (int)(stamp % 60), hh < 12 ? "AM" : "PM");
        c.buf += ts;
      }
      c.buf += period ? '\n' : '\t';
    }
    c.flush();
  }
  c.close();
  return 0;
}
