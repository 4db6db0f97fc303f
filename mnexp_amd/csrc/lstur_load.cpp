// Native loaders for the three files the LSTUR path reads (host C++; include/lstur_data.h) -- what the reference does in
// Python at task construction, restated so that a 1 M-user / 200 k-doc set loads in seconds instead of minutes:
//   DocMeta.tsv    Seq2Vec._load_docs             task/seq2vec.py:85-110  + document.parse_document / pad_document(1, T)
//                                                                           (document.py:23-54)
//   ClickData.tsv  Seq2VecPaper._load_data        task/paper.py:20-35     + Impression (:7-18: pos ids, neg ids, time)
//   Vocab.tsv      utils.load_textual_embedding   utils.py:34-52
// Same results as the Python mirror (mnexp_amd/task/*.py, pinned by tests/golden/ against the reference itself), same
// failures: where the reference raises (int('') on an empty token, a line with too few columns, an unknown doc id, a
// malformed time stamp, no usable Vocab row) the loader returns -1 with the line number in lstur_data_last_error().
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "../../include/lstur_data.h"

int lstur_data_fail(const std::string& m);   // (lstur_data.cpp: sets the thread-local message, returns -1)

namespace {

struct Span {
  const char* p;
  size_t n;
  bool starts_with(const char* s) const {
    const size_t k = strlen(s);
    return n >= k && memcmp(p, s, k) == 0;
  }
};

// the whole file in memory (+ a trailing '\n' so that every line is terminated)
bool read_file(const char* path, std::string* out, std::string* why) {
  FILE* f = fopen(path, "rb");
  if (!f) {
    *why = std::string("cannot open ") + path + ": " + strerror(errno);
    return false;
  }
  fseek(f, 0, SEEK_END);
  const long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  out->resize((size_t)(sz < 0 ? 0 : sz));
  const size_t got = sz > 0 ? fread(&(*out)[0], 1, (size_t)sz, f) : 0;
  fclose(f);
  out->resize(got);
  if (out->empty() || out->back() != '\n') out->push_back('\n');
  return true;
}

// str.split(sep) for a multi-character separator (every occurrence splits; empty pieces are kept)
void split(Span s, const char* sep, std::vector<Span>* out) {
  out->clear();
  const size_t k = strlen(sep);
  const char* p = s.p;
  const char* end = s.p + s.n;
  if (k == 1) {                       // (the token and column separators: memchr, no needle set-up per call)
    for (;;) {
      const char* q = (const char*)memchr(p, sep[0], (size_t)(end - p));
      if (!q) {
        out->push_back(Span{p, (size_t)(end - p)});
        return;
      }
      out->push_back(Span{p, (size_t)(q - p)});
      p = q + 1;
    }
  }
  for (;;) {
    const char* q = (const char*)memmem(p, (size_t)(end - p), sep, k);
    if (!q) {
      out->push_back(Span{p, (size_t)(end - p)});
      return;
    }
    out->push_back(Span{p, (size_t)(q - p)});
    p = q + k;
  }
}
inline bool py_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }
Span strip(Span s) {
  while (s.n && py_space(s.p[0])) ++s.p, --s.n;
  while (s.n && py_space(s.p[s.n - 1])) --s.n;
  return s;
}
// int(token): optional surrounding whitespace, optional sign, decimal digits ('_' between digits accepted like Python 3.6+)
bool py_int(Span s, int64_t* out) {
  s = strip(s);
  if (!s.n) return false;
  bool neg = false;
  size_t i = 0;
  if (s.p[0] == '+' || s.p[0] == '-') neg = s.p[0] == '-', i = 1;
  if (i >= s.n) return false;
  int64_t v = 0;
  bool prev_digit = false;
  for (; i < s.n; ++i) {
    const char c = s.p[i];
    if (c == '_') {
      if (!prev_digit || i + 1 >= s.n) return false;
      prev_digit = false;
      continue;
    }
    if (c < '0' || c > '9') return false;
    if (v > (INT64_MAX - 9) / 10) return false;
    v = v * 10 + (c - '0');
    prev_digit = true;
  }
  *out = neg ? -v : v;
  return true;
}
// [int(k) for k in text.split(' ')]  (an empty token is an error, as int('') is)
bool int_list(Span s, std::vector<int64_t>* out, std::vector<Span>* tmp) {
  split(s, " ", tmp);
  out->clear();
  for (const Span& t : *tmp) {
    int64_t v;
    if (!py_int(t, &v)) return false;
    out->push_back(v);
  }
  return true;
}
std::string at(const char* path, int64_t line) { return std::string(path) + " line " + std::to_string(line) + ": "; }

// days since 1970-01-01 of a proleptic Gregorian date
int64_t days_from_civil(int64_t y, int m, int d) {
  y -= m <= 2;
  const int64_t era = (y >= 0 ? y : y - 399) / 400;
  const int64_t yoe = y - era * 400;
  const int64_t doy = (153 * (m + (m > 2 ? -3 : 9)) + 2) / 5 + d - 1;
  const int64_t doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
  return era * 146097 + doe - 719468;
}
bool num(const char*& p, const char* end, int max_digits, int* out) {
  int v = 0, n = 0;
  while (p < end && n < max_digits && *p >= '0' && *p <= '9') v = v * 10 + (*p++ - '0'), ++n;
  *out = v;
  return n > 0;
}
// datetime.strptime(s, '%m/%d/%Y %I:%M:%S %p') -> seconds since 1970-01-01 (naive)
bool parse_time(Span s, int64_t* out) {
  const char* p = s.p;
  const char* end = s.p + s.n;
  int mo, d, y, h, mi, se;
  if (!num(p, end, 2, &mo) || p >= end || *p++ != '/') return false;
  if (!num(p, end, 2, &d) || p >= end || *p++ != '/') return false;
  if (!num(p, end, 4, &y) || p >= end || *p != ' ') return false;
  while (p < end && *p == ' ') ++p;
  if (!num(p, end, 2, &h) || p >= end || *p++ != ':') return false;
  if (!num(p, end, 2, &mi) || p >= end || *p++ != ':') return false;
  if (!num(p, end, 2, &se) || p >= end || *p != ' ') return false;
  while (p < end && *p == ' ') ++p;
  if (end - p != 2 || (p[1] != 'M' && p[1] != 'm')) return false;
  const bool pm = p[0] == 'P' || p[0] == 'p';
  if (!pm && p[0] != 'A' && p[0] != 'a') return false;
  static const int mdays[] = {31, 29, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
  if (mo < 1 || mo > 12 || d < 1 || d > mdays[mo - 1] || y < 1 || h < 1 || h > 12 || mi > 59 || se > 61) return false;
  if (mo == 2 && d == 29 && !((y % 4 == 0 && y % 100 != 0) || y % 400 == 0)) return false;
  const int h24 = (h % 12) + (pm ? 12 : 0);
  *out = days_from_civil(y, mo, d) * 86400 + h24 * 3600 + mi * 60 + se;
  return true;
}

}  // namespace

// ---- DocMeta.tsv ------------------------------------------------------------------------------------------------------
struct lstur_docs {
  int32_t doc_count = 0, T = 0;
  std::vector<int32_t> titles;     // [doc_count, T]
  std::vector<uint8_t> present;    // [doc_count]
};
// ---- ClickData.tsv ----------------------------------------------------------------------------------------------------
struct ClickSet {                  // impressions of one period (train | valid) in CSR form
  std::vector<int64_t> imp_off{0}, pos_off{0}, neg_off{0}, time;
  std::vector<int32_t> pos, neg;
};
struct lstur_clicks {
  int32_t n_users = 0;
  int32_t timed = 1;               // every impression carries a time stamp
  ClickSet set[2];
};

extern "C" {

int lstur_docs_load(const char* path, int32_t title_len, lstur_docs** out) {
  if (!path || !out || title_len < 1) return lstur_data_fail("docs_load: bad argument");
  std::string text, why;
  if (!read_file(path, &text, &why)) return lstur_data_fail(why);
  struct Row { int64_t idx; size_t off; };
  std::vector<Row> rows;
  std::vector<int32_t> arena;
  std::vector<Span> cols, segs, toks;
  std::vector<int64_t> ids;
  int64_t lineno = 0, max_idx = -1;
  const char* p = text.data();
  const char* end = p + text.size();
  while (p < end) {
    const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
    ++lineno;
    Span line{p, (size_t)(nl - p)};
    p = nl + 1;
    split(line, "\t", &cols);
    if (cols.size() < 5) return lstur_data_fail(at(path, lineno) + "fewer than 5 tab-separated columns");
    int64_t idx;
    if (!py_int(cols[1], &idx) || idx < 0 || idx >= INT32_MAX)
      return lstur_data_fail(at(path, lineno) + "column 1 is not a doc index");
    // parse_document: every '#N#' segment is parsed (an empty token raises); pad_document(1, T) keeps the first one
    split(cols[4], "#N#", &segs);
    const size_t off = arena.size();
    arena.resize(off + title_len, 0);
    for (size_t s = 0; s < segs.size(); ++s) {
      if (!int_list(segs[s], &ids, &toks)) return lstur_data_fail(at(path, lineno) + "title: invalid literal for int()");
      if (s == 0)
        for (size_t t = 0; t < ids.size() && t < (size_t)title_len; ++t) arena[off + t] = (int32_t)ids[t];
    }
    rows.push_back(Row{idx, off});
    if (idx > max_idx) max_idx = idx;
  }
  if (rows.empty()) return lstur_data_fail(std::string(path) + ": no document rows");
  lstur_docs* d = new lstur_docs();
  d->doc_count = (int32_t)(max_idx + 1);
  d->T = title_len;
  d->titles.assign((size_t)d->doc_count * title_len, 0);
  d->present.assign(d->doc_count, 0);
  for (const Row& r : rows) {                                  // a later line with the same index wins (dict assignment)
    memcpy(&d->titles[(size_t)r.idx * title_len], &arena[r.off], (size_t)title_len * sizeof(int32_t));
    d->present[r.idx] = 1;
  }
  memset(&d->titles[0], 0, (size_t)title_len * sizeof(int32_t));   // doc 0 = the all-zero pad title (seq2vec.py:106-108)
  d->present[0] = 1;
  *out = d;
  return 0;
}
int32_t lstur_docs_count(const lstur_docs* d) { return d ? d->doc_count : -1; }
const int32_t* lstur_docs_titles(const lstur_docs* d) { return d ? d->titles.data() : nullptr; }
const uint8_t* lstur_docs_present(const lstur_docs* d) { return d ? d->present.data() : nullptr; }
int lstur_docs_destroy(lstur_docs* d) {
  delete d;
  return 0;
}

// one chunk of whole lines -> its own CSR pair (offsets relative to the chunk); returns "" or the error (line = 1-based
// line number inside the chunk)
static std::string parse_click_chunk(const char* p, const char* end, const uint8_t* present, int32_t doc_count, ClickSet (&set)[2],
                                     int32_t* n_users, int32_t* timed, int64_t* err_line) {
  std::vector<Span> cols, imps, fields, toks;
  std::vector<int64_t> ids;
  int64_t lineno = 0;
  auto bad = [&](const char* m) {
    *err_line = lineno;
    return std::string(m);
  };
  while (p < end) {
    const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
    ++lineno;
    Span line{p, (size_t)(nl - p)};
    p = nl + 1;
    split(line, "\t", &cols);
    if (cols.size() < 4) return bad("fewer than 4 tab-separated columns");
    for (int which = 0; which < 2; ++which) {
      ClickSet& s = set[which];
      const Span col = cols[2 + which];
      if (col.n) {                                             // `if cols[k] else []`
        split(col, "#N#", &imps);
        for (const Span& d : imps) {
          if (d.starts_with("#TAB#")) continue;                // impressions without positives are dropped (paper.py:21)
          split(d, "#TAB#", &fields);
          if (fields.size() < 2) return bad("impression without a '#TAB#' negatives field");
          for (int f = 0; f < 2; ++f) {
            if (!int_list(fields[f], &ids, &toks)) return bad("impression: invalid literal for int()");
            std::vector<int32_t>& dst = f ? s.neg : s.pos;
            for (int64_t v : ids) {
              if (present && (v < 0 || v >= doc_count || !present[v])) {
                *err_line = lineno;
                return "unknown doc id " + std::to_string(v);
              }
              if (v < INT32_MIN || v > INT32_MAX) return bad("doc id out of range");
              dst.push_back((int32_t)v);
            }
          }
          int64_t t = INT64_MIN;
          if (fields.size() >= 3) {
            if (!parse_time(fields[2], &t)) return bad("time data does not match format '%m/%d/%Y %I:%M:%S %p'");
          } else {
            *timed = 0;
          }
          s.time.push_back(t);
          s.pos_off.push_back((int64_t)s.pos.size());
          s.neg_off.push_back((int64_t)s.neg.size());
        }
      }
      s.imp_off.push_back((int64_t)s.time.size());
    }
    ++*n_users;
  }
  return std::string();
}

int lstur_clicks_load(const char* path, const uint8_t* present, int32_t doc_count, lstur_clicks** out) {
  if (!path || !out) return lstur_data_fail("clicks_load: bad argument");
  std::string text, why;
  if (!read_file(path, &text, &why)) return lstur_data_fail(why);
  // (an empty line -- also a trailing one -- has fewer than 4 columns: the reference raises IndexError there)
  const char* begin = text.data();
  const char* end = begin + text.size();
  // chunks of whole lines, one per hardware thread (a 1 M-user file is ~3 GB of text)
  unsigned nt = std::thread::hardware_concurrency();
  nt = nt ? std::min(nt, 32u) : 4u;
  if ((size_t)(end - begin) < ((size_t)1 << 22)) nt = 1;
  std::vector<const char*> cut{begin};
  for (unsigned t = 1; t < nt; ++t) {
    const char* q = begin + (size_t)(end - begin) * t / nt;
    if (q <= cut.back()) continue;
    const char* nl = (const char*)memchr(q, '\n', (size_t)(end - q));
    if (!nl || nl + 1 >= end) break;
    cut.push_back(nl + 1);
  }
  cut.push_back(end);
  const size_t nc = cut.size() - 1;
  struct Part {
    ClickSet set[2];
    int32_t users = 0, timed = 1;
    int64_t err_line = 0;
    std::string err;
  };
  std::vector<Part> parts(nc);
  std::vector<std::thread> th;
  for (size_t k = 0; k < nc; ++k)
    th.emplace_back([&, k]() {
      Part& P = parts[k];
      const size_t bytes = (size_t)(cut[k + 1] - cut[k]);      // (one allocation per array instead of a doubling chain)
      for (int w = 0; w < 2; ++w) {
        const size_t share = w ? bytes / 5 : bytes;             // the training period holds most of a line
        P.set[w].pos.reserve(share / 16);
        P.set[w].neg.reserve(share / 9);
        P.set[w].time.reserve(share / 80);
        P.set[w].pos_off.reserve(share / 80);
        P.set[w].neg_off.reserve(share / 80);
      }
      P.err = parse_click_chunk(cut[k], cut[k + 1], present, doc_count, P.set, &P.users, &P.timed, &P.err_line);
    });
  for (auto& t : th) t.join();
  int64_t lines_before = 0;
  for (size_t k = 0; k < nc; ++k) {                            // the first failure in file order, with its absolute line
    if (!parts[k].err.empty()) return lstur_data_fail(at(path, lines_before + parts[k].err_line) + parts[k].err);
    lines_before += parts[k].users;
  }
  lstur_clicks* c = new lstur_clicks();
  for (int which = 0; which < 2; ++which) {
    ClickSet& s = c->set[which];
    size_t n_imp = 0, n_pos = 0, n_neg = 0, n_usr = 0;
    for (const Part& P : parts) n_imp += P.set[which].time.size(), n_pos += P.set[which].pos.size(), n_neg += P.set[which].neg.size(), n_usr += (size_t)P.users;
    s.imp_off.reserve(n_usr + 1);
    s.pos_off.reserve(n_imp + 1);
    s.neg_off.reserve(n_imp + 1);
    s.time.reserve(n_imp);
    s.pos.reserve(n_pos);
    s.neg.reserve(n_neg);
    for (Part& P : parts) {
      const ClickSet& q = P.set[which];
      const int64_t i0 = (int64_t)s.time.size(), p0 = (int64_t)s.pos.size(), n0 = (int64_t)s.neg.size();
      for (size_t i = 1; i < q.imp_off.size(); ++i) s.imp_off.push_back(i0 + q.imp_off[i]);
      for (size_t i = 1; i < q.pos_off.size(); ++i) s.pos_off.push_back(p0 + q.pos_off[i]);
      for (size_t i = 1; i < q.neg_off.size(); ++i) s.neg_off.push_back(n0 + q.neg_off[i]);
      s.time.insert(s.time.end(), q.time.begin(), q.time.end());
      s.pos.insert(s.pos.end(), q.pos.begin(), q.pos.end());
      s.neg.insert(s.neg.end(), q.neg.begin(), q.neg.end());
      P.set[which] = ClickSet();                                // (free the part as soon as it is merged)
    }
  }
  for (const Part& P : parts) {
    c->n_users += P.users;
    c->timed &= P.timed;
  }
  if (c->n_users == 0) {
    delete c;
    return lstur_data_fail(std::string(path) + ": no user rows");
  }
  *out = c;
  return 0;
}
int32_t lstur_clicks_users(const lstur_clicks* c) { return c ? c->n_users : -1; }
int32_t lstur_clicks_timed(const lstur_clicks* c) { return c ? c->timed : -1; }
int lstur_clicks_arrays(const lstur_clicks* c, int32_t which, const int64_t** imp_off, const int64_t** pos_off,
                        const int32_t** pos, const int64_t** neg_off, const int32_t** neg, const int64_t** time,
                        int64_t* n_imp, int64_t* n_pos, int64_t* n_neg) {
  if (!c || which < 0 || which > 1) return lstur_data_fail("clicks_arrays: bad argument");
  const ClickSet& s = c->set[which];
  if (imp_off) *imp_off = s.imp_off.data();
  if (pos_off) *pos_off = s.pos_off.data();
  if (pos) *pos = s.pos.data();
  if (neg_off) *neg_off = s.neg_off.data();
  if (neg) *neg = s.neg.data();
  if (time) *time = s.time.data();
  if (n_imp) *n_imp = (int64_t)s.time.size();
  if (n_pos) *n_pos = (int64_t)s.pos.size();
  if (n_neg) *n_neg = (int64_t)s.neg.size();
  return 0;
}
int lstur_clicks_destroy(lstur_clicks* c) {
  delete c;
  return 0;
}

// Vocab.tsv -> float32 [*rows, dim] (malloc'ed: lstur_vocab_free).  Rows whose vector does not have exactly dim - 1 spaces are
// skipped; missing indices (row 0 included) are zero; values go through double like Python's float() -> np.float32.
int lstur_vocab_load(const char* path, int32_t dim, float** matrix, int32_t* rows_out) {
  if (!path || !matrix || !rows_out || dim < 1) return lstur_data_fail("vocab_load: bad argument");
  std::string text, why;
  if (!read_file(path, &text, &why)) return lstur_data_fail(why);
  struct Row { int64_t idx; size_t off; };
  std::vector<Row> rows;
  std::vector<float> arena;
  std::vector<Span> cols, toks;
  int64_t lineno = 0, max_idx = -1;
  std::string tok;
  const char* p = text.data();
  const char* end = p + text.size();
  while (p < end) {
    const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
    ++lineno;
    Span line = strip(Span{p, (size_t)(nl - p)});              // line.strip()
    p = nl + 1;
    split(line, "\t", &cols);
    const Span vec = cols.back();
    int64_t spaces = 0;
    for (size_t i = 0; i < vec.n; ++i) spaces += vec.p[i] == ' ';
    if (spaces != dim - 1) continue;
    if (cols.size() < 2) return lstur_data_fail(at(path, lineno) + "no index column in front of the vector");
    int64_t idx;
    if (!py_int(cols[cols.size() - 2], &idx) || idx < 0 || idx >= INT32_MAX)
      return lstur_data_fail(at(path, lineno) + "invalid literal for int() in the index column");
    split(vec, " ", &toks);
    const size_t off = arena.size();
    arena.resize(off + dim);
    for (int j = 0; j < dim; ++j) {
      const Span t = strip(toks[j]);
      if (!t.n) return lstur_data_fail(at(path, lineno) + "could not convert string to float");
      tok.assign(t.p, t.n);
      char* e = nullptr;
      const double v = strtod(tok.c_str(), &e);
      if (e != tok.c_str() + tok.size()) return lstur_data_fail(at(path, lineno) + "could not convert string to float");
      arena[off + j] = (float)v;
    }
    rows.push_back(Row{idx, off});
    if (idx > max_idx) max_idx = idx;
  }
  if (rows.empty()) return lstur_data_fail(std::string(path) + ": no row with a " + std::to_string(dim) + "-d vector");
  float* m = (float*)calloc((size_t)(max_idx + 1) * dim, sizeof(float));
  if (!m) return lstur_data_fail("vocab_load: out of memory");
  for (const Row& r : rows) memcpy(m + (size_t)r.idx * dim, &arena[r.off], (size_t)dim * sizeof(float));
  *matrix = m;
  *rows_out = (int32_t)(max_idx + 1);
  return 0;
}
void lstur_vocab_free(float* matrix) { free(matrix); }

}  // extern "C"
