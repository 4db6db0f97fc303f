/*
 * examples/lstur_step.c -- the C-ABI of liblstur_hip.so (include/lstur.h) driven from plain C, no Python, no torch:
 * the caller owns every device buffer (hipMalloc) and the stream; the library only launches kernels.
 *
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/lstur_step.c \
 *       -Lmnexp_amd -llstur_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/mnexp_amd -Wl,-rpath,/opt/rocm/lib -lm -o lstur_step
 *   ./lstur_step            # 20 training steps of LSTUR-ini on random ids; prints the loss every 5 steps
 *
 * What a reference maintainer would call from `_build_model` / `fit_generator` (paper.py:635-665, main.py:73-78):
 * lstur_query_sizes -> allocate -> lstur_create -> per batch lstur_forward_backward + lstur_adam_step.
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lstur.h"

#define HIP(x)                                                                      \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                       \
      return 2;                                                                     \
    }                                                                               \
  } while (0)
#define LSTUR(x)                                                                    \
  do {                                                                              \
    int rc_ = (x);                                                                  \
    if (rc_ != LSTUR_OK) {                                                          \
      fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, lstur_last_error());         \
      return 3;                                                                     \
    }                                                                               \
  } while (0)

static uint32_t rng_state = 12345u;
static uint32_t rnd(void) {   /* xorshift32 */
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 17;
  rng_state ^= rng_state << 5;
  return rng_state;
}
static float unif(float a) { return ((rnd() >> 8) * (1.0f / 16777216.0f) * 2.0f - 1.0f) * a; }

static int upload_f(float* dev, size_t n, float amp) {
  float* h = (float*)malloc(n * sizeof(float));
  for (size_t i = 0; i < n; ++i) h[i] = unif(amp);
  hipError_t e = hipMemcpy(dev, h, n * sizeof(float), hipMemcpyHostToDevice);
  free(h);
  return e == hipSuccess ? 0 : 1;
}

int main(void) {
  lstur_cfg cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.struct_size = sizeof(cfg);   /* ABI guard: the library refuses a lstur_cfg of another size */
  cfg.max_batch = 16; cfg.window = 10; cfg.title_len = 12; cfg.cands = 5; cfg.vocab = 2000; cfg.word_dim = 64;
  cfg.filters = 96; cfg.kernel = 3; cfg.user_dim = 48; cfg.n_users = 300; cfg.arch = LSTUR_ARCH_IGRU;
  cfg.dropout = 0.2f; cfg.seed = 1; cfg.beta1 = 0.9f; cfg.beta2 = 0.999f; cfg.eps = 1e-7f;
  cfg.head = LSTUR_HEAD_SOFTMAX; cfg.gain = 1.0f; cfg.loss_negatives = 4; cfg.id_keep = 1.0f;

  lstur_sizes sz;
  LSTUR(lstur_query_sizes(&cfg, &sz));
  printf("%s: dense bucket %zu floats, workspace %.1f MiB\n", lstur_version(), sz.dense, sz.workspace_bytes / 1048576.0);

  lstur_buffers buf;
  memset(&buf, 0, sizeof(buf));
  HIP(hipMalloc((void**)&buf.word_emb, sz.word_emb * 4));
  HIP(hipMalloc((void**)&buf.dense_param, sz.dense * 4));
  HIP(hipMalloc((void**)&buf.dense_grad, sz.dense * 4));
  HIP(hipMalloc((void**)&buf.dense_m, sz.dense * 4));
  HIP(hipMalloc((void**)&buf.dense_v, sz.dense * 4));
  HIP(hipMalloc((void**)&buf.user_emb, sz.user_emb * 4));
  HIP(hipMalloc((void**)&buf.user_m, sz.user_emb * 4));
  HIP(hipMalloc((void**)&buf.user_v, sz.user_emb * 4));
  HIP(hipMalloc(&buf.workspace, sz.workspace_bytes));
  buf.workspace_bytes = sz.workspace_bytes;
  HIP(hipMemset(buf.dense_grad, 0, sz.dense * 4));
  HIP(hipMemset(buf.dense_m, 0, sz.dense * 4));
  HIP(hipMemset(buf.dense_v, 0, sz.dense * 4));
  HIP(hipMemset(buf.user_m, 0, sz.user_emb * 4));
  HIP(hipMemset(buf.user_v, 0, sz.user_emb * 4));
  HIP(hipMemset(buf.dense_param, 0, sz.dense * 4));
  if (upload_f(buf.word_emb, sz.word_emb, 0.1f) || upload_f(buf.user_emb, sz.user_emb, 0.05f)) return 2;
  /* weights by name: offsets inside the caller-owned bucket (biases stay 0) */
  const char* mats[] = {"conv_w", "att_w", "dense_w", "gru_w", "gru_u"};
  for (int i = 0; i < 5; ++i) {
    size_t off, cnt;
    LSTUR(lstur_param(&cfg, mats[i], &off, &cnt));
    if (upload_f(buf.dense_param + off, cnt, 0.08f)) return 2;
  }

  lstur_ctx* ctx;
  LSTUR(lstur_create(&cfg, &buf, &ctx));
  hipStream_t stream;
  HIP(hipStreamCreate(&stream));

  const int B = cfg.max_batch, W = cfg.window, T = cfg.title_len, K1 = cfg.cands;
  const size_t nh = (size_t)B * W * T, nc = (size_t)B * K1 * T;
  int32_t *h_user = (int32_t*)malloc(B * 4), *h_hist = (int32_t*)malloc(nh * 4), *h_cand = (int32_t*)malloc(nc * 4);
  int32_t *d_user, *d_hist, *d_cand;
  float* d_stats;
  HIP(hipMalloc((void**)&d_user, B * 4));
  HIP(hipMalloc((void**)&d_hist, nh * 4));
  HIP(hipMalloc((void**)&d_cand, nc * 4));
  HIP(hipMalloc((void**)&d_stats, 4 * sizeof(float)));
  HIP(hipMemset(d_stats, 0, 4 * sizeof(float)));

  float first = 0.f, last = 0.f;
  for (int step = 0; step < 20; ++step) {
    /* one fixed batch (so the loss must fall): token ids in [1, vocab), titles padded with 0 at the end */
    rng_state = 777u;
    for (int b = 0; b < B; ++b) h_user[b] = (int32_t)(rnd() % cfg.n_users);
    for (size_t i = 0; i < nh; ++i) h_hist[i] = ((int)(i % T) < 8) ? (int32_t)(1 + rnd() % (cfg.vocab - 1)) : 0;
    for (size_t i = 0; i < nc; ++i) h_cand[i] = ((int)(i % T) < 9) ? (int32_t)(1 + rnd() % (cfg.vocab - 1)) : 0;
    HIP(hipMemcpyAsync(d_user, h_user, B * 4, hipMemcpyHostToDevice, stream));
    HIP(hipMemcpyAsync(d_hist, h_hist, nh * 4, hipMemcpyHostToDevice, stream));
    HIP(hipMemcpyAsync(d_cand, h_cand, nc * 4, hipMemcpyHostToDevice, stream));
    LSTUR(lstur_forward_backward(ctx, d_user, d_hist, d_cand, B, (uint32_t)step, 1.0f, d_stats, stream));
    LSTUR(lstur_adam_step(ctx, 1e-2f, step + 1, NULL, NULL, 0, stream));
    float st[4];
    HIP(hipMemcpyAsync(st, d_stats, sizeof(st), hipMemcpyDeviceToHost, stream));
    HIP(hipStreamSynchronize(stream));
    if (step == 0) first = st[0];
    last = st[0];
    if (step % 5 == 0 || step == 19) printf("step %2d  loss %.5f  accuracy %.3f\n", step, st[0], st[1]);
  }
  LSTUR(lstur_destroy(ctx));
  if (!(isfinite(last) && last < first)) {
    fprintf(stderr, "loss did not fall: %.5f -> %.5f\n", first, last);
    return 4;
  }
  printf("ok: loss %.5f -> %.5f\n", first, last);
  return 0;
}
