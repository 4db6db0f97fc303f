"""where does the end-to-end leg's time go?  (diagnostics)"""
import os, sys, time, contextlib, io, tempfile, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from mnexp_amd import synth, settings, task as T
from mnexp_amd.data_native import NativeTrainFeed
from mnexp_amd.native_model import NativeModel
wl = bench.WORKLOADS['lstur_ini_b64']
titles0, word, _ = bench.synth_inputs(wl, np.random.default_rng(1234), 0)
d = tempfile.mkdtemp()
rng = np.random.default_rng(0)
topic, titles = synth.make_docs(wl['N_d'], wl['V'], wl['T'], rng)
synth.write_docmeta(os.path.join(d, 'DocMeta.tsv'), titles)
synth.write_clickdata(os.path.join(d, 'ClickData.tsv'), 6000, topic, 20, 4, rng)
open(os.path.join(d, 'Vocab.tsv'), 'w').close()
settings.public_config = dict(input_training_data_path=d, input_validation_data_path=d, input_previous_model_path=d, output_model_path=d, log_dir=d)
base = dict(task='Seq2VecPaperSoftmaxId', arch=wl['arch'], batch_size=wl['B'], window_size=wl['W'], title_shape=wl['T'], body_shape=4, negative_samples=wl['K1'] - 1, training_step=100, validation_step=1, max_impression=200, news_encoder='cnnatt', textual_embedding_dim=wl['D'], days=30, debug=False, epochs=1, validation_impression=10, testing_impression=10, learning_rate=1e-3, learning_rate_decay=0.5, dropout=0.2, user_embedding_dim=wl['U'], title_filter_shape=(wl['F'], wl['k']), score_model='dot', enable_pretrain_encoder=False, textual_embedding_trainable=False, name='', gain=1.0, seed=1, ids=True, native_feed=True, background=True, pretrain_name='')
with contextlib.redirect_stdout(io.StringIO()):
    cfg = settings.Config(base)
h = T.get(cfg)
model = NativeModel(word, len(h.data), wl['B'], wl['W'], wl['T'], wl['K1'] - 1, wl['F'], wl['k'], wl['U'], arch=wl['arch'], dropout=0.2, lr=1e-3, seed=7, device='cuda:0', title_table=h.docs.title_table)
feed = NativeTrainFeed(h, seed=1)
batches = [next(feed)[0] for _ in range(64)]
n = 200
# (a) device-resident copies of the same batches: GPU-bound time of these shapes
dev = [tuple(model._dev_i32(np.asarray(a)) for a in b) for b in batches]
for k in range(10):
    model._train_step(*dev[k % 64], True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(n):
    u, hh, c = dev[k % 64]
    model.prefetch_ids(dev[(k + 1) % 64][1], dev[(k + 1) % 64][2])
    model._train_step(u, hh, c, True)
torch.cuda.synchronize()
print('device-resident real-history batches: %.1f us/step' % ((time.perf_counter() - t0) / n * 1e6))
# (b) numpy batches through train_on_ids (H2D inside)
staged = None
for k in range(10):
    staged = model.train_on_ids(*batches[k % 64], next_ids=batches[(k + 1) % 64], staged=staged)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(n):
    staged = model.train_on_ids(*batches[k % 64], next_ids=batches[(k + 1) % 64], staged=staged)
t1 = time.perf_counter()
torch.cuda.synchronize()
print('train_on_ids on host batches: host %.1f us/step, wall %.1f us/step' % ((t1 - t0) / n * 1e6, (time.perf_counter() - t0) / n * 1e6))
# (c) + the feed in the loop
X, _ = next(feed)
t0 = time.perf_counter()
for k in range(n):
    Xn, _ = next(feed)
    staged = model.train_on_ids(*X, next_ids=Xn, staged=staged)
    X = Xn
torch.cuda.synchronize()
print('feed + train_on_ids: wall %.1f us/step' % ((time.perf_counter() - t0) / n * 1e6))
prof_t = model.profile(1)
for k in range(20):
    model._train_step(*dev[k % 64], True)
torch.cuda.synchronize()
print({k: round(v[0] * 1e3 / 20, 1) for k, v in model.profile_read().items()})
