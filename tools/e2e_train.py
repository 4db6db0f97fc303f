"""End-to-end training run of the drop-in CLI path on a synthetic data set in the reference's file formats
(mnexp_amd.synth preset, default a quarter-size cfg2): data generation, task loading, N epochs through
`run_training` with the chosen feed; prints wall-clock impressions/s per epoch (host pipeline included) and the
per-epoch validation metrics of the task callback.  Usage: python tools/e2e_train.py [--feed native|ids|titles]"""
import argparse, logging, os, sys, tempfile, time
sys.path.insert(0, '.')
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--feed', default='native', choices=['native', 'ids', 'titles'])
    ap.add_argument('--users', type=int, default=12000)
    ap.add_argument('--epochs', type=int, default=3)
    ap.add_argument('--steps', type=int, default=400)
    ap.add_argument('--task', default='Seq2VecPaperSoftmaxId')
    ap.add_argument('--arch', default='igru')
    args = ap.parse_args()
    from mnexp_amd import synth, settings
    from mnexp_amd.main import run_training
    logging.basicConfig(level=logging.WARNING)
    d = tempfile.mkdtemp(prefix='lstur_e2e_')
    t0 = time.time()
    synth.generate(d, 'cfg2', seed=0, n_users=args.users)
    print('generated %d users in %.1f s -> %s' % (args.users, time.time() - t0, d), flush=True)
    settings.public_config = dict(input_training_data_path=d, input_validation_data_path=d, input_previous_model_path=d,
                                  output_model_path=d, log_dir=d)
    base = dict(task=args.task, arch=args.arch, batch_size=64, window_size=50, title_shape=30, body_shape=4,
                negative_samples=4, training_step=args.steps, validation_step=5, max_impression=200, news_encoder='cnnatt',
                textual_embedding_dim=300, days=30, debug=False, epochs=1, validation_impression=200,
                testing_impression=200, learning_rate=1e-3, learning_rate_decay=0.5, dropout=0.2, user_embedding_dim=200,
                title_filter_shape=(400, 3), score_model='dot', enable_pretrain_encoder=False,
                textual_embedding_trainable=False, name='', gain=1.0, seed=1, ids=args.feed != 'titles',
                native_feed=args.feed == 'native', background=True, pretrain_name='')
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = settings.Config(base)
    from mnexp_amd import task as T
    t0 = time.time()
    h = T.get(cfg)
    print('task loaded in %.1f s (%d docs, %d users)' % (time.time() - t0, h.doc_count, len(h.data)), flush=True)
    np.random.seed(0)
    import random
    random.seed(0)
    import torch
    # one epoch at a time so each can be timed; the handler keeps model and optimizer state
    h.build_model(0)
    from mnexp_amd.data_native import NativeTrainFeed
    feed = NativeTrainFeed(h, seed=1) if args.feed == 'native' else (h.train_ids if args.feed == 'ids' else h.train)
    for e in range(args.epochs):
        torch.cuda.synchronize()
        t0 = time.time()
        hist = h.model.fit_generator(feed, args.steps, epochs=e + 1, initial_epoch=e, verbose=0, ids_form=args.feed != 'titles')
        torch.cuda.synchronize()
        el = time.time() - t0
        cfg.epochs = args.epochs
        h.callback(e)
        m = h.last_metrics
        print('epoch %d: %.0f impressions/s wall (%.2f ms/step, feed=%s)  loss %.4f acc %.3f | valid auc %.4f ndcg@10 %.4f mrr %.4f'
              % (e, args.steps * 64 / el, 1e3 * el / args.steps, args.feed, hist.history['loss'][-1],
                 hist.history[h.model.metrics_names[1]][-1], m['auc'], m['ndcgx'], m['mrr']), flush=True)


if __name__ == '__main__':
    main()
