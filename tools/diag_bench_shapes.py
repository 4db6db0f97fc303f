"""Errors of one step's logits / gradients at a bench workload's exact shapes against the fp32 and the fp64 oracle, per conv
scheme (f16x2 default, bf16x3, exact fp32).   python tools/diag_bench_shapes.py [workload]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main(name='lstur_con_b64'):
    import bench
    from oracle import lstur_oracle as O
    from mnexp_amd.native_model import NativeModel
    from mnexp_amd._lib import lib, check
    wl = bench.WORKLOADS[name]
    titles, word, batches = bench.synth_inputs(wl, np.random.default_rng(1234), 1)
    enc = wl.get('encoder', 'cnnatt')
    rng = np.random.default_rng(5)
    p = O.init_params(rng, V=wl['V'], D=wl['D'], F=wl['F'], k=wl['k'], U=wl['U'], N_u=wl['N_u'], arch=wl['arch'],
                      word_emb=word, dtype=np.float32, encoder=enc, heads=wl.get('heads', 1))
    for n in (('qkv_b' if enc == 'mhsa' else 'conv_b'), 'att_b', 'dense_b', 'gru_b'):
        p[n] = rng.normal(0, 0.1, size=p[n].shape).astype(np.float32)
    user, hist, cands = batches[0]
    refs = {}
    for tag, dt in (('f32', np.float32), ('f64', np.float64)):
        q = {n: (a.astype(dt) if isinstance(a, np.ndarray) and a.dtype.kind == 'f' else a) for n, a in p.items()}
        refs[tag] = O.forward_backward(q, wl['arch'], user, titles[hist], titles[cands], drop_p=0.2, seed=7, step=0)
    names = O.dense_param_names(wl['arch'], encoder=enc)
    print('%-10s' % 'oracle f32 vs f64:', ' '.join('%s %.1e' % (n, np.abs(refs['f32']['grads'][n] - refs['f64']['grads'][n]).max() /
                                                                   np.abs(refs['f64']['grads'][n]).max()) for n in names))
    for env in ({}, {'LSTUR_CONV_SCHEME': 'bf16x3'}, {'LSTUR_CONV_F32': '1'}, {'LSTUR_CAND_SPLIT': '0'}, {'LSTUR_WG_STREAM': '0'}):
        for k in ('LSTUR_CONV_SCHEME', 'LSTUR_CONV_F32', 'LSTUR_CAND_SPLIT', 'LSTUR_WG_STREAM'):
            os.environ.pop(k, None)
        os.environ.update(env)
        model = NativeModel(word, wl['N_u'], wl['B'], wl['W'], wl['T'], wl['K1'] - 1, wl['F'], wl['k'], wl['U'], arch=wl['arch'],
                            dropout=0.2, lr=1e-3, seed=7, title_table=titles, news_encoder=enc, heads=wl.get('heads', 0))
        model.set_weights({n: p[n] for n in p if n not in ('word_emb', 'mhsa_heads')})
        u, h, c = (model._dev_i32(a) for a in (user, hist, cands))
        check(lib.lstur_forward_backward_ids(model._ctx, u.data_ptr(), h.data_ptr(), c.data_ptr(), wl['B'], 0, 1.0,
                                             model.stats.data_ptr(), model._stream()), 'fb')
        torch.cuda.synchronize()
        g = model.get_grads()
        logits = model.peek('logits').cpu().numpy().reshape(wl['B'], wl['K1'])
        ref = refs['f64']
        print('%-28s logits %.1e |' % (str(env) + ' ' + model.conv_scheme, np.abs(logits - ref['logits']).max() / np.abs(ref['logits']).max()),
              ' '.join('%s %.1e' % (n, np.abs(g[n] - ref['grads'][n]).max() / np.abs(ref['grads'][n]).max()) for n in names),
              'rows %.1e' % (np.abs(g['user_row_grads'] - ref['user_row_grads']).max() / np.abs(ref['user_row_grads']).max()))
        del model


if __name__ == '__main__':
    main(*sys.argv[1:])
