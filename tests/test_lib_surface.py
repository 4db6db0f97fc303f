"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/lstur.h declares, sizes
and parameter layout are queryable without a GPU, and creating a context without a GPU FAILS LOUDLY
(there is no CPU fallback in the product path)."""
import ctypes as C
import os
import re

import pytest

from mnexp_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'lstur.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(lstur_[a-z0-9_]+)\s*\(', text)))


def test_every_declared_symbol_is_exported_and_bound():
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(_lib.lib, s), 'not exported: ' + s
        assert s in _lib.SIGNATURES, 'no ctypes signature: ' + s
    assert sorted(_lib.SIGNATURES) == syms


def test_sizes_and_layout_without_gpu():
    cfg = _lib.make_cfg(64, 50, 30, 5, 50000, 300, 400, 3, 200, 50000, 'igru', 0.2)
    s = _lib.query_sizes(cfg)
    assert s.word_emb == 50000 * 300 and s.user_emb == 50000 * 200
    assert s.user_emb_dim == 200 and s.gru_dim == 200 and s.max_predict_rows == 64 * 55 // 51
    lay = _lib.param_layout(cfg)
    assert [n for n, _, _ in lay] == ['conv_w', 'conv_b', 'att_w', 'att_b', 'dense_w', 'dense_b', 'gru_w', 'gru_u', 'gru_b']
    counts = dict((n, c) for n, _, c in lay)
    assert counts == dict(conv_w=900 * 400, conv_b=400, att_w=400, att_b=1, dense_w=400 * 200, dense_b=200,
                          gru_w=200 * 600, gru_u=200 * 600, gru_b=600)
    assert sum(counts.values()) == 681601                      # 681 601 dense floats at U=200
    offs = [o for _, o, _ in lay]
    assert offs == sorted(offs) and all(o % 64 == 0 for o in offs)
    assert s.dense >= lay[-1][1] + lay[-1][2]
    lay_con = _lib.param_layout(_lib.make_cfg(8, 5, 8, 5, 100, 16, 16, 3, 8, 10, 'gru'))
    assert [n for n, _, _ in lay_con][-2:] == ['con_w', 'con_b']
    h = _lib.query_sizes(_lib.make_cfg(8, 5, 8, 5, 100, 16, 16, 3, 8, 10, 'hgru'))
    assert h.user_emb_dim == 4 and h.gru_dim == 4


def test_bad_configs_are_rejected_with_a_message():
    with pytest.raises(_lib.LsturError, match='multiple of 4'):
        _lib.query_sizes(_lib.make_cfg(4, 3, 5, 2, 10, 6, 8, 3, 8, 4))
    with pytest.raises(_lib.LsturError, match='even'):
        _lib.query_sizes(_lib.make_cfg(4, 3, 5, 2, 10, 8, 8, 3, 7, 4, 'hgru'))
    with pytest.raises(_lib.LsturError, match='not supported'):
        _lib.make_cfg(4, 3, 5, 2, 10, 8, 8, 3, 8, 4, 'satt')
    with pytest.raises(_lib.LsturError, match='2U wide'):          # dgru / ngru are not dot-compatible (paper.py:600-602)
        _lib.query_sizes(_lib.make_cfg(4, 3, 5, 2, 10, 8, 8, 3, 8, 4, 'dgru'))
    _lib.query_sizes(_lib.make_cfg(4, 3, 5, 2, 10, 8, 8, 3, 8, 4, 'dgru', score_model='dnn'))
    with pytest.raises(_lib.LsturError, match='listwise'):
        _lib.query_sizes(_lib.make_cfg(4, 3, 5, 1, 10, 8, 8, 3, 8, 4, 'igru', head='point_dot', score_model='ddot'))
    off, cnt = C.c_size_t(), C.c_size_t()
    cfg = _lib.make_cfg(4, 3, 5, 2, 10, 8, 8, 3, 8, 4)
    assert _lib.lib.lstur_param(C.byref(cfg), b'nope', C.byref(off), C.byref(cnt)) == -1
    assert b'nope' in _lib.lib.lstur_last_error()


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from mnexp_amd.native_model import NativeModel
    import numpy as np
    with pytest.raises(_lib.LsturError, match='no CPU fallback'):
        NativeModel(np.zeros((10, 8), np.float32), 4, 4, 3, 5, 1, 8, 3, 8)
    cfg = _lib.make_cfg(4, 3, 5, 2, 10, 8, 8, 3, 8, 4)
    bufs = _lib.Buffers(*([256] * 9), 1 << 20)
    ctx = C.c_void_p()
    rc = _lib.lib.lstur_create(C.byref(cfg), C.byref(bufs), C.byref(ctx))
    assert rc == -4 and b'no CPU fallback' in _lib.lib.lstur_last_error()


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'mnexp_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in src and 'from oracle' not in src, f


def test_weight_files_are_read_without_executing_them(tmp_path):
    """--enable-pretrain-encoder / NativeModel.load take a path from the user's config: the reader admits ndarrays only
    (reference format: pickle of model.get_weights(), utils.py:74-79) and refuses any other global."""
    import pickle
    import numpy as np
    from mnexp_amd.native_model import load_weight_list
    good = tmp_path / 'w.pkl'
    arrays = [np.arange(6, dtype=np.float32).reshape(2, 3), np.zeros(4, np.float64), np.ones((3, 1), np.float32)]
    with open(good, 'wb') as f:
        pickle.dump(arrays, f, protocol=pickle.HIGHEST_PROTOCOL)
    back = load_weight_list(str(good))
    assert len(back) == 3 and all(np.array_equal(a, b) and a.dtype == b.dtype for a, b in zip(arrays, back))
    with open(good, 'wb') as f:                        # protocol 2: what Python-3.5-era Keras runs wrote
        pickle.dump(arrays, f, protocol=2)
    assert all(np.array_equal(a, b) for a, b in zip(arrays, load_weight_list(str(good))))

    class Evil:
        def __reduce__(self):
            import os
            return (os.system, ('echo pwned > %s' % (tmp_path / 'pwned'),))
    bad = tmp_path / 'evil.pkl'
    with open(bad, 'wb') as f:
        pickle.dump([arrays[0], Evil()], f)
    with pytest.raises(pickle.UnpicklingError):
        load_weight_list(str(bad))
    assert not (tmp_path / 'pwned').exists()
    with open(bad, 'wb') as f:                         # an object array could smuggle arbitrary instances
        pickle.dump([np.array([{'a': 1}], dtype=object)], f)
    with pytest.raises((pickle.UnpicklingError, ValueError)):
        load_weight_list(str(bad))
    with open(bad, 'wb') as f:
        pickle.dump({'not': 'a list'}, f)
    with pytest.raises(ValueError):
        load_weight_list(str(bad))


def test_batches_past_the_32_bit_offset_reach_are_refused():
    """Every kernel addresses a step's token-level tensors through 32-bit buffer offsets; a max_batch whose embeddings or
    conv outputs exceed 4 GiB is refused with a message by lstur_query_sizes (it used to be handed to kernels whose
    descriptors wrap silently)."""
    ok = _lib.make_cfg(512, 50, 30, 5, 100000, 300, 400, 3, 200, 1000000)          # BASELINE config 4 per GPU
    assert _lib.query_sizes(ok).workspace_bytes > 0
    edge = _lib.make_cfg(1300, 50, 30, 5, 1000, 300, 400, 3, 200, 10)
    assert _lib.query_sizes(edge).workspace_bytes > 0
    for B in (1400, 2000, 40000):
        with pytest.raises(_lib.LsturError, match='4 GiB'):
            _lib.query_sizes(_lib.make_cfg(B, 50, 30, 5, 1000, 300, 400, 3, 200, 10))


def test_prefetched_draws_exactly_count_items_in_order_and_reraises():
    """native_model.prefetched: the background generator thread of fit_generator (Keras' GeneratorEnqueuer for workers=1)
    takes exactly `count` items, in order, and hands an exception of the generator to the consumer."""
    from mnexp_amd.native_model import prefetched
    drawn = []

    def gen():
        i = 0
        while True:
            drawn.append(i)
            yield i
            i += 1

    g = gen()
    assert list(prefetched(g, 25, depth=3)) == list(range(25))
    assert drawn == list(range(25))                 # not one draw beyond the count: the RNG streams behind it stay put
    assert next(g) == 25

    def bad():
        yield 1
        raise RuntimeError('feed broke')

    it = prefetched(bad(), 5)
    assert next(it) == 1
    with pytest.raises(RuntimeError, match='feed broke'):
        next(it)


# ---- the ABI guard and the binding a maintainer copies from INTEGRATION.md ------------------------------------------
_CTYPE = {'int32_t': C.c_int32, 'uint32_t': C.c_uint32, 'float': C.c_float, 'size_t': C.c_size_t}


def header_struct(name):
    """[(field, ctypes type)] of `typedef struct <name> { ... } <name>;` in include/lstur.h (pointers -> c_void_p)"""
    text = open(os.path.join(ROOT, 'include', 'lstur.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    body = re.search(r'typedef struct %s \{(.*?)\} %s;' % (name, name), text, flags=re.S).group(1)
    out = []
    for decl in body.split(';'):
        decl = decl.strip()
        if not decl:
            continue
        m = re.match(r'(const\s+)?(\w+)\s*(\*?)\s*(.+)$', decl)
        ctype = C.c_void_p if m.group(3) or '*' in m.group(4) else _CTYPE[m.group(2)]
        for fname in m.group(4).split(','):
            out.append((fname.replace('*', '').strip(), ctype))
    return out


def integration_snippet():
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    a = text.index('# --- BEGIN GENERATED')
    b = text.index('# --- END GENERATED ---')
    ns = {'C': C}
    exec(text[a:b], ns)              # the generated block is plain ctypes declarations
    return ns


def test_integration_snippet_is_the_header_and_the_binding():
    """INTEGRATION.md's binding block is generated from mnexp_amd/_lib.py (tools/gen_integration_snippet.py) and both agree
    with include/lstur.h field by field: a maintainer who copies it passes structs of the size the library reads."""
    import subprocess
    import sys
    assert subprocess.call([sys.executable, os.path.join(ROOT, 'tools', 'gen_integration_snippet.py'), '--check']) == 0
    ns = integration_snippet()
    for cname, hname, ours in (('Cfg', 'lstur_cfg', _lib.Cfg), ('Sizes', 'lstur_sizes', _lib.Sizes),
                               ('Buffers', 'lstur_buffers', _lib.Buffers)):
        want = header_struct(hname)
        assert [(n, t) for n, t in ns[cname]._fields_] == want, cname
        assert [(n, t) for n, t in ours._fields_] == want, cname
        assert C.sizeof(ns[cname]) == C.sizeof(ours)
    assert ns['ARCH'] == _lib.ARCH and ns['SCORE'] == _lib.SCORE
    text = open(os.path.join(ROOT, 'include', 'lstur.h')).read()
    hdr_abi = int(re.search(r'#define LSTUR_ABI_VERSION (\d+)', text).group(1))
    assert ns['ABI_VERSION'] == _lib.ABI_VERSION == hdr_abi == _lib.lib.lstur_abi_version()
    for i, nm in enumerate(re.findall(r'LSTUR_ARCH_(\w+) = (\d+)', text)):
        assert _lib.ARCH[nm[0].lower()] == int(nm[1])


def test_a_cfg_of_another_size_is_refused_not_read_past():
    """the round-2 binding (no struct_size, no id_keep: 96 bytes) and a cfg whose struct_size lies are refused by every
    entry point that takes a lstur_cfg"""
    cfg = _lib.make_cfg(4, 3, 5, 2, 10, 8, 8, 3, 8, 4)
    assert cfg.struct_size == C.sizeof(_lib.Cfg)
    for bad in (0, C.sizeof(_lib.Cfg) - 4, C.sizeof(_lib.Cfg) + 4):
        cfg.struct_size = bad
        s = _lib.Sizes()
        assert _lib.lib.lstur_query_sizes(C.byref(cfg), C.byref(s)) == -1
        assert b'struct_size' in _lib.lib.lstur_last_error()
        off, cnt = C.c_size_t(), C.c_size_t()
        assert _lib.lib.lstur_param(C.byref(cfg), b'conv_w', C.byref(off), C.byref(cnt)) == -1
        assert _lib.lib.lstur_param_name(C.byref(cfg), 0) is None
        assert _lib.lib.lstur_dp_tail_floats(C.byref(cfg), 2) == 0
        bufs = _lib.Buffers(*([256] * 9), 1 << 20)
        ctx = C.c_void_p()
        assert _lib.lib.lstur_create(C.byref(cfg), C.byref(bufs), C.byref(ctx)) == -1


def test_conv_width_is_bounded_by_the_reduction_launch():
    """the backward pass hands one reduction job per conv tap to ONE launch of at most 32 jobs; a width that does
    not fit is refused when the configuration is made, not truncated at run time"""
    _lib.query_sizes(_lib.make_cfg(4, 3, 5, 2, 10, 8, 8, 20, 8, 4))
    with pytest.raises(_lib.LsturError, match='kernel'):
        _lib.query_sizes(_lib.make_cfg(4, 3, 5, 2, 10, 8, 8, 21, 8, 4))


def test_product_library_carries_no_timing_only_kernels():
    """the ablation kernels (WRONG results by design) and the measurement knobs are compiled into the bench-only library
    (make ablate, -DLSTUR_ABLATE) alone"""
    blob = open(_lib.LIB_PATH, 'rb').read()
    for needle in (b'ABLATE', b'LSTUR_GRU4_DEBUG', b'LSTUR_GRU_DEBUG', b'LSTUR_CONV_GEN1', b'LSTUR_CONV_FWD3', b'LSTUR_PIPELINE',
                   b'LSTUR_LATE_SLOTS', b'LSTUR_SLOT_MAJOR', b'LSTUR_SPLIT_BG', b'LSTUR_ONE_SIDE', b'LSTUR_WG_STREAMS'):
        assert needle not in blob, needle
