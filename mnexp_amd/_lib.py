"""ctypes binding of liblstur_hip.so (C-ABI declared in include/lstur.h).

There is NO CPU fallback: if the shared library is missing the import of this module
raises, and ``lstur_create`` fails with LSTUR_ERR_NOGPU when no gfx950 device is visible.
Build the library with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C mnexp_amd/csrc``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (LSTUR_LIB: another build of the same library, for same-box A/B runs of two kernel versions)
LIB_PATH = os.environ.get('LSTUR_LIB') or os.path.join(_HERE, 'liblstur_hip.so')

ARCH = {'igru': 0, 'gru': 1, 'hgru': 2, 'nigru': 3, 'ngru': 4, 'dgru': 5, 'pgru': 6, 'vo': 7, 'niavg': 8, 'niatt': 9, 'iigru': 10, 'iicgru': 11}
SCORE = {'dot': 0, 'ddot': 1, 'dnn': 2}     # --score-model (paper.py:443-458)
HEAD = {'softmax': 0, 'point_mlp': 1, 'point_dot': 2}
ENCODER = {'cnnatt': 0, 'mhsa': 1}     # include/lstur.h: lstur_encoder (mhsa is build-defined, no reference counterpart)


class LsturError(RuntimeError):
    pass


ABI_VERSION = 3     # include/lstur.h: LSTUR_ABI_VERSION this binding was written against


class Cfg(C.Structure):
    _fields_ = [('struct_size', C.c_uint32)] + [(n, C.c_int32) for n in ('max_batch', 'window', 'title_len', 'cands', 'vocab', 'word_dim',
                                         'filters', 'kernel', 'user_dim', 'n_users', 'arch')] + [
        ('dropout', C.c_float), ('seed', C.c_uint32), ('beta1', C.c_float), ('beta2', C.c_float), ('eps', C.c_float),
        ('head', C.c_int32), ('gain', C.c_float), ('loss_negatives', C.c_int32), ('word_trainable', C.c_int32),
        ('freeze_encoder', C.c_int32), ('news_encoder', C.c_int32), ('heads', C.c_int32), ('score_model', C.c_int32),
        ('id_keep', C.c_float)]


class Sizes(C.Structure):
    _fields_ = [('word_emb', C.c_size_t), ('dense', C.c_size_t), ('user_emb', C.c_size_t),
                ('workspace_bytes', C.c_size_t), ('user_emb_dim', C.c_int32), ('gru_dim', C.c_int32),
                ('max_predict_rows', C.c_int32)]


class Buffers(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('word_emb', 'dense_param', 'dense_grad', 'dense_m', 'dense_v',
                                          'user_emb', 'user_m', 'user_v', 'workspace')] + [
        ('workspace_bytes', C.c_size_t), ('word_grad', C.c_void_p), ('word_m', C.c_void_p), ('word_v', C.c_void_p),
        ('user_emb2', C.c_void_p), ('user_m2', C.c_void_p), ('user_v2', C.c_void_p)]


_P, _I, _U, _F, _Z = C.c_void_p, C.c_int32, C.c_uint32, C.c_float, C.c_size_t

# name -> (restype, argtypes); every symbol include/lstur.h declares
SIGNATURES = {
    'lstur_last_error': (C.c_char_p, []),
    'lstur_version': (C.c_char_p, []),
    'lstur_abi_version': (_I, []),
    'lstur_query_sizes': (_I, [C.POINTER(Cfg), C.POINTER(Sizes)]),
    'lstur_param': (_I, [C.POINTER(Cfg), C.c_char_p, C.POINTER(_Z), C.POINTER(_Z)]),
    'lstur_param_name': (C.c_char_p, [C.POINTER(Cfg), _I]),
    'lstur_create': (_I, [C.POINTER(Cfg), C.POINTER(Buffers), C.POINTER(_P)]),
    'lstur_destroy': (_I, [_P]),
    'lstur_set_title_table': (_I, [_P, _P, _I]),
    'lstur_forward_backward': (_I, [_P, _P, _P, _P, _I, _U, _F, _P, _P]),
    'lstur_forward_backward_ids': (_I, [_P, _P, _P, _P, _I, _U, _F, _P, _P]),
    'lstur_prefetch_ids': (_I, [_P, _P, _P, _I, _U]),
    'lstur_prefetch_hits': (_I, [_P]),
    'lstur_prefetch_rows': (_I, [_P, _P]),
    'lstur_user_grad': (_I, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_I)]),
    'lstur_adam_step': (_I, [_P, _F, _I, _P, _P, _I, _P]),
    'lstur_params_changed': (_I, [_P]),
    'lstur_flush_tables': (_I, [_P, _P]),
    'lstur_lazy_adam': (_I, [_P]),
    'lstur_train_step': (_I, [_P, _P, _P, _P, _P, _I, _U, _F, _I, _P, _P]),
    'lstur_train_step_ids': (_I, [_P, _P, _P, _P, _P, _I, _U, _F, _I, _P, _P]),
    'lstur_dp_tail_floats': (_Z, [C.POINTER(Cfg), _I]),
    'lstur_dp_stage': (_I, [_P, _P, _I, _I, _P]),
    'lstur_dp_unpack': (_I, [_P, _P, _I, _I, C.POINTER(_P), C.POINTER(_P), _P]),
    'lstur_dp_attach': (_I, [_P, _P, _I, _I]),
    'lstur_adam_step_dp': (_I, [_P, _F, _I, _P, _I, _I, _P]),
    'lstur_predict': (_I, [_P, _P, _P, _P, _I, _P, _P]),
    'lstur_predict_ids': (_I, [_P, _P, _P, _P, _I, _P, _P]),
    'lstur_evaluate': (_I, [_P, _P, _P, _P, _I, _P, _P]),
    'lstur_point_forward_backward': (_I, [_P, _P, _P, _P, _P, _I, _U, _F, _P, _P]),
    'lstur_point_forward_backward_ids': (_I, [_P, _P, _P, _P, _P, _I, _U, _F, _P, _P]),
    'lstur_point_evaluate': (_I, [_P, _P, _P, _P, _P, _I, _P, _P]),
    'lstur_profile': (_I, [_P, _I]),
    'lstur_profile_read': (_I, [_P, _I, C.POINTER(C.c_char_p), C.POINTER(_F), C.POINTER(_I)]),
    'lstur_conv_scheme': (_I, [_P]),
    'lstur_peek': (_I, [_P, C.c_char_p, C.POINTER(_P), C.POINTER(_Z)]),
    'lstur_gather_f32': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _F, _U, _P]),
    'lstur_valid_tokens_scratch': (_Z, [_I]),
    'lstur_valid_tokens': (_I, [_P, _I, _P, _P, _P, _P]),
    'lstur_conv1d_relu_mask_fwd': (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    'lstur_convx_scales_bytes': (_Z, []),
    'lstur_convx_scales_init': (_I, [_P, _P]),
    'lstur_convx_embedding_scale': (_I, [_P, _Z, _F, _P, _P]),
    'lstur_convx_e_elems': (_Z, [_I, _I, _I, _I, _I]),
    'lstur_convx_gather': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _F, _U, _I, _I, _P, _P, _P]),
    'lstur_convx_w_elems': (_Z, [_I, _I, _I, _I]),
    'lstur_convx_pack_w': (_I, [_P, _I, _I, _I, _P, _I, _P, _P]),
    'lstur_convx_fwd_scratch': (_Z, []),
    'lstur_convx_fwd': (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _P, _P, _P]),
    'lstur_convx_wgrad_scratch': (_Z, [_I, _I, _I, _I, _I]),
    'lstur_convx_wgrad': (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _P, _P]),
    'lstur_conv1d_wgrad_scratch': (_Z, [_I, _I, _I]),
    'lstur_conv1d_wgrad': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    'lstur_pack_b_floats': (_Z, [_I, _I]),
    'lstur_pack_b': (_I, [_P, _I, _I, _I, _P, _P]),
    'lstur_pack_bt': (_I, [_P, _I, _I, _I, _P, _P]),
    'lstur_attpool_fwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _U, _P]),
    'lstur_attpool_bwd_parts': (_I, [_I]),
    'lstur_attpool_bwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _U, _P]),
    'lstur_mhsa_fwd': (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    'lstur_mhsa_bwd': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    'lstur_proj_fwd_bf16': (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    'lstur_proj_wgrad_scratch': (_Z, [_I, _I, _I]),
    'lstur_proj_wgrad_bf16': (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    'lstur_colsum': (_I, [_P, _I, _I, _I, _P, _F, _P]),
    'lstur_gemm_scratch': (_Z, [_I, _I, _I]),
    'lstur_gemm_bf16x3': (_I, [_P, _I, _P, _I, _I, _P, _I, _P, _I, _I, _I, _P]),
    'lstur_gemm_bf16x3_tn_scratch': (_Z, [_I, _I, _I]),
    'lstur_gemm_bf16x3_tn': (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P]),
    'lstur_gemm_f32': (_I, [_P, _I, _I, _P, _I, _I, _P, _I, _P, _I, _I, _I, _F, _P, _P]),
    'lstur_gru_pack_floats': (_Z, [_I]),
    'lstur_gru_pack_u': (_I, [_P, _I, _P, _P]),
    'lstur_gru_tape_floats': (_Z, [_I, _I, _I]),
    'lstur_gru_fwd': (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    'lstur_gru_bwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    'lstur_score_softmax_ce': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P]),
    'lstur_adam_flat': (_I, [_P, _P, _P, _P, _Z, _F, _F, _F, _F, _P]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise LsturError(
            'liblstur_hip.so is not built ({}): the LSTUR path is HIP-only and has no CPU fallback. '
            'Run `make -C mnexp_amd/csrc` (hipcc, gfx950).'.format(LIB_PATH))
    # PyTorch-ROCm bundles its own HIP runtime; if this library is loaded first it binds the system one and the two
    # runtimes then disagree about the devices (hipGetDeviceCount fails in whichever came second).  Loading torch
    # first makes both share torch's copy.  (The plain-C example links the system runtime and needs no torch.)
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.lstur_abi_version() != ABI_VERSION:
        raise LsturError('{} was built from ABI version {}, this binding is written against {}: rebuild the library '
                         '(make -C mnexp_amd/csrc)'.format(LIB_PATH, lib.lstur_abi_version(), ABI_VERSION))
    return lib


lib = _load()


def check(rc, what=''):
    if rc != 0:
        raise LsturError('{} failed ({}): {}'.format(what or 'lstur call', rc, lib.lstur_last_error().decode()))


def make_cfg(max_batch, window, title_len, cands, vocab, word_dim, filters, kernel, user_dim, n_users,
             arch='igru', dropout=0.0, seed=0, beta1=0.9, beta2=0.999, eps=1e-7, head='softmax', gain=1.0,
             loss_negatives=4, freeze_encoder=False, word_trainable=False, news_encoder='cnnatt', heads=0,
             score_model='dot', id_keep=1.0):
    if arch not in ARCH:
        raise LsturError('arch {!r} is not supported by the native path (have {})'.format(arch, sorted(ARCH)))
    if head not in HEAD:
        raise LsturError('head {!r} is not supported by the native path (have {})'.format(head, sorted(HEAD)))
    if score_model not in SCORE:
        raise LsturError('score model {!r} is not supported by the native path (have {})'.format(score_model, sorted(SCORE)))
    if news_encoder not in ENCODER:
        raise LsturError('news encoder {!r} is not supported by the native path (have {})'.format(
            news_encoder, sorted(ENCODER)))
    return Cfg(C.sizeof(Cfg), max_batch, window, title_len, cands, vocab, word_dim, filters, kernel, user_dim, n_users,
               ARCH[arch], dropout, seed, beta1, beta2, eps, HEAD[head], gain, loss_negatives,
               int(bool(word_trainable)), int(bool(freeze_encoder)), ENCODER[news_encoder], int(heads),
               SCORE[score_model], float(id_keep))


def query_sizes(cfg):
    s = Sizes()
    check(lib.lstur_query_sizes(C.byref(cfg), C.byref(s)), 'lstur_query_sizes')
    return s


def param_layout(cfg):
    """[(name, offset, count)] in bucket order."""
    out = []
    i = 0
    while True:
        nm = lib.lstur_param_name(C.byref(cfg), i)
        if nm is None:
            break
        off, cnt = _Z(), _Z()
        check(lib.lstur_param(C.byref(cfg), nm, C.byref(off), C.byref(cnt)), 'lstur_param')
        out.append((nm.decode(), off.value, cnt.value))
        i += 1
    return out
