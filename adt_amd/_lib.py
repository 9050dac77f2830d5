"""ctypes binding of libadt_hip.so (C ABI: include/adt_hip.h).

The HIP library IS the product path: if it is missing or fails to load this module raises -- there is no
CPU or PyTorch fallback anywhere in adt_amd/ (the numpy oracle under oracle/ is test infrastructure only).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libadt_hip.so")

_P, _I, _F, _U, _L = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_uint32, ctypes.c_int64


class SasrecCfg(ctypes.Structure):
    """struct adt_sasrec_cfg (include/adt_hip.h)."""
    _fields_ = [("item_num", ctypes.c_int32), ("maxlen", ctypes.c_int32), ("hidden", ctypes.c_int32),
                ("num_heads", ctypes.c_int32), ("num_layers", ctypes.c_int32), ("dropout", ctypes.c_float),
                ("prec", ctypes.c_int32)]


_CP = ctypes.POINTER(SasrecCfg)


class EncLayerPtrs(ctypes.Structure):
    """struct adt_enc_layer_ptrs: device pointers of one encoder layer's tensors (or of their gradient accumulators)."""
    NAMES = ("ln1_w", "ln1_b", "in_w", "in_b", "out_w", "out_b", "ln2_w", "ln2_b", "c1_w", "c1_b", "c2_w", "c2_b", "cls_w", "cls_b")
    _fields_ = [(n, ctypes.c_void_p) for n in NAMES]


class DecLayerPtrs(ctypes.Structure):
    """struct adt_dec_layer_ptrs."""
    NAMES = ("ln_w", "ln_b", "sin_w", "sin_b", "so_w", "so_b", "ein_w", "ein_b", "eo_w", "eo_b", "c1_w", "c1_b", "c2_w", "c2_b")
    _fields_ = [(n, ctypes.c_void_p) for n in NAMES]


_EP, _DP = ctypes.POINTER(EncLayerPtrs), ctypes.POINTER(DecLayerPtrs)

# name -> (restype, argtypes); the single source of truth for symbol coverage (tests/test_capi_symbols.py)
SIGNATURES = {
    "adt_version": (_I, []),
    "adt_last_error": (ctypes.c_char_p, []),
    "adt_rng_keep": (_I, [_U, _U, _U, _F]),
    "adt_embed_fwd": (_I, [_P, _P, _P, _I, _I, _I, _F, _P, _U, _U, _P, _P]),
    "adt_embed_bwd": (_I, [_P, _P, _I, _I, _I, _F, _P, _U, _U, _P, _P, _P]),
    "adt_layernorm_fwd": (_I, [_P, _I, _P, _P, _F, _I, _I, _P, _I, _P]),
    "adt_layernorm_bwd": (_I, [_P, _I, _P, _I, _P, _F, _I, _I, _P, _I, _I, _P, _P, _P]),
    "adt_linear_fwd": (_I, [_I, _P, _I, _P, _P, _I, _I, _I, _P, _I, _F, _P, _U, _U, _I, _P, _I, _P, _I, _P, _P]),
    "adt_linear_bwd": (_I, [_I, _P, _I, _P, _I, _P, _I, _I, _I, _P, _F, _P, _U, _U, _P, _I, _P, _I, _I, _P, _I, _P,
                            _P, _P, _P]),
    "adt_attn_fwd": (_I, [_I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _F, _P, _U, _U, _P, _I, _P, _P, _P]),
    "adt_attn_bwd": (_I, [_I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P, _U, _U, _P, _I,
                          _P, _I, _P, _I, _P, _P]),
    "adt_headcls_fwd": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _P, _P]),
    "adt_headcls_bwd": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _P, _I, _P, _P, _P]),
    "adt_logits_fwd": (_I, [_P, _I, _P, _P, _P, _I, _I, _P, _P, _P]),
    "adt_logits_bwd": (_I, [_P, _I, _P, _P, _P, _P, _P, _I, _I, _P, _I, _P, _P]),
    "adt_item_scatter": (_I, [_P, _P, _I, _P, _I, _I, _F, _F, _P, _U, _U, _P, _I, _L, _P]),
    "adt_embed_bwd3": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _P, _U, _U, _U, _P, _P, _I, _L, _P]),
    "adt_replica_reduce": (_I, [_P, _P, _L, _I, _L, _P]),
    "adt_posemb_bwd": (_I, [_P, _P, _I, _I, _I, _F, _P, _U, _U, _P, _P]),
    "adt_logits_bwd_df": (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _I, _P]),
    "adt_bce_seed": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _P]),
    "adt_mse_seed": (_I, [_P, _P, _L, _F, _P, _P, _I, _P, _P, _P]),
    "adt_nll_seed": (_I, [_P, _I, _I, _F, _P, _P, _P, _P]),
    "adt_clip_adam": (_I, [_P, _P, _P, _P, _L, _L, _F, _F, _F, _F, _F, _F, _F, _P, _P]),
    "adt_clip_adam_pre": (_I, [_P, _P, _P, _P, _L, _L, _F, _F, _F, _F, _F, _F, _F, _P, _P]),
    "adt_score_rank": (_I, [_P, _I, _P, _P, _I, _I, _I, _P, _P, _P]),
    # ---- general ("wide") stage kernels
    "adt_dense_fwd": (_I, [_I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _I, _F, _P, _U, _U, _P, _I, _P, _I, _P, _P, _I, _P, _P]),
    "adt_axpy": (_I, [_P, _P, _F, _I, _L, _P, _I, _P]),
    "adt_grad_sumsq": (_I, [_P, _L, _P, _P]),
    "adt_adam_range": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _F, _F, _P, _P]),
    "adt_adamw_range": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _F, _F, _P, _P]),
    "adt_log_softmax_fwd": (_I, [_P, _L, _I, _P, _P]),
    "adt_log_softmax_bwd": (_I, [_P, _P, _L, _I, _P, _I, _P]),
    "adt_dense_bwd": (_I, [_I, _P, _I, _I, _I, _I, _P, _F, _P, _U, _U, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _P, _I, _P, _P, _P]),
    "adt_attn_masked_fwd": (_I, [_I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P, _F, _F, _P, _U, _U, _P, _I, _P, _P]),
    "adt_attn_masked_bwd": (_I, [_I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P, _F, _F, _P, _U, _U,
                                 _P, _I, _P, _I, _P, _I, _P]),
    "adt_embed_sum_fwd": (_I, [_P, _P, _P, _P, _F, _I, _I, _I, _P, _P]),
    "adt_dropact_fwd": (_I, [_P, _L, _F, _P, _U, _U, _I, _P, _P]),
    "adt_dropact_bwd": (_I, [_P, _P, _L, _F, _P, _U, _U, _I, _P, _I, _P]),
    "adt_gather_rows": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _P]),
    "adt_scatter_rows": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _P]),
    "adt_ce_rows": (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _P]),
    "adt_lce_supported": (_I, [_I, _I]),
    "adt_lce_slots": (_I, [_I]),
    "adt_lce_workspace_bytes": (_L, [_I, _I, _I]),
    "adt_lce_fwd_bwd": (_I, [_P, _I, _P, _P, _I, _P, _P, _I, _P, _I, _I, _P, _P, _P, _P, _I, _P, _I, _P, _P, _L, _P]),
    "adt_clip_adam_l2": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _F, _F, _P, _P]),
    "adt_score_rank_bias": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _P, _P, _P]),
    "adt_wattn_mfma_fwd": (_I, [_I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _F, _P, _U, _U, _P, _I, _P, _I, _P, _P]),
    "adt_wattn_mfma_bwd": (_I, [_I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _I, _P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _I, _F,
                                _P, _U, _U, _P, _P, _P, _P, _P, _P, _I, _P]),
    "adt_wattn_fwd": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _F, _P, _U, _U, _P, _I, _P, _I, _P, _P]),
    "adt_wattn_bwd": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _I, _P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _I, _F,
                           _P, _U, _U, _P, _P, _P, _P, _P, _P, _I, _P]),
    "adt_wdist_bpr": (_I, [_P, _P, _I, _P, _P, _P, _P, _I, _I, _F, _P, _P, _P, _I, _P, _P, _P, _P]),
    "adt_wdist_full": (_I, [_P, _P, _I, _P, _P, _I, _I, _I, _P, _I, _P]),
    "adt_dense_rows_enable": (_I, [_I]),
    "adt_dense_workspace": (_I, [_P, _L]),
    "adt_dense_gradsrc": (_I, [_P, _I, _I, _I, _P, _F, _P, _U, _U, _I, _P, _I, _P, _I, _P, _P]),
    "adt_topk_masked": (_I, [_P, _I, _I, _I, _P, _P, _I, _P, _P, _P]),
    "adt_item_sort_supported": (_I, [_I]),
    "adt_item_sort_work_ints": (_L, [_I, _I, _I]),
    "adt_item_sort": (_I, [_P, _I, _I, _I, _P, _P, _P, _U, _P, _P]),
    "adt_item_segsum": (_I, [_P, _I, _I, _I, _U, _P, _F, _P, _F, _P, _I, _P]),
    "adt_posemb_sum": (_I, [_P, _P, _P, _I, _I, _I, _F, _P, _U, _P, _P]),
    "adt_item_segsum_posemb": (_I, [_P, _I, _I, _I, _U, _P, _F, _P, _F, _P, _I, _P, _P, _P, _I, _I, _I, _U, _P, _P]),
    "adt_sasrec_param_layout": (_L, [_CP, _P]),
    "adt_sasrec_workspace_floats": (_L, [_CP, _I]),
    "adt_sasrec_ws_offset": (_L, [_CP, _I, _I, _I]),
    "adt_sasrec_forward": (_I, [_CP, _P, _P, _P, _P, _P, _P, _I, _I, _P, _U, _P]),
    "adt_sasrec_probe_dec_layer_fwd": (_I, [_CP, _P, _P, _P, _I, _I, _P, _U, _I, _P]),
    "adt_seq_layer_supported": (_I, [_I, _I, _I, _I]),
    "adt_seq_layer_save_floats": (_L, [_I, _I, _I, _I]),
    "adt_pack_wimg": (_I, [_P, _P, _P, _I, _P]),
    "adt_seq_enc_layer_fwd": (_I, [_I, _I, _I, _P, _P, _EP, _P, _P, _F, _P, _U, _U, _U, _U, _I, _P, _P, _F, _I, _P, _P]),
    "adt_seq_enc_layer_bwd": (_I, [_I, _I, _I, _P, _P, _EP, _EP, _P, _P, _F, _P, _U, _U, _U, _U, _P, _P, _F, _P, _P, _P, _I, _P, _P]),
    "adt_seq_dec_layer_fwd": (_I, [_I, _I, _I, _P, _P, _P, _DP, _P, _P, _F, _P, _U, _U, _U, _U, _U, _P, _P, _F, _I, _P]),
    "adt_seq_dec_layer_bwd": (_I, [_I, _I, _I, _P, _P, _P, _DP, _DP, _P, _P, _F, _P, _U, _U, _U, _U, _U, _P, _P, _F, _P, _I, _P, _P, _P]),
    "adt_sasrec_loss_seed": (_I, [_CP, _P, _P, _I, _P, _P, _P]),
    "adt_sasrec_loss_seed_nz": (_I, [_CP, _P, _P, _I, _P, _P, _P]),
    "adt_sasrec_forward_loss": (_I, [_CP, _P, _P, _P, _P, _P, _P, _I, _I, _P, _U, _P, _P, _P]),
    "adt_sasrec_bce_deferred": (_I, [_CP]),
    "adt_sasrec_forward_loss_prefetch": (_I, [_CP, _P, _P, _P, _P, _P, _P, _I, _I, _P, _U, _P, _P, _P, _L, _I, _P, _P, _P, _P]),
    "adt_sasrec_step_begin_ring_staged": (_I, [_CP, _P, _I, _P, _U, _P, _L, _I, _P, _P, _P, _P, _P, _P, _P, _L, _P, _P]),
    "adt_sasrec_fold_clip_adam": (_I, [_CP, _P, _I, _P, _P, _P, _P, _F, _F, _F, _F, _F, _F, _P, _P]),
    "adt_sasrec_fold_grads": (_I, [_CP, _P, _I, _P, _P, _P, _P]),
    "adt_sasrec_step_begin": (_I, [_CP, _P, _I, _P, _U, _P, _P, _P, _L, _P, _P]),
    "adt_sasrec_step_begin_ring": (_I, [_CP, _P, _I, _P, _U, _P, _L, _I, _P, _P, _P, _P, _P, _L, _P, _P]),
    "adt_sasrec_backward": (_I, [_CP, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _U, _I, _P]),
    "adt_sasrec_predict": (_I, [_CP, _P, _P, _P, _P, _I, _I, _P, _P, _P]),
}

def dropout_rate(p, what="dropout"):
    """The rate a kernel actually applies for a requested dropout probability: drop decisions compare one random BYTE with the threshold
    round(256 p), so the rate is round(256 p) / 256 (0.5 -> 0.5, 0.2 -> 0.19922, 0.3 -> 0.30078; survivors are scaled by the inverse of the
    quantised keep rate, so the mask stays unbiased) and is capped at 255 / 256.  A p > 0 that would quantise to ZERO (p < 1/512) is an
    error instead of a silent `dropout off`.  Returns float(p) unchanged: the quantisation happens in adt_make_drop / oracle.rng.threshold."""
    p = float(p)
    if not 0.0 <= p < 1.0:
        raise ValueError("%s = %r: need 0 <= p < 1" % (what, p))
    if p > 0.0 and int(p * 256.0 + 0.5) == 0:
        raise ValueError("%s = %g is below the 8-bit resolution of the dropout RNG (1/512): it would silently disable dropout; use 0 or >= 0.002" % (what, p))
    return p


_lib = None


class AdtError(RuntimeError):
    pass


def load():
    """Load libadt_hip.so once.  Raises AdtError (never falls back) when it is absent or unloadable."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: libadt_hip.so's NEEDED libamdhip64.so.7 must bind to the HIP runtime torch has already
    # loaded (its bundled copy), otherwise the process ends up with two runtimes and ours sees no device
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise AdtError("libadt_hip.so not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise AdtError("cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        raise AdtError("%s failed: %s" % (what, load().adt_last_error().decode()))
