"""TEST INFRASTRUCTURE ONLY -- ctypes wrapper of oracle/libadt_cpu.so (oracle/csrc/adt_cpu.cpp): the C++ / OpenMP fp32 restatement of the
SASRec-ADT training + ranking step, with the model-level signatures of include/adt_hip.h (`adt_cpu_sasrec_forward / loss_seed / backward`,
`adt_cpu_clip_adam`, `adt_cpu_sasrec_predict`).  Only tests/, __graft_entry__ and bench.py's cpu_baseline leg import this module.

Parity status: PINNED (tests/test_cpu_restatement.py: the goldens recorded from the reference, and the numpy oracle with dropout on)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "adt_cpu.cpp")
LIB = os.path.join(HERE, "libadt_cpu.so")
# x86-64-v3 (AVX2 + FMA), not -march=native: the library is built in the build container and travels to the GPU box's host
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-march=x86-64-v3", "-fno-math-errno"]

WS_ENC_X, WS_DEC_X, WS_REC, WS_POS_LOGITS, WS_NEG_LOGITS, WS_F, WS_G_ENC_X, WS_G_DEC_X, WS_G_REC, WS_G_POS, WS_G_NEG, WS_LOSS, WS_NORMS = range(13)


class Cfg(ctypes.Structure):
    _fields_ = [("item_num", ctypes.c_int32), ("maxlen", ctypes.c_int32), ("hidden", ctypes.c_int32), ("num_heads", ctypes.c_int32),
                ("num_layers", ctypes.c_int32), ("dropout", ctypes.c_float), ("prec", ctypes.c_int32)]


def build(force=False, out=LIB, extra=(), verbose=False):
    if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(SRC):
        return out
    cmd = [os.environ.get("CXX", "g++")] + FLAGS + list(extra) + ["-o", out, SRC]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


_lib = None


def load():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(build())
        P, I, L, F, U = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_uint32
        CP = ctypes.POINTER(Cfg)
        sig = {"adt_cpu_version": (I, []), "adt_cpu_set_threads": (None, [I]), "adt_cpu_threads": (I, []),
               "adt_cpu_sasrec_param_layout": (L, [CP, P]), "adt_cpu_sasrec_workspace_floats": (L, [CP, I]),
               "adt_cpu_sasrec_ws_offset": (L, [CP, I, I, I]),
               "adt_cpu_sasrec_forward": (I, [CP, P, P, P, P, P, P, I, I, P, U, P]),
               "adt_cpu_sasrec_loss_seed": (I, [CP, P, P, I, P, P, P]),
               "adt_cpu_sasrec_backward": (I, [CP, P, P, P, P, P, P, P, I, I, P, U, I, P]),
               "adt_cpu_clip_adam": (I, [P, P, P, P, L, L, F, F, F, F, F, F, F, P, P]),
               "adt_cpu_sasrec_predict": (I, [CP, P, P, P, P, I, I, P, P])}
        for name, (res, args) in sig.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data


class CpuSasrec:
    """The restatement behind the oracle's calling conventions: parameters as a dict of numpy arrays keyed by the reference's state_dict names."""

    def __init__(self, item_num, maxlen, hidden, num_heads, num_layers, dropout=0.0, threads=0):
        from . import sasrec_oracle as so
        self.lib = load()
        if threads:
            self.lib.adt_cpu_set_threads(threads)
        self.cfg = Cfg(item_num, maxlen, hidden, num_heads, num_layers, dropout, 0)
        self.ocfg = so.Cfg(item_num, maxlen, hidden, num_heads, num_layers, dropout)
        n = 4 + 30 * num_layers
        offs = (ctypes.c_int64 * n)()
        self.n = int(self.lib.adt_cpu_sasrec_param_layout(ctypes.byref(self.cfg), offs))
        # slot order of adt_sasrec_param_layout -> state_dict names (adt_amd/sasrec/model.py:param_table, restated here: no product import)
        enc = ["attention_layernorm.weight", "attention_layernorm.bias", "attention_layer.in_proj_weight", "attention_layer.in_proj_bias",
               "attention_layer.out_proj.weight", "attention_layer.out_proj.bias", "forward_layernorm.weight", "forward_layernorm.bias",
               "forward_layer.conv1.weight", "forward_layer.conv1.bias", "forward_layer.conv2.weight", "forward_layer.conv2.bias", "sparse.weight", "sparse.bias"]
        dec = ["layer_norm.weight", "layer_norm.bias", "slf_attn.in_proj_weight", "slf_attn.in_proj_bias", "slf_attn.out_proj.weight", "slf_attn.out_proj.bias",
               "enc_attn.in_proj_weight", "enc_attn.in_proj_bias", "enc_attn.out_proj.weight", "enc_attn.out_proj.bias", "pos_ffn.conv1.weight",
               "pos_ffn.conv1.bias", "pos_ffn.conv2.weight", "pos_ffn.conv2.bias", "pos_ffn_layernorm.weight", "pos_ffn_layernorm.bias"]
        names = ["item_emb.weight", "pos_emb.weight", "last_layernorm.weight", "last_layernorm.bias"]
        for i in range(num_layers):
            names += ["encoder.encoder_layers.%d.%s" % (i, s) for s in enc]
        for i in range(num_layers):
            names += ["decoder.decoder_layers.%d.%s" % (i, s) for s in dec]
        shapes = dict(so.param_shapes(self.ocfg))
        self.views = {nm: (int(o), shapes[nm]) for nm, o in zip(names, offs)}
        self.P = np.zeros(self.n, np.float32)
        self.G = np.zeros(self.n, np.float32)
        self.M = np.zeros(self.n, np.float32)
        self.V = np.zeros(self.n, np.float32)
        self.scal = np.zeros(8, np.float32)
        self._ws = {}
        self.seed = np.zeros(1, np.uint32)

    def view(self, flat, name):
        o, shp = self.views[name]
        return flat[o:o + int(np.prod(shp))].reshape(shp)

    def load_params(self, P):
        for k, v in P.items():
            self.view(self.P, k)[...] = v

    def params(self):
        return {k: self.view(self.P, k).copy() for k in self.views}

    def grads(self):
        return {k: self.view(self.G, k).copy() for k in self.views}

    def ws(self, B):
        if B not in self._ws:
            self._ws[B] = np.zeros(int(self.lib.adt_cpu_sasrec_workspace_floats(ctypes.byref(self.cfg), B)), np.float32)
        return self._ws[B]

    def out(self, B, what, layer, shape):
        o = int(self.lib.adt_cpu_sasrec_ws_offset(ctypes.byref(self.cfg), B, what, layer))
        return self.ws(B)[o:o + int(np.prod(shape))].reshape(shape)

    @staticmethod
    def _ids(a):
        return np.ascontiguousarray(a, dtype=np.int32)

    def forward(self, seq, dec, pos, neg, training=False, seed=0, b_offset=0):
        seq, dec, pos, neg = (self._ids(a) for a in (seq, dec, pos, neg))
        B = seq.shape[0]
        self.seed[0] = seed
        rc = self.lib.adt_cpu_sasrec_forward(ctypes.byref(self.cfg), _p(self.P), _p(self.ws(B)), _p(seq), _p(dec), _p(pos), _p(neg), B, int(training),
                                             _p(self.seed), b_offset, None)
        assert rc == 0
        self._batch = (seq, dec, pos, neg, B, int(training), b_offset)

    def loss_seed(self, lambdas1, lambdas2, norms=None):
        seq, dec, pos, neg, B, _, _ = self._batch
        c = self.cfg
        if norms is None:
            norms = (float(np.count_nonzero(pos)), float(B * c.maxlen * c.hidden), float(B * c.maxlen * c.num_heads))
        self.out(B, WS_NORMS, 0, (4,))[:3] = norms
        l1, l2 = np.asarray(lambdas1, np.float32), np.asarray(lambdas2, np.float32)
        assert self.lib.adt_cpu_sasrec_loss_seed(ctypes.byref(c), _p(self.ws(B)), _p(pos), B, _p(l1), _p(l2), None) == 0
        nl = c.num_layers
        slots = self.out(B, WS_LOSS, 0, (2 + 2 * nl,))
        lam2 = l2[nl - 1] if c.num_heads > 1 else 0.0          # stale loop index (sasrec/main.py:169)
        return float(slots[0] + slots[1] + (l1 * slots[2:2 + nl]).sum() + lam2 * slots[2 + nl:].sum())

    def backward(self):
        seq, dec, pos, neg, B, training, b_offset = self._batch
        self.G[:] = 0.0
        assert self.lib.adt_cpu_sasrec_backward(ctypes.byref(self.cfg), _p(self.P), _p(self.G), _p(self.ws(B)), _p(seq), _p(dec), _p(pos), _p(neg), B,
                                                training, _p(self.seed), b_offset, 0, None) == 0

    def clip_adam(self, weight_decay, clip=5.0, lr=1e-3, betas=(0.9, 0.98), eps=1e-8):
        nE = (self.cfg.item_num + 1) * self.cfg.hidden
        assert self.lib.adt_cpu_clip_adam(_p(self.P), _p(self.G), _p(self.M), _p(self.V), self.n, nE, weight_decay, clip, lr, betas[0], betas[1], eps,
                                          1.0, _p(self.scal), None) == 0
        return float(np.sqrt(self.scal[1])), float(self.scal[3])      # total gradient norm (before clipping), wd * ||E||

    def train_step(self, batch, lambdas1, lambdas2, weight_decay, seed=0, training=True, lr=1e-3, clip=5.0, norms=None, b_offset=0):
        """One pass of the loop body sasrec/main.py:143-173; returns (loss, gradient norm)."""
        self.forward(*batch, training=training, seed=seed, b_offset=b_offset)
        loss = self.loss_seed(lambdas1, lambdas2, norms)
        self.backward()
        tn, wdterm = self.clip_adam(weight_decay, clip, lr)
        return loss + wdterm, tn

    def predict(self, seq, cand=None):
        seq = self._ids(seq)
        B = seq.shape[0]
        C = self.cfg.item_num + 1 if cand is None else cand.shape[1]
        cand = None if cand is None else self._ids(cand)
        out = np.zeros((B, C), np.float32)
        assert self.lib.adt_cpu_sasrec_predict(ctypes.byref(self.cfg), _p(self.P), _p(self.ws(B)), _p(seq), _p(cand), B, C, _p(out), None) == 0
        return out
