"""SASRecADT on the MI355X hot path -- drop-in for the reference's sasrec/model.py:SASRecADT.

Same constructor, `forward(user_ids, log_seqs, dec_seqs, pos_seqs, neg_seqs)` 5-tuple, `predict(...)`,
`state_dict()` key names and shapes -- but every parameter is a view into ONE flat fp32
buffer and all arithmetic runs in libadt_hip.so (C ABI, include/adt_hip.h).  Two ways to train:

  * autograd-compatible: `forward()` returns tensors wired into torch.autograd (custom Function whose
    backward calls adt_sasrec_backward), so the reference's inline loop (sasrec/main.py:146-173: torch loss
    assembly, loss.backward(), clip_grad_norm_, torch.optim.Adam) runs unchanged;
  * fused: `train_step()` (adt_amd/sasrec/trainer.py) keeps loss assembly, clipping and Adam on the device
    in the same launch sequence -- the path bench.py measures.
"""
import ctypes

import numpy as np
import torch

from .. import _lib

# slot order of adt_sasrec_param_layout (include/adt_hip.h)
_ENC_SLOTS = ["attention_layernorm.weight", "attention_layernorm.bias", "attention_layer.in_proj_weight",
              "attention_layer.in_proj_bias", "attention_layer.out_proj.weight", "attention_layer.out_proj.bias",
              "forward_layernorm.weight", "forward_layernorm.bias", "forward_layer.conv1.weight",
              "forward_layer.conv1.bias", "forward_layer.conv2.weight", "forward_layer.conv2.bias",
              "sparse.weight", "sparse.bias"]
_DEC_SLOTS = ["layer_norm.weight", "layer_norm.bias", "slf_attn.in_proj_weight", "slf_attn.in_proj_bias",
              "slf_attn.out_proj.weight", "slf_attn.out_proj.bias", "enc_attn.in_proj_weight", "enc_attn.in_proj_bias",
              "enc_attn.out_proj.weight", "enc_attn.out_proj.bias", "pos_ffn.conv1.weight", "pos_ffn.conv1.bias",
              "pos_ffn.conv2.weight", "pos_ffn.conv2.bias", "pos_ffn_layernorm.weight", "pos_ffn_layernorm.bias"]

# path components in the order the reference's constructors register them (sasrec/model.py:18-28, modules.py:636-665)
REF_ORDER = ["item_emb", "pos_emb", "encoder", "decoder", "last_layernorm", "encoder_layers", "decoder_layers",
             "attention_layernorm", "attention_layer", "forward_layernorm", "forward_layer", "sparse",
             "layer_norm", "slf_attn", "enc_attn", "pos_ffn", "pos_ffn_layernorm",
             "in_proj_weight", "in_proj_bias", "out_proj", "conv1", "conv2", "weight", "bias"]

WS_ENC_X, WS_DEC_X, WS_REC, WS_POS_LOGITS, WS_NEG_LOGITS, WS_F = 0, 1, 2, 3, 4, 5
WS_G_ENC_X, WS_G_DEC_X, WS_G_REC, WS_G_POS, WS_G_NEG, WS_LOSS, WS_NORMS, WS_SCAL = 6, 7, 8, 9, 10, 11, 12, 13


def param_table(item_num, maxlen, d, H, nl):
    """[(state_dict name, shape)] in flat-slot order."""
    hd = d // H
    t = [("item_emb.weight", (item_num + 1, d)), ("pos_emb.weight", (maxlen, d)),
         ("last_layernorm.weight", (d,)), ("last_layernorm.bias", (d,))]
    enc_shapes = [(d,), (d,), (3 * d, d), (3 * d,), (d, d), (d,), (d,), (d,), (d, d, 1), (d,), (d, d, 1), (d,), (H, hd), (H,)]
    dec_shapes = [(d,), (d,), (3 * d, d), (3 * d,), (d, d), (d,), (3 * d, d), (3 * d,), (d, d), (d,), (d, d, 1), (d,),
                  (d, d, 1), (d,), (d,), (d,)]
    for i in range(nl):
        t += [("encoder.encoder_layers.%d.%s" % (i, n), s) for n, s in zip(_ENC_SLOTS, enc_shapes)]
    for i in range(nl):
        t += [("decoder.decoder_layers.%d.%s" % (i, n), s) for n, s in zip(_DEC_SLOTS, dec_shapes)]
    return t


class _Holder(torch.nn.Module):
    """Plain container so that named_parameters()/state_dict() reproduce the reference's dotted names."""


def _set_nested(root, dotted, param):
    parts = dotted.split(".")
    m = root
    for p in parts[:-1]:
        if not hasattr(m, p):
            m.add_module(p, _Holder())
        m = getattr(m, p)
    m.register_parameter(parts[-1], param)


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class SASRecADT(torch.nn.Module):
    def __init__(self, user_num, item_num, args):
        super().__init__()
        self.lib = _lib.load()   # raises when the HIP library is missing: no fallback
        self.user_num, self.item_num = user_num, item_num
        self.dev = torch.device(args.device)
        if self.dev.type != "cuda":
            raise _lib.AdtError("SASRecADT (adt_amd) needs a GPU device, got %r" % (args.device,))
        self.num_heads, self.maxlen, self.num_layers = args.num_heads, args.maxlen, args.num_layers
        self.hidden_units, self.dropout = args.hidden_units, _lib.dropout_rate(args.dropout, "dropout")
        self.args = args
        prec = getattr(args, "precision", "bf16")
        self.cfg = _lib.SasrecCfg(item_num, args.maxlen, args.hidden_units, args.num_heads, args.num_layers,
                                  self.dropout, {"f32": 0, "fp32": 0, "bf16": 1}[prec])
        nslots = 4 + 30 * args.num_layers
        offs = (ctypes.c_int64 * nslots)()
        total = self.lib.adt_sasrec_param_layout(ctypes.byref(self.cfg), offs)
        if total < 0:
            raise _lib.AdtError(self.lib.adt_last_error().decode())
        self.n_flat = int(total)
        self.offsets = [int(o) for o in offs]
        # one flat buffer; default init = torch defaults of the reference modules is irrelevant because
        # sasrec/main.py:95-99 re-initialises everything >= 2-D with xavier_normal_; 1-D: LN weight 1, rest 0
        self.flat = torch.zeros(self.n_flat, device=self.dev, dtype=torch.float32)
        self.flat_grad = torch.zeros_like(self.flat)
        self.table = param_table(item_num, args.maxlen, args.hidden_units, args.num_heads, args.num_layers)
        self._views = {}
        g = torch.Generator(device="cpu").manual_seed(torch.initial_seed() % (1 << 31))
        for (name, shape), off in zip(self.table, self.offsets):
            n = int(np.prod(shape))
            view = self.flat[off:off + n].view(shape)
            if name.endswith("norm.weight"):
                view.fill_(1.0)
            elif len(shape) >= 2:
                fan_out, fan_in = shape[0], shape[1] * (shape[2] if len(shape) > 2 else 1)
                bound = (1.0 / max(fan_in, 1)) ** 0.5
                view.copy_((torch.rand(shape, generator=g) * 2 - 1) * bound)
            self._views[name] = (off, n, shape)
        from ..wide import ref_sorted
        for name in ref_sorted([n for n, _ in self.table], REF_ORDER):   # registration order = the reference's (Adam-state interop)
            off, n, shape = self._views[name]
            _set_nested(self, name, torch.nn.Parameter(self.flat[off:off + n].view(shape), requires_grad=True))
        self._ws_cache = {}   # one arena per batch size, never freed: a captured HIP graph holds raw pointers into it
        self._seed = torch.zeros(1, device=self.dev, dtype=torch.int32)   # uint32 bits, device resident
        self._step_seed = 0

    # ------------------------------------------------------------------------------------------
    def _apply(self, fn, recurse=True):
        # .to(same device) / .cuda() must keep every parameter a view of the flat buffer
        probe = fn(self.flat)
        if probe.device != self.flat.device or probe.dtype != self.flat.dtype:
            raise _lib.AdtError("SASRecADT (adt_amd) parameters live in one flat fp32 GPU buffer; .to(%s, %s) is unsupported"
                                % (probe.device, probe.dtype))
        return self

    def grad_view(self, name):
        off, n, shape = self._views[name]
        return self.flat_grad[off:off + n].view(shape)

    def workspace(self, B):
        """The arena for batch size B.  Arenas are cached per B for the life of the model (train batch, trailing partial
        batch and eval batch each get their own), so predict() at another B can never free or move a buffer that a captured
        training graph still points into."""
        ws = self._ws_cache.get(B)
        if ws is None:
            n = self.lib.adt_sasrec_workspace_floats(ctypes.byref(self.cfg), B)
            ws = self._ws_cache[B] = torch.empty(int(n), device=self.dev, dtype=torch.float32)
        return ws

    def ws_view(self, B, what, layer, numel):
        off = self.lib.adt_sasrec_ws_offset(ctypes.byref(self.cfg), B, what, layer)
        return self.workspace(B)[off:off + numel]

    def set_seed(self, seed):
        self._seed.fill_(int(np.array([seed & 0xFFFFFFFF], dtype=np.uint32).view(np.int32)[0]))

    def _ids(self, a):
        if isinstance(a, torch.Tensor):
            return a.to(device=self.dev, dtype=torch.int32).contiguous()
        return torch.from_numpy(np.ascontiguousarray(np.asarray(a), dtype=np.int32)).to(self.dev)

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    # raw launches (no autograd); ids are device int32 tensors
    def run_forward(self, seq, dec, pos, neg, B, training, b_offset=0, packed=False):
        """packed: run_step_begin / run_step_begin_ring of THIS step already wrote the bf16 weight images (bit 1 of `training`)."""
        ws = self.workspace(B)
        _lib.check(self.lib.adt_sasrec_forward(ctypes.byref(self.cfg), _ptr(self.flat), _ptr(ws), _ptr(seq), _ptr(dec), _ptr(pos),
                                               _ptr(neg), B, int(bool(training)) | (2 if packed else 0), _ptr(self._seed), b_offset, self._stream()),
                   "sasrec_forward")

    def probe_dec_layer_forward(self, dec, B, layer, training=True, b_offset=0):
        """Measurement hook: only the fused forward launch of decoder layer `layer`, on the workspace of a completed run_forward."""
        _lib.check(self.lib.adt_sasrec_probe_dec_layer_fwd(ctypes.byref(self.cfg), _ptr(self.flat), _ptr(self.workspace(B)), _ptr(dec), B,
                                                           int(training), _ptr(self._seed), b_offset, layer, self._stream()), "probe_dec_layer_fwd")

    def run_step_begin(self, B, norms_src, scal, seed_inc=0x9E3779B1):
        """One launch: zero_grad, loss slots, normalisers, the dropout seed += seed_inc, ||E||^2 partials, the parameter-gradient replicas of
        the backward zeroed and (bf16) the step's weight images packed (adt_sasrec_step_begin)."""
        _lib.check(self.lib.adt_sasrec_step_begin(ctypes.byref(self.cfg), _ptr(self.workspace(B)), B, _ptr(self._seed), seed_inc, _ptr(norms_src),
                                                  _ptr(self.flat), _ptr(self.flat_grad), self.flat_grad.numel(), _ptr(scal), self._stream()),
                   "sasrec_step_begin")

    def run_step_begin_ring_staged(self, B, ring, slot_ints, nslots, ids_dst, state, consumed, staging, produced, scal, seed_inc=0x9E3779B1):
        """run_step_begin_ring that takes the batch from `staging` when the previous step's run_forward_loss(..., prefetch=...) copied it there
        (adt_sasrec_step_begin_ring_staged); `produced`: pinned host word with the producer's count of completely written batches."""
        _lib.check(self.lib.adt_sasrec_step_begin_ring_staged(ctypes.byref(self.cfg), _ptr(self.workspace(B)), B, _ptr(self._seed), seed_inc, _ptr(ring),
                                                              slot_ints, nslots, _ptr(ids_dst), _ptr(state), _ptr(consumed), _ptr(staging), _ptr(produced),
                                                              _ptr(self.flat), _ptr(self.flat_grad), self.flat_grad.numel(), _ptr(scal), self._stream()),
                   "sasrec_step_begin_ring_staged")

    def run_step_begin_ring(self, B, ring, slot_ints, nslots, ids_dst, state, consumed, scal, seed_inc=0x9E3779B1):
        """run_step_begin that first fetches the packed id batch of ring slot (state[0] % nslots) into ids_dst (adt_sasrec_step_begin_ring);
        `ring` / `consumed` are pinned host tensors (read / written by the kernel over PCIe) or device tensors."""
        _lib.check(self.lib.adt_sasrec_step_begin_ring(ctypes.byref(self.cfg), _ptr(self.workspace(B)), B, _ptr(self._seed), seed_inc, _ptr(ring),
                                                       slot_ints, nslots, _ptr(ids_dst), _ptr(state), _ptr(consumed), _ptr(self.flat),
                                                       _ptr(self.flat_grad), self.flat_grad.numel(), _ptr(scal), self._stream()), "sasrec_step_begin_ring")

    def bce_deferred(self):
        """True when run_forward_loss leaves logits + BCE seed to run_backward(..., bce=True) (adt_sasrec_bce_deferred)."""
        return bool(self.lib.adt_sasrec_bce_deferred(ctypes.byref(self.cfg)))

    def run_forward_loss(self, seq, dec, pos, neg, B, lambdas1, lambdas2, b_offset=0, training=True, packed=True, prefetch=None, bce_side=False):
        """run_forward(training) + run_loss_seed(zero_loss=False) of one step in one call (adt_sasrec_forward_loss).  Returns True when the
        logits + BCE seed were deferred to the backward: pass bce=True to run_backward of the same step -- or "fwd" when bce_side was set (the
        kernel is then launched here, on the side stream beside the loss pass, and the backward only joins it; needs run_step_begin* of this
        step, which zeroes the item-table replicas)."""
        nl = self.num_layers
        l1 = (ctypes.c_float * nl)(*[float(x) for x in lambdas1])
        l2 = (ctypes.c_float * nl)(*[float(x) for x in lambdas2])
        ring, slot_ints, nslots, state, consumed, staging = prefetch if prefetch is not None else (None, 0, 0, None, None, None)
        deferred = bool(training) and packed and self.bce_deferred()
        side = bool(bce_side) and deferred
        _lib.check(self.lib.adt_sasrec_forward_loss_prefetch(ctypes.byref(self.cfg), _ptr(self.flat), _ptr(self.workspace(B)), _ptr(seq), _ptr(dec),
                                                             _ptr(pos), _ptr(neg), B, int(bool(training)) | (2 if packed else 0) | (4 if side else 0), _ptr(self._seed),
                                                             b_offset, l1, l2, _ptr(ring), slot_ints, nslots, _ptr(state), _ptr(consumed),
                                                             _ptr(staging), self._stream()), "sasrec_forward_loss")
        return "fwd" if side else deferred

    def run_loss_seed(self, pos, B, lambdas1, lambdas2, zero_loss=True):
        nl = self.num_layers
        l1 = (ctypes.c_float * nl)(*[float(x) for x in lambdas1])
        l2 = (ctypes.c_float * nl)(*[float(x) for x in lambdas2])
        fn = self.lib.adt_sasrec_loss_seed if zero_loss else self.lib.adt_sasrec_loss_seed_nz
        _lib.check(fn(ctypes.byref(self.cfg), _ptr(self.workspace(B)), _ptr(pos), B, l1, l2, self._stream()), "sasrec_loss_seed")

    def run_backward(self, seq, dec, pos, neg, B, training, b_offset=0, phase=0, prezeroed=False, defer_fold=False, bce=False):
        """prezeroed: run_step_begin / run_step_begin_ring of THIS step already zeroed the parameter-gradient replicas (bit 2 of `phase`).
        defer_fold (phase 0 only): the last replica fold is left to run_fold_clip_adam, which must follow (bit 3).
        bce: what run_forward_loss of this step returned (pass it to EVERY phase: bits 4 / 5 in phases 0 and 1, bit 6 in phase 2 -- that forward
        leaves the reconstruction seeds of the block inputs to the backward kernels)."""
        _lib.check(self.lib.adt_sasrec_backward(ctypes.byref(self.cfg), _ptr(self.flat), _ptr(self.flat_grad), _ptr(self.workspace(B)),
                                                _ptr(seq), _ptr(dec), _ptr(pos), _ptr(neg), B, int(training), _ptr(self._seed),
                                                b_offset, phase | (4 if prezeroed else 0) | (8 if defer_fold and phase == 0 else 0) |
                                                ((32 if bce == "fwd" else 16) if bce and phase in (0, 1) else 0) | (64 if bce and phase == 2 else 0),
                                                self._stream()),
                   "sasrec_backward")

    def run_fold_clip_adam(self, B, m, v, wd, clip, lr, b1, b2, eps, scal):
        """Optimizer step behind run_backward(..., defer_fold=True): replica fold + weight-decay term + ||g||^2 in one pass, then clip + Adam."""
        _lib.check(self.lib.adt_sasrec_fold_clip_adam(ctypes.byref(self.cfg), _ptr(self.workspace(B)), B, _ptr(self.flat), _ptr(self.flat_grad),
                                                      _ptr(m), _ptr(v), float(wd), float(clip), float(lr), float(b1), float(b2), float(eps),
                                                      _ptr(scal), self._stream()), "sasrec_fold_clip_adam")

    def run_fold_grads(self, B, scal):
        """The gradient sums behind run_backward(..., defer_fold=True) without the optimizer step: flat_grad is complete afterwards (the
        data-parallel step all-reduces it, then clip_adam_pre)."""
        _lib.check(self.lib.adt_sasrec_fold_grads(ctypes.byref(self.cfg), _ptr(self.workspace(B)), B, _ptr(self.flat), _ptr(self.flat_grad), _ptr(scal),
                                                  self._stream()), "sasrec_fold_grads")

    # ------------------------------------------------------------------------------------------
    def forward(self, user_ids, log_seqs, dec_seqs, pos_seqs, neg_seqs):
        """sasrec/model.py:67-81.  Returns (pos_logits, neg_logits, encoder_layer_input, decoder_layer_output
        [reversed], rec_layer_ind) wired into autograd."""
        seq, dec, pos, neg = (self._ids(a) for a in (log_seqs, dec_seqs, pos_seqs, neg_seqs))
        if self.training:
            self._step_seed += 1
            self.set_seed(self._step_seed * 2654435761 + 12345)
        params = [p for _, p in self.named_parameters()]
        outs = _SasrecFn.apply(self, seq, dec, pos, neg, *params)
        nl = self.num_layers
        return outs[0], outs[1], list(outs[2:2 + nl]), list(outs[2 + nl:2 + 2 * nl]), list(outs[2 + 2 * nl:2 + 3 * nl])

    def predict(self, user_ids, log_seqs, item_indices, full=False):
        """sasrec/model.py:83-97: scores for the given candidates, or for every item when full=True."""
        logits, _ = self.predict_rank(log_seqs, None if full else item_indices, want_rank=False)
        return logits

    @torch.no_grad()
    def predict_rank(self, log_seqs, item_indices, want_rank=True):
        seq = self._ids(log_seqs)
        B = seq.shape[0]
        if item_indices is None:
            cand, C = None, self.item_num + 1
        else:
            cand = self._ids(item_indices)
            C = cand.shape[1]
        logits = torch.empty(B, C, device=self.dev, dtype=torch.float32)
        rank = torch.empty(B, device=self.dev, dtype=torch.int32) if want_rank else None
        _lib.check(self.lib.adt_sasrec_predict(ctypes.byref(self.cfg), _ptr(self.flat), _ptr(self.workspace(B)), _ptr(seq), _ptr(cand),
                                               B, C, _ptr(logits), _ptr(rank), self._stream()), "sasrec_predict")
        return logits, rank


class _SasrecFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, seq, dec, pos, neg, *params):
        B, L = seq.shape
        d, H, nl = model.hidden_units, model.num_heads, model.num_layers
        T = B * L
        model.run_forward(seq, dec, pos, neg, B, model.training)
        ctx.model, ctx.ids, ctx.training = model, (seq, dec, pos, neg), model.training
        outs = [model.ws_view(B, WS_POS_LOGITS, 0, T).view(B, L).clone(), model.ws_view(B, WS_NEG_LOGITS, 0, T).view(B, L).clone()]
        outs += [model.ws_view(B, WS_ENC_X, i, T * d).view(B, L, d).clone() for i in range(nl)]
        outs += [model.ws_view(B, WS_DEC_X, nl - j, T * d).view(B, L, d).clone() for j in range(nl)]   # reversed
        outs += [model.ws_view(B, WS_REC, i, T * H * H).view(B, L, H, H).clone() for i in range(nl)]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        model = ctx.model
        seq, dec, pos, neg = ctx.ids
        B, L = seq.shape
        d, H, nl = model.hidden_units, model.num_heads, model.num_layers
        T = B * L

        def put(view, g):
            if g is None:
                view.zero_()
            else:
                view.copy_(g.reshape(-1))
        put(model.ws_view(B, WS_G_POS, 0, T), gouts[0])
        put(model.ws_view(B, WS_G_NEG, 0, T), gouts[1])
        for i in range(nl):
            put(model.ws_view(B, WS_G_ENC_X, i, T * d), gouts[2 + i])
            put(model.ws_view(B, WS_G_DEC_X, nl - i, T * d), gouts[2 + nl + i])
            put(model.ws_view(B, WS_G_REC, i, T * H * H), gouts[2 + 2 * nl + i])
        model.flat_grad.zero_()
        model.run_backward(seq, dec, pos, neg, B, ctx.training)
        grads = []
        for name, _ in model.table:
            if "pos_ffn_layernorm" in name or (H == 1 and ".sparse." in name):
                grads.append(None)   # torch leaves these None (sasrec/modules.py:664 unused; main.py:160)
            else:
                grads.append(model.grad_view(name).clone())
        order = {n: k for k, (n, _) in enumerate(model.table)}
        named = [n for n, _ in model.named_parameters()]
        return (None, None, None, None, None) + tuple(grads[order[n]] for n in named)
