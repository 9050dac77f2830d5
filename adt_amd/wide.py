"""Host-side plumbing shared by the BERT4Rec-ADT and STOSA-ADT models: one flat fp32 parameter buffer with the
reference's state_dict names as views, and a small launch tape that strings the C-ABI stage kernels
(include/adt_hip.h, "wide" section) into a forward pass and its reverse.

Nothing here computes on the host or through torch operators: tensors are allocated with torch (device memory,
streams, HIP-graph capture) and every arithmetic step is a kernel of libadt_hip.so.  A missing library raises.
"""
import numpy as np
import torch

from . import _lib, ops


class _Holder(torch.nn.Module):
    """Plain container so that named_parameters()/state_dict() reproduce the reference's dotted names.  A holder of exactly one
    2-D `weight` (an embedding table) is callable like the reference's nn.Embedding: the reference's trainers look item / user
    rows up through the module (stosa/trainer.py:361-364, stosa/models.py:236), a plain torch gather outside the hot path."""

    def forward(self, ids):
        w = self._parameters.get("weight")
        if w is None or w.dim() != 2:
            raise TypeError("this container holds parameters only; it is not a layer")
        ids = ids if isinstance(ids, torch.Tensor) else torch.as_tensor(np.asarray(ids))
        return torch.nn.functional.embedding(ids.to(device=w.device, dtype=torch.long), w)


def _set_nested(root, dotted, param):
    parts = dotted.split(".")
    m = root
    for p in parts[:-1]:
        if not hasattr(m, p):
            m.add_module(p, _Holder())
        m = getattr(m, p)
    m.register_parameter(parts[-1], param)


def ref_sorted(names, order):
    """`names` (dotted parameter names) in the order the reference's nn.Module tree registers them, which is the order of
    model.parameters() and therefore the key order of torch.optim.Adam.state_dict().  `order` lists path components in the
    order the reference's constructors create them (siblings only are ever compared); numeric components sort as integers.
    Pinned by tests/golden/param_order.json (tests/test_utils_cpu.py)."""
    rank = {c: i for i, c in enumerate(order)}

    def key(n):
        return tuple((0, int(c)) if c.isdigit() else (1, rank[c]) for c in n.split("."))
    return sorted(names, key=key)


class FlatModule(torch.nn.Module):
    """nn.Module whose parameters are views into ONE flat fp32 GPU buffer (`flat`), with an identically laid-out
    gradient buffer (`flat_grad`): the optimizer and the gradient all-reduce see one array.  Tensors start on 16-byte
    boundaries; tensors listed consecutively with sizes that are multiples of 4 floats are contiguous, which is what
    lets q/k/v projections run as one GEMM over a (3d, d) view."""

    def _build_flat(self, table, device, ref_order=None):
        self.lib = _lib.load()   # raises when the HIP library is missing: no fallback
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise _lib.AdtError("%s (adt_amd) needs a GPU device, got %r" % (type(self).__name__, device))
        self.table = list(table)
        off = 0
        self._views = {}
        for name, shape in self.table:
            n = int(np.prod(shape))
            self._views[name] = (off, n, tuple(shape))
            off += (n + 3) // 4 * 4
        self.n_flat = off
        self.flat = torch.zeros(off, device=self.dev, dtype=torch.float32)
        self.flat_grad = torch.zeros_like(self.flat)
        # The flat layout follows `table` (what the kernels want: q/k/v consecutive, never-trained tensors last); the nn.Module
        # registration follows the reference's constructor order so that parameters() lines up with the reference's optimizer.
        names = [n for n, _ in self.table]
        for name in (ref_sorted(names, ref_order) if ref_order else names):
            o, n, shape = self._views[name]
            _set_nested(self, name, torch.nn.Parameter(self.flat[o:o + n].view(shape), requires_grad=True))
        self._seed = torch.zeros(1, device=self.dev, dtype=torch.int32)   # uint32 bits of the dropout seed, device resident
        self._step_seed = 0
        self.dp_hook = None       # see Tape.mark_decoder_start

    def offset_of(self, name):
        """Flat offset (floats) of a parameter: bucket boundaries of the gradient all-reduce."""
        return self._views[name][0]

    def _apply(self, fn, recurse=True):
        probe = fn(self.flat)
        if probe.device != self.flat.device or probe.dtype != self.flat.dtype:
            raise _lib.AdtError("%s (adt_amd) parameters live in one flat fp32 GPU buffer; .to(%s, %s) is unsupported"
                                % (type(self).__name__, probe.device, probe.dtype))
        return self

    def P(self, name):
        o, n, shape = self._views[name]
        return self.flat[o:o + n].view(shape)

    def G(self, name):
        o, n, shape = self._views[name]
        return self.flat_grad[o:o + n].view(shape)

    def span(self, first, last, shape, grad=False):
        """One view over the consecutive tensors first..last (e.g. query/key/value weights as (3d, d))."""
        o0 = self._views[first][0]
        o1, n1, _ = self._views[last]
        buf = self.flat_grad if grad else self.flat
        v = buf[o0:o1 + n1]
        assert v.numel() == int(np.prod(shape)), (first, last, shape, v.numel())
        return v.view(shape)

    def set_seed(self, seed):
        self._seed.fill_(int(np.array([seed & 0xFFFFFFFF], dtype=np.uint32).view(np.int32)[0]))

    def next_seed(self):
        self._step_seed += 1
        self.set_seed(self._step_seed * 2654435761 + 12345)

    def ids(self, a):
        if isinstance(a, torch.Tensor):
            return a.to(device=self.dev, dtype=torch.int32).contiguous()
        return torch.from_numpy(np.ascontiguousarray(np.asarray(a), dtype=np.int32)).to(self.dev)

    def load_numpy(self, P):
        for k, v in P.items():
            self.P(k).copy_(torch.from_numpy(np.ascontiguousarray(v)))


class Act:
    """An activation and (lazily) its gradient buffer."""
    __slots__ = ("t", "g")

    def __init__(self, t):
        self.t = t
        self.g = None


def add_into(dst, src):
    """dst += src through the library's elementwise kernel (dropout off, identity activation)."""
    ops.dropact_bwd(src, src, 0.0, None, 0, dst, True)


def give(a, g):
    """Hand gradient `g` (complete, not used again by the caller) to activation `a`."""
    if a.g is None:
        a.g = g
    else:
        add_into(a.g, g)


class Tape:
    """Forward launches + the closures that undo them.  `prec`: ops.PREC_*; `seed`: device uint32; `row_offset`: first
    global token row of this data-parallel shard (dropout indices are global)."""

    def __init__(self, model, prec, training, row_offset=0, b_offset=0):
        self.m, self.prec, self.training = model, prec, training
        self.seed = model._seed
        self.row_offset, self.b_offset = row_offset, b_offset
        self.bw = []
        self.marks = {}

    def p_eff(self, p):
        return float(p) if self.training else 0.0

    def mark_decoder_start(self):
        """Called by a model's forward where its decoder stack begins: every closure recorded from here on belongs to the decoder
        (and the output head after it), whose parameters sit at the END of the flat buffer.  In the backward, `model.dp_hook`
        (the data-parallel trainers set it to GradBuckets.tail_ready) fires as soon as those closures have run, so that bucket's
        all-reduce overlaps the encoder's backward and the embedding scatter."""
        hook = getattr(self.m, "dp_hook", None)
        if hook is not None:
            self.marks[len(self.bw)] = hook

    def backward(self):
        for i in range(len(self.bw) - 1, -1, -1):
            self.bw[i]()
            hook = self.marks.get(i)
            if hook is not None:
                hook()
        self.bw, self.marks = [], {}

    # ---- Linear (+ activation, dropout, residual) -------------------------------------------------------------------
    def dense(self, x, W, b, gW, gb, act=ops.ACT_NONE, p=0.0, site=0, R=None, t_dev=None, ldy=None, R2=None, mask_ids=None):
        """y = mask(R + R2 + dropout(act(x W^T + b))).  With a row mask the residuals receive the masked gradient."""
        p = self.p_eff(p)
        Y, U = ops.dense_fwd(self.prec, x.t, W, b, act, act != ops.ACT_NONE, p, self.seed, site, self.row_offset, None if R is None else R.t,
                             mask_ids, None, t_dev, ldy, None if R2 is None else R2.t)
        y = Act(Y)

        def bw():
            if y.g is None:
                return
            if x.g is None:
                x.g = torch.empty_like(x.t)
                beta = False
            else:
                beta = True
            if act != ops.ACT_NONE and y.g.shape[1] >= 256:
                # wide layers with an activation: pull the gradient through the epilogue once (one elementwise pass) instead of
                # inside every column tile of the two backward GEMMs
                G = ops.dense_gradsrc(y.g, act, U, p, self.seed, site, self.row_offset, mask_ids, t_dev)
                ops.dense_bwd(self.prec, G, x.t, W, gW, gb, x.g, beta, ops.ACT_NONE, None, 0.0, None, 0, 0, None, t_dev)
            else:
                ops.dense_bwd(self.prec, y.g, x.t, W, gW, gb, x.g, beta, act, U, p, self.seed, site, self.row_offset, mask_ids, t_dev)
            for res in (R, R2):
                if res is None:
                    continue
                if mask_ids is None and R2 is None:
                    give(res, y.g)          # y.g is not used again: hand the buffer over
                else:
                    g = torch.empty_like(y.g)
                    ops.axpy(g, y.g, 1.0, False, mask_ids, y.g.shape[1])
                    give(res, g)
        self.bw.append(bw)
        return y

    def mix(self, parts):
        """sum_k w_k * a_k (supernet candidate mixing, sasrec/super_modules.py:42-49)."""
        out = torch.empty_like(parts[0][0].t)
        for k, (a, w) in enumerate(parts):
            ops.axpy(out, a.t, w, k > 0)
        y = Act(out)

        def bw():
            if y.g is None:
                return
            for a, w in parts:
                if a.g is None:
                    a.g = torch.empty_like(a.t)
                    ops.axpy(a.g, y.g, w, False)
                else:
                    ops.axpy(a.g, y.g, w, True)
        self.bw.append(bw)
        return y

    def log_softmax(self, x, H):
        y = Act(ops.log_softmax_fwd(x.t, H))

        def bw():
            if y.g is None:
                return
            acc = x.g is not None
            if not acc:
                x.g = torch.empty_like(x.t)
            ops.log_softmax_bwd(y.t, y.g, H, x.g, acc)
        self.bw.append(bw)
        return y

    def layernorm(self, x, w, b, gw, gb, eps):
        y = Act(ops.layernorm_fwd(x.t, w, b, eps))

        def bw():
            if y.g is None:
                return
            acc = x.g is not None
            if not acc:
                x.g = torch.empty_like(x.t)
            ops.layernorm_bwd(y.g, x.t, w, eps, x.g, acc, gw, gb)
        self.bw.append(bw)
        return y

    def dropact(self, x, p, site, act=ops.ACT_NONE):
        p = self.p_eff(p)
        if p == 0.0 and act == ops.ACT_NONE:
            return x
        off = self.row_offset * x.t.shape[1]
        y = Act(ops.dropact_fwd(x.t, p, self.seed, site, off, act))

        def bw():
            if y.g is None:
                return
            acc = x.g is not None
            if not acc:
                x.g = torch.empty_like(x.t)
            ops.dropact_bwd(y.g, x.t, p, self.seed, site, x.g, acc, off, act)
        self.bw.append(bw)
        return y

    def headcls(self, o, Ws, bs, gWs, gbs):
        """Independence head classifier + log_softmax on (T, H*hd) attention outputs, natural (b, l) row order."""
        T = o.t.shape[0]
        rec = Act(ops.headcls_fwd(o.t, Ws, bs, 1, T))

        def bw():
            if rec.g is None:
                return
            if o.g is None:
                o.g = torch.zeros_like(o.t)
            ops.headcls_bwd(o.t, Ws, rec.t, rec.g, 1, T, o.g, gWs, gbs)
        self.bw.append(bw)
        return rec
