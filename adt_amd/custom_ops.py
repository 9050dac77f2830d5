"""PyTorch custom operators (torch.library) that put the HIP forward / backward of every model mirror behind torch.autograd, so
that the reference's own training-loop bodies -- torch loss assembly on the returned tensors, `loss.backward()`,
`clip_grad_norm_`, `torch.optim.Adam.step()` (sasrec/main.py:146-173, bert4rec/trainer.py:100-138, stosa/trainer.py:534-559) --
run unchanged on `SASRecADTWide`, `BertModel`, `DisenDistSAModel` and `SuperSASRecModel`.

    adt_amd::model_forward(Tensor[] params, Tensor[] ids, int handle, int call, bool training) -> Tensor[]
    adt_amd::model_backward(Tensor[] grads, int handle, int call) -> Tensor[]          (one gradient per parameter)

`handle` names a live model (registry below), `call` one forward pass whose launch tape (adt_amd/wide.py:Tape -- the closures that
replay the C-ABI backward kernels in reverse) is parked until its backward.  All arithmetic happens in libadt_hip.so; these
operators only sequence launches and hand device tensors to autograd.  The fused trainers (FusedBertTrainer, ...) do not go
through here: they keep loss assembly, clipping and Adam on the device and are the fast path.
"""
import itertools
import weakref
from typing import List

import torch
from torch import Tensor

_MODELS = weakref.WeakValueDictionary()     # handle -> model
_CALLS = {}                                 # call id -> state parked between forward and backward
_call_ids = itertools.count(1)
_MAX_PARKED = 8                             # forwards whose backward never ran (eval under grad mode, exceptions) are dropped


def register(model):
    h = id(model)
    _MODELS[h] = model
    return h


@torch.library.custom_op("adt_amd::model_forward", mutates_args=(), device_types="cuda")
def model_forward(params: List[Tensor], ids: List[Tensor], handle: int, call: int, training: bool) -> List[Tensor]:
    model = _MODELS[handle]
    outs, state = model._op_forward(ids, training)
    while len(_CALLS) >= _MAX_PARKED:
        _CALLS.pop(next(iter(_CALLS)))
    _CALLS[call] = state
    return list(outs)


@torch.library.custom_op("adt_amd::model_backward", mutates_args=(), device_types="cuda")
def model_backward(grads: List[Tensor], handle: int, call: int) -> List[Tensor]:
    model = _MODELS[handle]
    state = _CALLS.pop(call, None)
    if state is None:
        raise RuntimeError("adt_amd::model_backward: the forward pass %d was already consumed (retain_graph / double backward are "
                           "not supported: the launch tape is replayed once)" % call)
    return list(model._op_backward(state, grads))


def _setup_context(ctx, inputs, output):
    params, ids, handle, call, training = inputs
    ctx.handle, ctx.call = handle, call
    ctx.meta = [(tuple(o.shape), o.dtype, o.device) for o in output]
    ctx.nparams, ctx.nids = len(params), len(ids)


def _backward(ctx, grads):
    gs = [g if g is not None else torch.zeros(shape, dtype=dt, device=dev) for g, (shape, dt, dev) in zip(grads, ctx.meta)]
    pg = list(model_backward(gs, ctx.handle, ctx.call))
    model = _MODELS.get(ctx.handle)
    for i in getattr(model, "_op_none_idx", ()):      # parameters the reference leaves at grad None (unused modules)
        pg[i] = None
    return pg, [None] * ctx.nids, None, None, None


torch.library.register_autograd("adt_amd::model_forward", _backward, setup_context=_setup_context)


def forward_with_grad(model, ids):
    """Run model._op_forward under autograd: returns the list of output tensors, differentiable w.r.t. model.parameters()."""
    handle = getattr(model, "_op_handle", None)
    if handle is None:
        handle = model._op_handle = register(model)
    params = list(model.parameters())
    return model_forward(params, list(ids), handle, next(_call_ids), bool(model.training))


def wants_grad(model):
    """Training-mode forward under grad mode goes through the operator (its tape is parked for the backward); evaluation and
    no_grad calls take the plain path and park nothing."""
    return model.training and torch.is_grad_enabled() and any(p.requires_grad for p in model.parameters())


def take_grad(g, shape):
    """An incoming autograd gradient as a tensor the backward kernels may read and add into: contiguous, our own copy unless it is
    large (the all-item logits gradient of BERT4Rec-ADT is only read)."""
    g = g.reshape(shape)
    if not g.is_contiguous():
        return g.contiguous()
    return g if g.numel() * 4 > (64 << 20) else g.clone()


def param_grads(model, none_if=lambda name: False):
    """One gradient per parameter, in model.parameters() order: copies of the views into the flat gradient buffer."""
    out, none_idx = [], []
    for i, (name, p) in enumerate(model.named_parameters()):
        o, n, shape = model._views[name]
        if none_if(name):
            none_idx.append(i)
            out.append(torch.zeros(shape, dtype=torch.float32, device=model.flat_grad.device))
        else:
            out.append(model.flat_grad[o:o + n].view(shape).clone())
    model._op_none_idx = none_idx
    return out
