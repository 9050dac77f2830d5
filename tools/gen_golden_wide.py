#!/usr/bin/env python3
"""Generate tests/golden/bert_*.npz and stosa_*.npz by IMPORTING the reference (read-only, /root/reference/bert4rec and
/root/reference/stosa) in the build container.  Never runs on the GPU box (the reference does not travel); the .npz
fixtures are data only: seeded inputs, the seed that regenerates the numpy weights, and the tensors the reference
produced for them.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_wide.py [bert|stosa]

The loss assembly calls the same torch functions, in the same order, as the reference's training loops
(bert4rec/trainer.py:100-138, stosa/trainer.py:392-447 + :358-391), which are not importable as functions without the
datasets; dropout is 0 so that the ATen RNG stream does not enter.
"""
import importlib
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")


class Args:
    pass


def _import_from(root, modname):
    """Import `modname` with `root` first on sys.path, dropping same-named modules of the other backbone."""
    for k in [k for k in sys.modules if k.split(".")[0] in ("model", "modules", "models", "utils", "datasets", "trainer")]:
        del sys.modules[k]
    sys.path.insert(0, root)
    try:
        return importlib.import_module(modname)
    finally:
        sys.path.remove(root)


# =====================================================================================================================
def bert_batch(r, B, L, V, mask_prob=0.3):
    """Masked-item batch in the layout of BertTrainDataset (bert4rec/datasets/dataset.py:70-158): left-padded history,
    masked positions replaced by the [MASK] token V + 1 with the original item as label, label 0 elsewhere; the decoder
    input is the unmasked sequence."""
    src = np.zeros((B, L), np.int64)
    dec = np.zeros((B, L), np.int64)
    lab = np.zeros((B, L), np.int64)
    for b in range(B):
        n = L if b == 0 else int(r.randint(2, L + 1))
        items = r.randint(1, V + 1, size=n)
        m = r.rand(n) < mask_prob
        m[-1] = True
        dec[b, L - n:] = items
        src[b, L - n:] = np.where(m, V + 1, items)
        lab[b, L - n:] = np.where(m, items, 0)
    return src, dec, lab


def gen_bert(tag, cfg_kw, B, seed, lam1, lam2, wd=1e-4, lr=1e-3, clip=5.0, keep_w3=True, compact=False, steps=3):
    from oracle import bert_oracle as bo
    bert = _import_from("/root/reference/bert4rec", "model.bert")
    cfg = bo.Cfg(**cfg_kw)
    P = bo.init_params(cfg, seed)
    r = np.random.RandomState(seed + 1)
    # head classifier / mask bias away from their zero init so that their gradients are exercised
    for k in P:
        if k.endswith("head_classifier.bias") or k == "mask_bias" or (k.endswith(".bias") and "layer_norm" not in k):
            P[k] = (0.02 * r.standard_normal(P[k].shape)).astype(np.float32)
    src, dec, lab = bert_batch(r, B, cfg.maxlen, cfg.item_num)
    a = Args()
    a.maxlen, a.num_heads, a.num_layers, a.device, a.dropout = cfg.maxlen, cfg.num_heads, cfg.num_layers, "cpu", 0.0
    a.hidden_units, a.type_vocab_size, a.inner_units, a.attention_dropout = cfg.hidden_units, cfg.type_vocab_size, cfg.inner_units, 0.0
    m = bert.BertModel(1, cfg.item_num, a)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in P.items()}, strict=True)
    tsrc, tdec = torch.from_numpy(src), torch.from_numpy(dec)
    pos = torch.from_numpy(np.tile(np.arange(cfg.maxlen), (B, 1)))
    sent = torch.zeros(B, cfg.maxlen, dtype=torch.long)
    out = {"seed": seed, "src": src, "dec": dec, "labels": lab, "lambda1": np.array(lam1), "lambda2": np.array(lam2),
           "wd": wd, "lr": lr, "clip": clip, "cfg": np.array([cfg.item_num, cfg.maxlen, cfg.hidden_units, cfg.num_heads, cfg.num_layers, cfg.inner_units])}
    m.eval()
    with torch.no_grad():
        logits, enc_in, dec_out, rec = m(tsrc, tdec, pos, sent, pos, sent)
        out["logits"] = logits.numpy()
        for i in range(cfg.num_layers):
            out["enc_in_%d" % i], out["dec_out_%d" % i], out["rec_%d" % i] = enc_in[i].numpy(), dec_out[i].numpy(), rec[i].numpy()
        cand = r.randint(1, cfg.item_num + 1, size=(B, 11)).astype(np.int64)
        out["cand"] = cand
        out["predict"] = m.predict(None, tsrc, pos, sent, torch.from_numpy(cand)).numpy()
    # training steps: bert4rec/trainer.py:100-138
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=lr, betas=(0.9, 0.999), weight_decay=wd)
    ce = torch.nn.CrossEntropyLoss(ignore_index=0)
    labels = torch.from_numpy(lab)
    for step in range(steps):
        opt.zero_grad()
        logits, enc_in, dec_out, rec = m(tsrc, tdec, pos, sent, pos, sent)
        loss = ce(logits.view(-1, logits.size(-1)), labels.view(-1))
        if len(enc_in) != 0 and len(enc_in) == len(dec_out):
            for i in range(len(enc_in)):
                if lam1[i] != 0:
                    loss = loss + lam1[i] * F.mse_loss(enc_in[i], dec_out[i])
        if a.num_heads > 1 and len(rec) != 0:
            bs = rec[0].shape[0]
            label = torch.tile(torch.arange(a.num_heads), [bs * a.maxlen, 1])
            for l in range(len(rec)):
                if lam2[l] != 0:
                    loss = loss + lam2[l] * F.nll_loss(rec[l].view(bs * a.maxlen, a.num_heads, a.num_heads), label)
        loss.backward()
        if step == 0:
            out["loss"] = float(loss.item())
            for k, p in m.named_parameters():
                out["grad." + k] = p.grad.numpy().copy()
        tn = torch.nn.utils.clip_grad_norm_(m.parameters(), clip)
        if step == 0:
            out["grad_norm"] = float(tn)
        opt.step()
        if step == 0 or (step == 2 and keep_w3):
            for k, p in m.named_parameters():
                out["w%d." % (step + 1) + k] = p.detach().numpy().copy()
    path = os.path.join(OUT, "bert_%s.npz" % tag)
    if compact:
        from tools.gen_golden_inputs import compact as _compact
        out = _compact(out)
    np.savez_compressed(path, **out)
    print("wrote", path, "loss", out["loss"], "grad_norm", out["grad_norm"], "%.1f KB" % (os.path.getsize(path) / 1024))


def main():
    which = sys.argv[1:] or ["bert", "stosa"]
    if "bert" in which:
        # hidden sizes are multiples of 64 (the HIP LayerNorm / dense kernels' granularity) so that the same fixtures pin the
        # oracle on CPU and the HIP path on the GPU
        gen_bert("small", dict(item_num=40, maxlen=12, hidden_units=64, num_heads=2, num_layers=2, inner_units=96), B=4, seed=11,
                 lam1=[0.3, 0.2], lam2=[0.2, 0.1])
        gen_bert("h4", dict(item_num=30, maxlen=20, hidden_units=64, num_heads=4, num_layers=1, inner_units=128), B=3, seed=12,
                 lam1=[0.25], lam2=[0.15])
        gen_bert("hd64", dict(item_num=25, maxlen=9, hidden_units=128, num_heads=2, num_layers=1, inner_units=64), B=2, seed=13,
                 lam1=[0.1], lam2=[0.05], keep_w3=False)
    if "bert_cfg3" in which:
        # BASELINE configs[2]: BERT4Rec-ADT at the ml-20m shape (26,744 items => vocabulary 26,844; d=256, H=4, inner=1024, L=200,
        # 2 layers, get_lambda("ml-20m")), B=4: the reference's (4, 200, 26,844) logits are 86 MB; stored as norms + samples
        gen_bert("cfg3_ml20m", dict(item_num=26744, maxlen=200, hidden_units=256, num_heads=4, num_layers=2, inner_units=1024), B=4, seed=31,
                 lam1=[0.005435293808249262, 0.0019764407654292064], lam2=[0.0007068258408279514, 0.0013811031763964325], wd=1e-3,
                 keep_w3=False, compact=True, steps=1)
    if "stosa" in which:
        from tools import gen_golden_stosa
        gen_golden_stosa.main()
    if "stosa_cfg5" in which:
        from tools import gen_golden_stosa
        gen_golden_stosa.main_cfg5()


if __name__ == "__main__":
    main()
