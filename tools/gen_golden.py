#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING the reference (read-only, /root/reference/sasrec) in the build
container.  Never runs on the GPU box (the reference does not travel); the .npz fixtures are data only:
seeded inputs, numpy-generated weights (by state_dict name, or the seed that regenerates them) and the
tensors the reference produced for them.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py

The loss assembly below calls the same torch functions, in the same order, as the reference's inline loop
(sasrec/main.py:146-173); it is driven here because that loop is not importable as a function.
"""
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/sasrec"
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import sasrec_oracle as so  # noqa: E402
from tools.gen_golden_inputs import make_batch, sample_idx  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")


class Args:
    pass


def build_ref(cfg, P):
    import model as ref_model  # /root/reference/sasrec/model.py
    a = Args()
    a.device = "cpu"
    a.num_heads = cfg.num_heads
    a.maxlen = cfg.maxlen
    a.num_layers = cfg.num_layers
    a.hidden_units = cfg.hidden_units
    a.dropout = cfg.dropout
    m = ref_model.SASRecADT(1, cfg.item_num, a)
    sd = {k: torch.from_numpy(np.array(v)) for k, v in P.items()}
    m.load_state_dict(sd, strict=True)
    return m, a


def ref_train_steps(cfg, P, batch, lam1, lam2, wd, lr, clip, nsteps):
    """The loop body of sasrec/main.py:146-173, dropout = 0."""
    model, args = build_ref(cfg, P)
    model.train()
    bce = torch.nn.BCEWithLogitsLoss()
    opt = torch.optim.Adam(model.parameters(), lr=lr, betas=(0.9, 0.98))
    seq, dec, pos, neg = batch
    rec = {}
    for step in range(nsteps):
        pos_logits, neg_logits, enc_in, dec_out, rec_ind = model(np.zeros(len(seq)), seq, dec, pos, neg)
        pos_labels, neg_labels = torch.ones(pos_logits.shape), torch.zeros(neg_logits.shape)
        opt.zero_grad()
        indices = np.where(pos != 0)
        loss = bce(pos_logits[indices], pos_labels[indices])
        loss += bce(neg_logits[indices], neg_labels[indices])
        if len(enc_in) != 0 and len(enc_in) == len(dec_out):
            for i in range(len(enc_in)):
                loss += lam1[i] * F.mse_loss(enc_in[i], dec_out[i])
        if args.num_heads > 1:
            batch_size = rec_ind[0].shape[0]
            label = torch.arange(args.num_heads)
            label = torch.tile(label, [batch_size * args.maxlen, 1])
            for l in range(len(rec_ind)):
                loss += lam2[i] * F.nll_loss(rec_ind[l].view(batch_size * args.maxlen, args.num_heads, args.num_heads), label)
        for param in model.item_emb.parameters():
            loss += wd * torch.norm(param)
        loss.backward()
        if step == 0:
            rec["loss"] = float(loss.item())
            rec["pos_logits"] = pos_logits.detach().numpy().copy()
            rec["neg_logits"] = neg_logits.detach().numpy().copy()
            rec["enc_in"] = [t.detach().numpy().copy() for t in enc_in]
            rec["dec_out"] = [t.detach().numpy().copy() for t in dec_out]
            rec["rec_ind"] = [t.detach().numpy().copy() for t in rec_ind]
            rec["grads"] = {k: (None if p.grad is None else p.grad.detach().numpy().copy())
                            for k, p in model.named_parameters()}
        tn = torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
        if step == 0:
            rec["total_norm"] = float(tn)
        opt.step()
        rec["loss_step%d" % step] = float(loss.item())
        rec["weights_step%d" % step] = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
    return rec


def case_small(name, B, L, d, H, nl, V, seed, lam1, lam2, wd):
    cfg = so.Cfg(V, L, d, H, nl, dropout=0.0)
    P = so.init_params(cfg, seed=seed)
    r = np.random.RandomState(seed + 1)
    batch = make_batch(r, B, L, V)
    rec = ref_train_steps(cfg, P, batch, lam1, lam2, wd, 1e-3, 5.0, 3)
    out = dict(cfg=np.array([V, L, d, H, nl], np.int64), seed=np.int64(seed),
               lam1=np.array(lam1, np.float64), lam2=np.array(lam2, np.float64), wd=np.float64(wd),
               seq=batch[0], dec=batch[1], pos=batch[2], neg=batch[3],
               loss=np.float64(rec["loss"]), total_norm=np.float64(rec["total_norm"]),
               pos_logits=rec["pos_logits"], neg_logits=rec["neg_logits"])
    for i in range(nl):
        out["enc_in.%d" % i] = rec["enc_in"][i]
        out["dec_out.%d" % i] = rec["dec_out"][i]
        out["rec_ind.%d" % i] = rec["rec_ind"][i]
    for k, v in P.items():
        out["w." + k] = v
    for k, g in rec["grads"].items():
        if g is None:
            out["gnone." + k] = np.int8(1)
        else:
            out["g." + k] = g
    for s in (0, 2):
        for k, v in rec["weights_step%d" % s].items():
            out["w%d.%s" % (s + 1, k)] = v
        out["loss_step%d" % (s + 1)] = np.float64(rec["loss_step%d" % s])
    # predict (eval mode), candidates + full
    model, _ = build_ref(cfg, P)
    model.eval()
    cand = r.randint(1, V + 1, size=(B, 11)).astype(np.int64)
    with torch.no_grad():
        out["cand"] = cand
        out["predict_cand"] = model.predict(np.zeros(B), batch[0], cand).numpy()
        out["predict_full"] = model.predict(np.zeros(B), batch[0], None, full=True).numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "loss", rec["loss"], "norm", rec["total_norm"])


def case_cfga(name, B, seed):
    """cfg-A shaped slice (L=200, d=64, H=2, 2 blocks, V=3416): weights are regenerated from
    `seed` by oracle.init_params, outputs stored as norms + strided samples to keep the fixture small."""
    V, L, d, H, nl = 3416, 200, 64, 2, 2
    lam1, lam2 = [0.104292, 0.065892], [0.100833, 0.000607]  # sasrec/utils.py:855-856
    wd = 1e-3
    cfg = so.Cfg(V, L, d, H, nl, dropout=0.0)
    P = so.init_params(cfg, seed=seed)
    r = np.random.RandomState(seed + 1)
    batch = make_batch(r, B, L, V)
    rec = ref_train_steps(cfg, P, batch, lam1, lam2, wd, 1e-3, 5.0, 1)
    out = dict(cfg=np.array([V, L, d, H, nl], np.int64), seed=np.int64(seed), B=np.int64(B),
               lam1=np.array(lam1), lam2=np.array(lam2), wd=np.float64(wd),
               loss=np.float64(rec["loss"]), total_norm=np.float64(rec["total_norm"]),
               pos_logits=rec["pos_logits"], neg_logits=rec["neg_logits"])
    for i in range(nl):
        for nm in ("enc_in", "dec_out", "rec_ind"):
            t = rec[nm][i].reshape(-1)
            out["%s.%d.norm" % (nm, i)] = np.float64(np.sqrt((t.astype(np.float64) ** 2).sum()))
            out["%s.%d.sample" % (nm, i)] = t[sample_idx(t.size, 1024)]
    for k, g in rec["grads"].items():
        if g is None:
            out["gnone." + k] = np.int8(1)
        else:
            t = g.reshape(-1)
            out["gnorm." + k] = np.float64(np.sqrt((t.astype(np.float64) ** 2).sum()))
            out["gsample." + k] = t[sample_idx(t.size)]
    for k, v in rec["weights_step0"].items():
        t = v.reshape(-1)
        out["w1sample." + k] = t[sample_idx(t.size)]
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "loss", rec["loss"], "norm", rec["total_norm"])


def case_sampled(name, V, L, d, H, nl, B, seed, lam1, lam2, wd=1e-3):
    """Wide configuration (the shipped template: d = 256, H = 2 => head size 128, sasrec/templates/ml-1m.json:11-12) on a
    short sequence: outputs and gradients as norms + strided samples (weights regenerate from `seed`)."""
    cfg = so.Cfg(V, L, d, H, nl, dropout=0.0)
    P = so.init_params(cfg, seed=seed)
    r = np.random.RandomState(seed + 1)
    batch = make_batch(r, B, L, V)
    rec = ref_train_steps(cfg, P, batch, lam1, lam2, wd, 1e-3, 5.0, 1)
    out = dict(cfg=np.array([V, L, d, H, nl], np.int64), seed=np.int64(seed), B=np.int64(B), lam1=np.array(lam1), lam2=np.array(lam2),
               wd=np.float64(wd), loss=np.float64(rec["loss"]), total_norm=np.float64(rec["total_norm"]), pos_logits=rec["pos_logits"],
               neg_logits=rec["neg_logits"])
    for i in range(nl):
        for nm in ("enc_in", "dec_out", "rec_ind"):
            t = rec[nm][i].reshape(-1)
            out["%s.%d.norm" % (nm, i)] = np.float64(np.sqrt((t.astype(np.float64) ** 2).sum()))
            out["%s.%d.sample" % (nm, i)] = t[sample_idx(t.size, 1024)]
    for k, g in rec["grads"].items():
        if g is None:
            out["gnone." + k] = np.int8(1)
        else:
            t = g.reshape(-1)
            out["gnorm." + k] = np.float64(np.sqrt((t.astype(np.float64) ** 2).sum()))
            out["gsample." + k] = t[sample_idx(t.size)]
    for k, v in rec["weights_step0"].items():
        out["w1sample." + k] = v.reshape(-1)[sample_idx(v.size)]
    model, _ = build_ref(cfg, P)
    model.eval()
    cand = r.randint(1, V + 1, size=(B, 11)).astype(np.int64)
    with torch.no_grad():
        out["cand"] = cand
        out["predict_cand"] = model.predict(np.zeros(B), batch[0], cand).numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "loss", rec["loss"], "norm", rec["total_norm"])


def case_metrics():
    """evaluate_loader (sasrec/utils.py:395-428) on preset score matrices."""
    import utils as ref_utils
    r = np.random.RandomState(5)
    scores = r.randn(3, 64, 101).astype(np.float32)

    class FakeModel:
        def __init__(self):
            self.i = 0

        def predict(self, u, seq, item_idx):
            s = torch.from_numpy(scores[self.i])
            self.i += 1
            return s

    loader = []
    for i in range(3):
        u = torch.zeros(64, dtype=torch.int64)
        seq = torch.zeros(64, 4, dtype=torch.int64)
        item_idx = torch.zeros(64, 101, dtype=torch.int64)
        loader.append(((u, seq, item_idx), torch.zeros(64)))
    (ndcg, hr), auc = ref_utils.evaluate_loader(FakeModel(), loader, None, "val", [5, 10])
    np.savez_compressed(os.path.join(OUT, "metrics_kat.npz"), scores=scores,
                        ndcg5=ndcg[5], ndcg10=ndcg[10], hr5=hr[5], hr10=hr[10], auc=auc)
    print("metrics", ndcg, hr, auc)


def case_data():
    """data_partition / WarpDataset.sample_data / EvalDataset.sample_data on a tiny hand-made file."""
    import utils as ref_utils
    lines = []
    hist = {1: [3, 5, 7, 9, 2, 4], 2: [8, 1], 3: [6, 6, 2, 9, 1, 3, 5, 7, 8, 4, 2], 4: [5, 9, 3],
            5: list(range(10, 41)), 6: list(range(40, 9, -1)), 7: [11, 13, 17, 19, 23, 29, 31, 37]}
    for u, items in hist.items():
        for it in items:
            lines.append("%d %d" % (u, it))
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "data"))
    with open(os.path.join(tmp, "data", "toy.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    os.chdir(tmp)
    try:
        ut, uv, ute, usernum, itemnum = ref_utils.data_partition("toy")
    finally:
        os.chdir(cwd)
    out = dict(text=np.array("\n".join(lines) + "\n"), usernum=usernum, itemnum=itemnum)
    L = 5
    np.random.seed(0)
    wd = ref_utils.WarpDataset(ut, usernum, itemnum, L)
    for u in hist:
        out["train.%d" % u] = np.array(ut[u], np.int64)
        out["valid.%d" % u] = np.array(uv[u], np.int64)
        out["test.%d" % u] = np.array(ute[u], np.int64)
        _, seq, dec, pos, neg = wd.sample_data(u)
        out["warp.seq.%d" % u] = seq
        out["warp.dec.%d" % u] = dec
        out["warp.pos.%d" % u] = pos
        out["warp.negmask.%d" % u] = (neg != 0)
    sampler = ref_utils.PopularSampler(ut, uv, ute, usernum, itemnum, 3)
    out["popular_p"] = np.array(sampler.popular_p, np.float64)
    for mode in ("val", "test"):
        ed = ref_utils.EvalDataset(ut, uv, ute, usernum, itemnum, L, sampler, mode=mode, eval_set=-1)
        out["eval.%s.users" % mode] = np.array(ed.users, np.int64)
        for u in ed.users:
            _, seq, item_idx, label = ed.sample_data(u)
            out["eval.%s.seq.%d" % (mode, u)] = seq
            out["eval.%s.first.%d" % (mode, u)] = np.int64(item_idx[0])
            out["eval.%s.ncand.%d" % (mode, u)] = np.int64(len(item_idx))
    np.savez_compressed(os.path.join(OUT, "data_kat.npz"), **out)
    print("data ok", usernum, itemnum)


def case_config():
    """get_lambdas tables (sasrec/utils.py:850-862), templates, candidates_to_lambdas KAT."""
    import utils as ref_utils
    out = {}
    for ds in ("ml-1m", "beauty", "Beauty", "steam", "ml-20m"):
        l1, l2 = ref_utils.get_lambdas(ds)
        out["lam1." + ds] = np.array(l1)
        out["lam2." + ds] = np.array(l2)
    sys.path.insert(0, "/root/reference")
    import candidates_to_lambdas as c2l
    rec_choice = [0, 0.0001, 0.0005, 0.001, 0.005, 0.01]
    cand = [0.7053411308078107, 0.9542592593410837, 0.9296478828883573, 0.28425047269448145,
            0.1600125621449342, 0.47495464861462977]
    out["c2l.choices"] = np.array(rec_choice)
    out["c2l.cand"] = np.array(cand)
    out["c2l.out"] = np.array([c2l._get_weight(rec_choice, c) for c in cand])
    np.savez_compressed(os.path.join(OUT, "config_kat.npz"), **out)
    print("config ok")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    if sys.argv[1:] == ["beauty"]:
        # BASELINE configs[3]: SASRec-ADT at the Amazon-Beauty template shape (sasrec/templates/beauty.json: d=256, H=2, L=50,
        # wd 1e-4; sasrec/data/beauty.txt has 54,542 items; get_lambdas("beauty")), B=8
        case_sampled("sasrec_cfg4_beauty", V=54542, L=50, d=256, H=2, nl=2, B=8, seed=31, lam1=[0.0124, 0.122], lam2=[0.0001, 0.0], wd=1e-4)
        sys.exit(0)
    case_small("sasrec_small", B=4, L=16, d=32, H=2, nl=2, V=50, seed=11,
               lam1=[0.104292, 0.065892], lam2=[0.100833, 0.000607], wd=1e-3)
    case_small("sasrec_small_h1", B=3, L=12, d=16, H=1, nl=1, V=30, seed=13, lam1=[0.05], lam2=[0.02], wd=1e-4)
    case_small("sasrec_small_h4", B=3, L=20, d=64, H=4, nl=1, V=40, seed=19, lam1=[0.07], lam2=[0.3], wd=1e-3)
    case_small("sasrec_small_l3", B=5, L=24, d=32, H=2, nl=3, V=80, seed=17,
               lam1=[0.01, 0.2, 0.03], lam2=[0.3, 0.2, 0.05], wd=0.0)
    case_cfga("sasrec_cfga_b8", B=8, seed=23)
    case_sampled("sasrec_d256_h2", V=50, L=24, d=256, H=2, nl=1, B=3, seed=29, lam1=[0.104292], lam2=[0.100833])
    case_metrics()
    case_data()
    case_config()
