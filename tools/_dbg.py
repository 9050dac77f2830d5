import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import sasrec_oracle as so
from tests.test_hip_model import build, load_golden
z, cfg = load_golden("tests/golden", "sasrec_small_h4")
P = {k[2:]: z[k] for k in z.files if k.startswith("w.")}
m = build(cfg, P, "f32"); m.train()
from adt_amd.sasrec.trainer import FusedTrainer
tr = FusedTrainer(m, list(z["lam1"]), list(z["lam2"]), lr=1e-3, weight_decay=float(z["wd"]), clip=5.0)
w0 = m.flat.clone()
tr.step(z["seq"], z["dec"], z["pos"], z["neg"])
print("loss", float(tr.loss()), float(z["loss"]), "gn", float(tr.grad_norm()), float(z["total_norm"]))
from adt_amd.sasrec import model as mm
print("loss slots", m.ws_view(3, mm.WS_LOSS, 0, 6).cpu().numpy())
for k,_ in so.param_shapes(cfg):
    if "g."+k in z.files:
        g = m.grad_view(k).cpu().numpy(); r = z["g."+k]
        e = np.abs(g-r).max()/max(np.abs(r).max(),1e-9)
        if e > 1e-4: print("BAD", k, e, np.abs(r).max())
g = m.grad_view("item_emb.weight").cpu().numpy(); r = z["g.item_emb.weight"]
rows = np.where(np.abs(g-r).max(1) > 1e-6)[0]
print("bad rows", rows[:20], "of", len(r))
for rr in rows[:5]:
    print(rr, g[rr][:4], r[rr][:4], "in seq", (z["seq"]==rr).sum(), "dec", (z["dec"]==rr).sum(), "pos", (z["pos"]==rr).sum(), "neg", (z["neg"]==rr).sum())
