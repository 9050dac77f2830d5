#!/usr/bin/env python3
"""Record the reference models' parameter registration order (names only: data, not source) -> tests/golden/param_order.json.
torch.optim.Adam(model.parameters()) keys its state by position in this order, so optimizer-state interop
(adt_amd/checkpoint.py) depends on it.  Imports /root/reference: build container only."""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.dont_write_bytecode = True


class A:
    pass


def fresh(path):
    for k in list(sys.modules):
        if k in ("model", "modules", "utils", "models", "supersasrec", "super_modules", "base_super_modules", "supernet") or k.startswith("model."):
            del sys.modules[k]
    sys.path[:] = [p for p in sys.path if not p.startswith(REF)]
    sys.path.insert(0, path)


def main():
    out = {}
    np.float = float
    fresh(REF + "/sasrec")
    import model as sm
    a = A()
    a.device, a.num_heads, a.maxlen, a.num_layers, a.hidden_units, a.dropout = "cpu", 2, 8, 2, 16, 0.0
    out["sasrec_nl2"] = [n for n, _ in sm.SASRecADT(5, 20, a).named_parameters()]
    import supersasrec as ss
    m = ss.SuperSASRecModel(5, 20, [0.0, 0.1, 0.2, 0.3, 0.4, 0.5], [0.0, 0.1, 0.2, 0.3, 0.4, 0.5], a)
    out["supersasrec_nl2_c6"] = [n for n, _ in m.named_parameters()]
    fresh(REF + "/bert4rec")
    from model import bert as bm
    a = A()
    a.device, a.num_heads, a.maxlen, a.num_layers, a.hidden_units, a.inner_units = "cpu", 2, 8, 2, 16, 32
    a.dropout, a.attention_dropout, a.type_vocab_size, a.init_val = 0.0, 0.0, 2, 0.02
    out["bert_nl2"] = [n for n, _ in bm.BertModel(5, 20, a).named_parameters()]
    fresh(REF + "/stosa")
    import models as tm
    a = A()
    a.item_size, a.maxlen, a.hidden_units, a.num_heads, a.num_layers, a.num_users = 22, 8, 16, 2, 2, 5
    a.dropout = a.attention_dropout = 0.0
    a.hidden_act, a.initializer_range, a.distance_metric, a.cuda_condition = "gelu", 0.02, "wasserstein", False
    a.pvn_weight, a.kernel_param = 0.1, 1.0
    out["stosa_nl2"] = [n for n, _ in tm.DisenDistSAModel(a).named_parameters()]
    with open(os.path.join(REPO, "tests", "golden", "param_order.json"), "w") as f:
        json.dump(out, f, indent=0)
    for k, v in out.items():
        print(k, len(v))


if __name__ == "__main__":
    main()
