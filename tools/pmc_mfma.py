#!/usr/bin/env python3
"""Per-kernel MFMA / issue statistics from rocprofv3 --pmc passes (counter_collection.csv files), averaged per launch over the whole GPU.

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU \\
        --output-format csv -d out -o sq -- python3 bench.py ...
    python tools/pmc_mfma.py profiles/r03_mfma.json label=out/sq_counter_collection.csv [label2=...]

mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel cycles), kernel cycles = SQ_BUSY_CYCLES-derived wall: the counter
counts cycles in which a SIMD's matrix pipe executes (MI355X_MICROARCH.md: 16 per v_mfma_f32_16x16x32_bf16, 32 per 32x32x16), SQ_BUSY_CYCLES
counts, per shader engine, the cycles in which the SQ has waves; both are sums over the chip, so their quotient x (32 SEs / 1024 SIMDs) is the
fraction of SIMD-cycles with the matrix pipe busy while the kernel runs.  mfma_flops = MOPS_BF16 x 512 (the counter is in units of 512 FLOPs)."""
import csv
import json
import sys
from collections import defaultdict

KEEP = ("k_seqtt", "k_seq_attn", "k_attn_gen", "k_wattn_mfma", "k_lce<", "k_dense", "k_attn_fwd", "k_attn_bwd", "k_dwpart", "k_embed")


def main():
    out = sys.argv[1]
    res = {"note": __doc__.split("\n\n")[-1].strip(), "runs": {}}
    for spec in sys.argv[2:]:
        label, path = spec.split("=", 1)
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"]
            if not any(k in name for k in KEEP):
                continue
            short = name.replace("void adt::", "").replace("adt::", "").split("(")[0]
            acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
        run = {}
        for k, cs in acc.items():
            e = {c: sum(v) / len(v) for c, v in cs.items()}
            e["launches"] = max(len(v) for v in cs.values())
            busy, sqb = e.get("SQ_VALU_MFMA_BUSY_CYCLES"), e.get("SQ_BUSY_CYCLES")
            if busy is not None and sqb:
                e["mfma_busy_frac"] = busy / (sqb * 1024.0 / 32.0)
            if e.get("SQ_INSTS_VALU_MFMA_MOPS_BF16") is not None:
                e["mfma_bf16_flops_per_launch"] = e["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512.0
            if e.get("SQ_WAVE_CYCLES"):
                for c, n in (("SQ_WAIT_ANY", "wait_any_frac"), ("SQ_WAIT_INST_ANY", "wait_issue_frac"), ("SQ_ACTIVE_INST_VALU", "valu_active_frac")):
                    if e.get(c) is not None:
                        e[n] = e[c] / e["SQ_WAVE_CYCLES"]
            run[k] = e
        res["runs"][label] = run
    json.dump(res, open(out, "w"), indent=1)
    for label, run in res["runs"].items():
        for k, e in sorted(run.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0)):
            print("%-10s %-46s mfma_busy %.4f  wait_any %.3f  valu_active %.3f" % (label, k[:46], e.get("mfma_busy_frac", float("nan")), e.get("wait_any_frac", float("nan")),
                                                                                  e.get("valu_active_frac", float("nan"))))


if __name__ == "__main__":
    main()
