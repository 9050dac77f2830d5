#!/usr/bin/env python3
"""Per-kernel LDS statistics from a rocprofv3 --pmc pass (counter_collection.csv), averaged per launch over the whole GPU.

    rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM \\
        --output-format csv -d out -o lds -- python3 bench.py ...
    python tools/pmc_lds.py out/lds_counter_collection.csv [name filter]
"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
flt = sys.argv[2] if len(sys.argv) > 2 else "k_seq"
for r in csv.DictReader(open(sys.argv[1])):
    if flt not in r["Kernel_Name"]:
        continue
    short = r["Kernel_Name"].replace("void adt::", "").replace("adt::", "").split("(")[0]
    acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    e = {c: sum(v) / len(v) for c, v in cs.items()}
    sqb = e.get("SQ_BUSY_CYCLES", 0.0)
    cu_cycles = sqb * 256.0 / 32.0 if sqb else float("nan")          # SQ_BUSY_CYCLES is summed over 32 shader engines: per-CU wall cycles x 256 CUs
    print("%-44s" % k[:44], " ".join("%s=%.3g" % (c.replace("SQ_", ""), v) for c, v in sorted(e.items())),
          "| lds_active/CU-cycle %.3f  conflict/active %.3f" % (e.get("SQ_LDS_IDX_ACTIVE", float("nan")) / cu_cycles if sqb else float("nan"),
                                                              e.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(e.get("SQ_LDS_IDX_ACTIVE", 1.0), 1.0)))
