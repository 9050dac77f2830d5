#!/usr/bin/env python3
"""HBM traffic of ONE whole training step from two rocprofv3 PMC passes of bench.py (FETCH_SIZE and WRITE_SIZE cannot share a pass; both in KiB;
on gfx950 FETCH_SIZE counts a 128-byte request as 64 bytes and is doubled here, WRITE_SIZE is exact for 16-byte-per-lane stores and float
atomics: MI355X_MICROARCH.md, "HBM").  Sums every kernel dispatched between two consecutive k_step_begin launches of the graph-replayed steps and
takes the median step; also prints the per-kernel breakdown of that step.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out/f -o fetch --output-format csv -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d out/w -o write --output-format csv -- python3 bench.py ...
    python tools/pmc_step_traffic.py out/f/fetch_counter_collection.csv out/w/write_counter_collection.csv profiles/r04_step_traffic.json
"""
import csv
import json
import sys
from collections import defaultdict


def steps(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    out, cur = [], None
    for r in rows:
        name = r["Kernel_Name"]
        if "k_step_begin" in name:
            if cur is not None:
                out.append(cur)
            cur = defaultdict(float)
        if cur is not None:
            cur[name.split("(")[0].replace("adt::", "").replace("void ", "")] += float(r["Counter_Value"])
    return out


def median_step(st):
    tot = sorted((sum(s.values()), i) for i, s in enumerate(st))
    return st[tot[len(tot) // 2][1]]


def main():
    fpath, wpath, out = sys.argv[1:4]
    f, w = steps(fpath, "FETCH_SIZE"), steps(wpath, "WRITE_SIZE")
    # graph-replayed steps only: the same kernel set in every step (drop the eager / probe steps, which differ in length)
    nker = max(set(len(s) for s in f), key=[len(s) for s in f].count)
    f = [s for s in f if len(s) == nker]
    nkw = max(set(len(s) for s in w), key=[len(s) for s in w].count)
    w = [s for s in w if len(s) == nkw]
    mf, mw = median_step(f), median_step(w)
    fetch = 2 * 1024 * sum(mf.values())
    write = 1024 * sum(mw.values())
    per = {k: {"fetch_bytes": 2 * 1024 * mf.get(k, 0.0), "write_bytes": 1024 * mw.get(k, 0.0)} for k in sorted(set(mf) | set(mw))}
    res = {"steps_sampled": [len(f), len(w)], "fetch_bytes": fetch, "write_bytes": write, "traffic_bytes": fetch + write,
           "ideal_bytes_per_step": 256 * 2.4e6, "traffic_over_ideal": (fetch + write) / (256 * 2.4e6),
           "note": "FETCH_SIZE doubled (gfx950 reports 64 B per 128-B request); separate --pmc passes; median graph-replayed step of bench.py (batch 256)",
           "per_kernel": per}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "per_kernel"}))
    for k, v in sorted(per.items(), key=lambda kv: -(kv[1]["fetch_bytes"] + kv[1]["write_bytes"])):
        print("%-44s fetch %8.1f MB  write %8.1f MB" % (k[:44], v["fetch_bytes"] / 1e6, v["write_bytes"] / 1e6))


if __name__ == "__main__":
    main()
