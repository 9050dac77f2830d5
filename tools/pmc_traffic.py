#!/usr/bin/env python3
"""HBM traffic per launch of one kernel from two rocprofv3 PMC passes (MI355X_MICROARCH.md, "HBM" / "rocprofv3 PMC slots":
FETCH_SIZE and WRITE_SIZE cannot share a pass; both are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide
coalesced reads and is doubled here; WRITE_SIZE is exact for 16-B-per-lane stores and float atomics).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out/f -o fetch --output-format csv -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d out/w -o write --output-format csv -- python3 bench.py ...
    python tools/pmc_traffic.py out/f/fetch_counter_collection.csv out/w/write_counter_collection.csv k_attn_bwd profiles/r01_attn_bwd_pmc.json
"""
import csv
import json
import sys


def avg(path, counter, needle):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and needle in r["Kernel_Name"]]
    return sum(vals) / len(vals), len(vals)


def main():
    fpath, wpath, needle, out = sys.argv[1:5]
    f, nf = avg(fpath, "FETCH_SIZE", needle)
    w, nw = avg(wpath, "WRITE_SIZE", needle)
    res = {"kernel": needle, "launches_sampled": [nf, nw], "fetch_size_kib_raw": f, "write_size_kib_raw": w,
           "fetch_bytes": 2 * f * 1024, "write_bytes": w * 1024, "traffic_bytes": 2 * f * 1024 + w * 1024,
           "note": "FETCH_SIZE doubled (gfx950 reports 64 B per 128-B request); separate --pmc passes"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
