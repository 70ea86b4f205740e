#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes) into
profiles/r01_traffic.json: HBM bytes per launch for the kernels bench.py reports.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --cpu-rows -1 --no-hipgraph
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 3 --warmup 1 --cpu-rows -1 --no-hipgraph
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_traffic.json

gfx950 corrections: FETCH_SIZE counts 64 B per 128-B request -> doubled; both counters are in KiB."""
import csv
import glob
import json
import sys
from collections import defaultdict

# bench.py kernel key -> rocprof kernel-name substrings (several = one C-ABI call made of several launches: summed)
NAMES = {"allpairs_topk": ["allpairs_topk_ranked"], "spmm_fwd": ["spmm_fwd_kernel"], "spmm_bwd": ["sddmm_pair_kernel"],
         "edge_bwd": ["edge_bwd_rows", "edge_bwd_cols"], "edge_bwd_rows": ["edge_bwd_rows"],
         "edge_bwd_cols": ["edge_bwd_cols"], "norm_da_cols": ["norm_da_cols"],
         "part_build": ["part_pass", "part_sort", "part_scan"], "gemm_tn_partial": ["gemm_tn_partial"],
         "linear_fwd": ["linear_fwd_mfma"], "knet_x_fwd": ["knet_x_fwd_tpn"], "knet_x_bwd": ["knet_x_bwd_tpn"],
         "softk_bwd": ["softk_bwd_kernel"]}


def per_kernel(d, counter):
    acc, cnt = defaultdict(float), defaultdict(int)                  # per kernel-name substring
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            for pats in NAMES.values():
                for pat in pats:
                    if pat in r["Kernel_Name"]:
                        acc[pat] += float(r["Counter_Value"])
                        cnt[pat] += 1
    # average per launch of each kernel; launches of one call summed (part_pass runs twice per call: count both)
    mult = {"part_pass": 2.0}
    return {k: sum(acc[p_] / cnt[p_] * mult.get(p_, 1.0) for p_ in pats if cnt[p_]) for k, pats in NAMES.items()
            if any(cnt[p_] for p_ in pats)}


if __name__ == "__main__":
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        fb, wb = 2.0 * fetch.get(k, 0.0) * 1024.0, write.get(k, 0.0) * 1024.0
        out[k] = {"fetch_bytes_corrected": fb, "write_bytes": wb, "hbm_bytes_per_launch": fb + wb}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out, indent=1))
