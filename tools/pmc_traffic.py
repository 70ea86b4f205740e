#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes) into
profiles/r02_traffic.json: FABRIC bytes per launch (L2 -> fabric requests; Infinity-Cache hits are counted, the guide's
"HBM" section) for the kernels bench.py reports, stamped with the shape they were measured on.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --repeats 1 --cpu-rows -1 --no-hipgraph --no-variants
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 3 --warmup 1 --repeats 1 --cpu-rows -1 --no-hipgraph --no-variants
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r02_traffic.json [nodes feat latent]

gfx950 corrections: FETCH_SIZE counts 64 B per 128-B request -> doubled; both counters are in KiB."""
import csv
import glob
import json
import sys
from collections import defaultdict

# bench.py kernel key -> rocprof kernel-name substrings (several = one C-ABI call made of several launches: summed)
NAMES = {"allpairs_topk": ["allpairs_topk_ranked"], "spmm_fwd": ["spmm_fwd_narrow"], "conv_bwd": ["conv_bwd_node"],
         "edge_bwd": ["edge_bwd_rows", "edge_bwd_node"], "edge_bwd_rows": ["edge_bwd_rows"], "edge_bwd_node": ["edge_bwd_node"],
         "part_build": ["pp_count", "pp_scan", "pp_fill", "pp_sort"], "linear_fwd": ["linear_fwd_mfma", "linear_fwd_reg"],
         "gemm_tn_multi": ["gemm_tn_multi", "gemm_tn_wide", "gemm_tn_reduce_multi"],
         "knet_x_fwd": ["knet_x_fwd_reg"], "knet_x_bwd": ["knet_x_bwd_reg", "knet_bwd_reduce"], "softk_fwd": ["softk_fwd_kernel"],
         "normalize_fwd": ["normalize_fwd_kernel"]}


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
    mult = {}
    return {k: sum(acc[p_] / cnt[p_] * mult.get(p_, 1.0) for p_ in pats if cnt[p_]) for k, pats in NAMES.items()
            if any(cnt[p_] for p_ in pats)}


if __name__ == "__main__":
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    shape = {"nodes": int(sys.argv[4]), "feat": int(sys.argv[5]), "latent": int(sys.argv[6])} if len(sys.argv) > 6 else \
        {"nodes": 100000, "feat": 128, "latent": 64}
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        fb, wb = 2.0 * fetch.get(k, 0.0) * 1024.0, write.get(k, 0.0) * 1024.0
        kernels[k] = {"fetch_bytes_corrected": fb, "write_bytes": wb, "fabric_bytes_per_launch": fb + wb}
    out = {"shape": shape, "note": "fabric bytes (TCC_EA0 read / write requests): Infinity-Cache hits included, so this is an UPPER bound "
                                   "of the HBM traffic; FETCH_SIZE doubled per MI355X_MICROARCH.md", "kernels": kernels}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out, indent=1))
