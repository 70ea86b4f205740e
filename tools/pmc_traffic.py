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

NAMES = {"edge_bwd": "edge_bwd_kernel", "spmm_bwd": "spmm_bwd_kernel", "spmm_fwd": "spmm_fwd_kernel",
         "allpairs_topk_ranked": "allpairs_topk_ranked", "gemm_tn_partial": "gemm_tn_partial", "linear_fwd": "linear_fwd_mfma",
         "knet_x_fwd": "knet_x_fwd_kernel", "knet_x_bwd": "knet_x_bwd_kernel", "norm_bwd_da": "norm_bwd_da_kernel"}


def per_kernel(d, counter):
    acc, cnt = defaultdict(float), defaultdict(int)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            for key, pat in NAMES.items():
                if pat in r["Kernel_Name"]:
                    acc[key] += float(r["Counter_Value"])
                    cnt[key] += 1
    return {k: acc[k] / cnt[k] for k in acc}


if __name__ == "__main__":
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        fb, wb = 2.0 * fetch.get(k, 0.0) * 1024.0, write.get(k, 0.0) * 1024.0
        out[k] = {"fetch_bytes_corrected": fb, "write_bytes": wb, "hbm_bytes_per_launch": fb + wb}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out, indent=1))
