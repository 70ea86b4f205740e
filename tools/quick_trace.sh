#!/bin/bash
# kernel trace of the headline step, single stream (DGG_OVERLAP=0): per-kernel average durations -> gpurun_out/<name>_kernel_stats.csv
#   gpurun -- tools/quick_trace.sh <name> [bench.py args]
set -eu
R="${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
name=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/qt_$name
DGG_OVERLAP=0 rocprofv3 --kernel-trace --stats -d /tmp/qt_$name -o h -- python3 "$R/bench.py" --steps 20 --warmup 5 --repeats 3 --cpu-rows -1 --no-variants --no-configs "$@" > /dev/null 2>&1
cd "$R"
python3 tools/kernel_stats.py /tmp/qt_$name/h_results.db gpurun_out/${name}_kernel_stats.csv --skip-first 8 > /dev/null
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/${name}_kernel_stats.csv")))
for r in rows[:24]:
    print(f"{float(r['avg_us']):8.1f} us x{r['calls']:>4}  {r['kernel'][:70]}")
PY
