#!/bin/bash
# as quick_trace.sh, but the default (two-stream, hipGraph) step
set -eu
R="${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
name=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/qt_$name
rocprofv3 --kernel-trace --stats -d /tmp/qt_$name -o h -- python3 "$R/bench.py" --steps 20 --warmup 5 --repeats 3 --cpu-rows -1 --no-variants --no-configs "$@" > /tmp/qt_$name.json 2>/dev/null
cd "$R"
python3 tools/kernel_stats.py /tmp/qt_$name/h_results.db gpurun_out/${name}_kernel_stats.csv --skip-first 8 > /dev/null
python3 - <<PY
import csv, json
rows=list(csv.DictReader(open("gpurun_out/${name}_kernel_stats.csv")))
print("ms/step", json.loads(open("/tmp/qt_$name.json").read().strip().splitlines()[-1])["ms_per_step"])
tot=0
for r in rows[:22]:
    if int(r['calls'])>50: tot+=float(r['avg_us'])
    print(f"{float(r['avg_us']):8.1f} us x{r['calls']:>4}  {r['kernel'][:70]}")
print("sum of per-step kernels", tot)
PY
