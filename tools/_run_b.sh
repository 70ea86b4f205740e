R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3b2; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x -k "sharded or parallel or rccl or bench or linear or knet or full_size_ranked" 2>&1 | tail -4
python3 bench.py --steps 20 --warmup 5 --no-variants --cpu-rows -1 > $O/bench.json 2> $O/bench.err
tail -c 1500 $O/bench.err
python3 - <<'PY'
import json
j=json.load(open('gpurun_out/r3b2/bench.json'))
print(j['ms_per_step'], j['value'], j['repeats'])
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/trace -o h -- python3 $R/bench.py --steps 20 --warmup 5 --repeats 2 --cpu-rows -1 --no-variants > /dev/null 2>&1
cd $R; python3 tools/kernel_stats.py /tmp/trace/h_results.db $O/kernel_stats.csv --skip-first 8 > /dev/null
python3 - <<'PY'
import csv
rows=list(csv.reader(open('gpurun_out/r3b2/kernel_stats.csv')))
tot=0
for r in rows[1:40]:
    print(r[0][:60].ljust(62), *r[1:4], r[-1])
PY
