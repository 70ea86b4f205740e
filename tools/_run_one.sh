cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/tr -o h -- python3 $GRAFT_REPO_ROOT/bench.py --noise none --steps 10 --warmup 3 --repeats 2 --cpu-rows -1 --no-variants > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python3 tools/kernel_stats.py /tmp/tr/h_results.db /tmp/ks.csv --skip-first 3 > /dev/null; python3 - <<'PY'
import csv
for r in list(csv.reader(open('/tmp/ks.csv')))[1:40]:
    if 'sw_' in r[0]: print(r[0][18:50], r[1], r[3])
PY
