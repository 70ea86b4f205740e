R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ppi; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/trace_ppi -o h -- python3 $R/bench.py --workload ppi --bf16 --steps 5 --warmup 2 > $O/ppi_bf16.json 2>/dev/null
cd $R; python3 tools/kernel_stats.py /tmp/trace_ppi/h_results.db $O/ppi_bf16_kernel_stats.csv --skip-first 0 > /dev/null
python3 - <<'PY'
import csv,json
print(json.load(open('gpurun_out/r3ppi/ppi_bf16.json'))['ms_per_step'])
rows=list(csv.reader(open('gpurun_out/r3ppi/ppi_bf16_kernel_stats.csv')))
for r in rows[:32]: print(r[0][:60].ljust(62), *r[1:4], r[-1])
PY
