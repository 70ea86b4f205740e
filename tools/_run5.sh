R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3h
rm -rf $O; mkdir -p $O
python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/trace_none -o h -- python3 $R/bench.py --noise none --steps 10 --warmup 3 --repeats 1 --cpu-rows -1 --no-variants > $O/bench_none.json 2>/dev/null
cd $R; python3 tools/kernel_stats.py /tmp/trace_none/h_results.db $O/unperturbed_kernel_stats.csv --skip-first 3 > /dev/null
tail -c 3000 $O/bench.err
python3 - <<'PY'
import json
j=json.load(open('gpurun_out/r3h/bench.json'))
print(j['ms_per_step'], j['value'])
for k,v in j['variants'].items(): print(k, {a:b for a,b in v.items() if a!='roofline'}, v.get('roofline',{}).get('frac'))
PY
head -12 $O/unperturbed_kernel_stats.csv | cut -c1-60,100-
