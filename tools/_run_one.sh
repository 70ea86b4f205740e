cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_ranked_noise_stats.py -q -m gpu -k "ranked_symmetric" 2>&1 | grep -E "^E  |passed|failed|FAILED" | cut -c1-300 | head -20
python bench.py --steps 20 --warmup 5 --cpu-rows -1 > gpurun_out/bench_rsym.json 2> gpurun_out/bench_rsym.err; tail -3 gpurun_out/bench_rsym.err
python - <<'PY'
import json
j=json.load(open('gpurun_out/bench_rsym.json'))
print(j['ms_per_step'])
for k,v in j['variants'].items(): print(k, {a:b for a,b in v.items() if a!='roofline'}, {a:b for a,b in (v.get('roofline') or {}).items() if a not in('note','kernel')})
PY
