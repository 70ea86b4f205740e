cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t3
python3 -m pytest tests/test_hip_parity.py -x -q -m gpu -k "wider_than_the_ell or edgelist or module_matches" 2>&1 | tail -5 > gpurun_out/t3/tests.log
python3 bench.py --steps 20 --warmup 5 --workload pubmed --cpu-rows -1 > gpurun_out/t3/pubmed.json 2> gpurun_out/t3/err.log
python3 bench.py --steps 20 --warmup 5 --workload pubmed --edge-mode u-v-deg --cpu-rows -1 > gpurun_out/t3/pubmed_deg.json 2>> gpurun_out/t3/err.log
python3 bench.py --steps 10 --warmup 3 --workload ppi --bf16 > gpurun_out/t3/ppi_bf16.json 2>> gpurun_out/t3/err.log
python3 bench.py --steps 10 --warmup 3 --workload ppi > gpurun_out/t3/ppi.json 2>> gpurun_out/t3/err.log
cat gpurun_out/t3/tests.log
python3 - <<'PY'
import json
for n in ['pubmed','pubmed_deg','ppi_bf16','ppi']:
    try:
        j=json.load(open(f'gpurun_out/t3/{n}.json')); print(n, j['ms_per_step'], j.get('kernels_ms_per_step'))
    except Exception as e: print(n, 'ERR', e)
PY
