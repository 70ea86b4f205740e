cd $GRAFT_REPO_ROOT
for r in 128 64; do
DGG_LINEAR_MULTI_ROWS=$r python3 bench.py --steps 20 --warmup 5 --workload pubmed --cpu-rows -1 > gpurun_out/pubmed_$r.json 2> /dev/null; python -c "
import json; j=json.load(open('gpurun_out/pubmed_$r.json')); print($r, j['ms_per_step'])"
done
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "linear or module_matches" 2>&1 | tail -2
