cd $GRAFT_REPO_ROOT
for t in 128 64; do
DGG_BF16_TILE=$t python3 bench.py --steps 10 --warmup 3 --workload ppi --bf16 --cpu-rows -1 > gpurun_out/ppi_bf16_$t.json 2> /dev/null; python -c "
import json; j=json.load(open('gpurun_out/ppi_bf16_$t.json')); print($t, j['ms_per_step'], j['kernels_ms_per_step'])"
done
