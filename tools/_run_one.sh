cd $GRAFT_REPO_ROOT
timeout 2000 python -m pytest tests -q -m gpu -x -k "ranked and not symmetric" 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --no-variants --cpu-rows -1 > gpurun_out/bench_rk.json 2> /dev/null; python -c "
import json; j=json.load(open('gpurun_out/bench_rk.json')); print(j['ms_per_step'], j['roofline']['kernel_ms'])"
