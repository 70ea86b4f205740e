cd $GRAFT_REPO_ROOT
timeout 2000 python -m pytest tests -q -m gpu -k "unperturbed or k_limit or klimit or allpairs_topk_bit_exact or sharded or parallel" -s 2>&1 | grep -E "clustered=|passed|failed|^E  " | cut -c1-200
python bench.py --noise none --steps 10 --warmup 3 --no-variants --cpu-rows -1 > gpurun_out/bench_none.json 2> /dev/null; python -c "
import json; j=json.load(open('gpurun_out/bench_none.json')); print(j['ms_per_step'], j['roofline']['kernel_ms'])"
