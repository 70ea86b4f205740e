timeout 900 python -m pytest tests -q -m gpu -x -k "sharded or parallel or rccl or knet or full_size_ranked or config3 or module_matches or cora" 2>&1 | tail -6
python3 bench.py --steps 20 --warmup 5 --no-variants --cpu-rows -1 --repeats 11 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['kernels']['allpairs_topk']['ms'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/trace -o h -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --repeats 2 --cpu-rows -1 --no-variants > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3k2; python3 tools/kernel_stats.py /tmp/trace/h_results.db gpurun_out/r3k2/kernel_stats.csv --skip-first 8 > /dev/null
grep -i "knet\|gemm_tn\|linear" gpurun_out/r3k2/kernel_stats.csv | cut -c1-50,100-
