cd $GRAFT_REPO_ROOT
python bench.py --noise rsym --steps 10 --warmup 3 --no-variants --cpu-rows -1 > gpurun_out/bench_rsym_main.json 2> gpurun_out/bench_rsym_main.err; tail -2 gpurun_out/bench_rsym_main.err; python -c "
import json; j=json.load(open('gpurun_out/bench_rsym_main.json')); print(j['ms_per_step'], j['config']['hipgraph'], j['repeats']['eager_ms_per_step'], j['roofline']['kernel_ms'])"
python bench.py --noise rsym --nodes 500000 --steps 10 --warmup 3 --no-variants --cpu-rows -1 > gpurun_out/bench_rsym_500k.json 2> /dev/null; python -c "
import json; j=json.load(open('gpurun_out/bench_rsym_500k.json')); print('500k', j['ms_per_step'], j['roofline']['kernel_ms'])"
python bench.py --noise rsym --emulate-world 8 --nodes 62500 --steps 10 --warmup 3 --no-variants --cpu-rows -1 > gpurun_out/bench_rsym_emu.json 2> /dev/null; python -c "
import json; j=json.load(open('gpurun_out/bench_rsym_emu.json')); print('rank of 8', j['ms_per_step'], j['roofline']['kernel_ms'])"
