cd $GRAFT_REPO_ROOT
python3 bench.py --steps 20 --warmup 5 --workload pubmed --cpu-rows -1 > gpurun_out/pubmed_new.json 2> gpurun_out/pubmed_new.err; tail -2 gpurun_out/pubmed_new.err; python -c "
import json; j=json.load(open('gpurun_out/pubmed_new.json')); print(j['ms_per_step'], j['config']['hipgraph'])"
python3 bench.py --steps 20 --warmup 5 --workload pubmed --edge-mode u-v-deg --cpu-rows -1 > gpurun_out/pubmed_new2.json 2> /dev/null; python -c "
import json; j=json.load(open('gpurun_out/pubmed_new2.json')); print(j['ms_per_step'])"
timeout 900 python -m pytest tests -q -m gpu -k "harness or cora or train" 2>&1 | tail -2
