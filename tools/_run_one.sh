cd $GRAFT_REPO_ROOT
T0=$(date +%s.%N)
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
T1=$(date +%s.%N)
echo "bench.py wall seconds: $(echo "$T1 - $T0" | bc)"
python -c "
import json; j=json.load(open('gpurun_out/bench_default.json')); print(j['ms_per_step'], j['repeats']['timed_seconds_total'], j['cpu_baseline']['value'], list(j['variants'].keys()))"
