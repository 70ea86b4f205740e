# round-5 profile collection (run on the GPU box): bash tools/prof_r05.sh  -> gpurun_out/r05/
set -eu
R="${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
O="$R/gpurun_out/r05"
rm -rf -- "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
BA="--steps 3 --warmup 1 --repeats 1 --cpu-rows -1 --no-hipgraph --no-variants"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch -- python3 "$R/bench.py" $BA > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write -- python3 "$R/bench.py" $BA > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_mfma -- python3 "$R/bench.py" $BA > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_sq -- python3 "$R/bench.py" $BA > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/trace -o h -- python3 "$R/bench.py" --steps 20 --warmup 5 --repeats 3 --cpu-rows -1 --no-variants > "$O/bench_under_profiler.json" 2>/dev/null
DGG_OVERLAP=0 rocprofv3 --kernel-trace --stats -d /tmp/trace_ss -o h -- python3 "$R/bench.py" --steps 20 --warmup 5 --repeats 3 --cpu-rows -1 --no-variants --no-configs > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/trace_k128 -o h -- python3 "$R/bench.py" --prior 100,164 --steps 10 --warmup 3 --repeats 3 --cpu-rows -1 --no-variants --no-configs > "$O/r05_k128_bench_under_profiler.json" 2>/dev/null
for nz in none rsym hash; do
  rocprofv3 --kernel-trace --stats -d /tmp/trace_$nz -o h -- python3 "$R/bench.py" --noise $nz --steps 10 --warmup 3 --repeats 2 --cpu-rows -1 --no-variants > /dev/null 2>&1
done
rocprofv3 --kernel-trace --stats -d /tmp/trace_ppi -o h -- python3 "$R/bench.py" --steps 4 --warmup 2 --workload ppi --bf16 --graphs 20 --cpu-rows -1 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/trace_pub -o h -- python3 "$R/bench.py" --steps 20 --warmup 5 --workload pubmed --cpu-rows -1 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/trace_pubdeg -o h -- python3 "$R/bench.py" --steps 20 --warmup 5 --workload pubmed --edge-mode u-v-deg --cpu-rows -1 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/trace_emu -o h -- python3 "$R/bench.py" --steps 20 --warmup 5 --repeats 2 --emulate-world 8 --nodes 62500 --cpu-rows -1 --no-variants > /dev/null 2>&1
cd "$R"
python3 tools/pmc_traffic.py /tmp/pmc_fetch /tmp/pmc_write "$O/r05_traffic.json" 100000 128 64 > /dev/null
python3 tools/mfma_busy.py /tmp/pmc_mfma "$O/r05_mfma_busy.csv" > /dev/null
python3 tools/sq_breakdown.py /tmp/pmc_sq "$O/r05_sq_breakdown.csv" > /dev/null
python3 tools/kernel_stats.py /tmp/trace/h_results.db "$O/r05_kernel_stats.csv" --skip-first 8 > /dev/null
python3 tools/kernel_stats.py /tmp/trace_ss/h_results.db "$O/r05_kernel_stats_single_stream.csv" --skip-first 8 > /dev/null
python3 tools/kernel_stats.py /tmp/trace_k128/h_results.db "$O/r05_k128_chunked_rows_kernel_stats.csv" --skip-first 4 > /dev/null
python3 tools/kernel_stats.py /tmp/trace_none/h_results.db "$O/r05_unperturbed_kernel_stats.csv" --skip-first 3 > /dev/null
python3 tools/kernel_stats.py /tmp/trace_rsym/h_results.db "$O/r05_symmetric_kernel_stats.csv" --skip-first 3 > /dev/null
python3 tools/kernel_stats.py /tmp/trace_hash/h_results.db "$O/r05_hash_kernel_stats.csv" --skip-first 3 > /dev/null
python3 tools/kernel_stats.py /tmp/trace_ppi/h_results.db "$O/r05_ppi_bf16_kernel_stats.csv" --skip-first 0 > /dev/null
python3 tools/kernel_stats.py /tmp/trace_pub/h_results.db "$O/r05_pubmed_kernel_stats.csv" --skip-first 0 > /dev/null
python3 tools/kernel_stats.py /tmp/trace_pubdeg/h_results.db "$O/r05_pubmed_uvdeg_kernel_stats.csv" --skip-first 0 > /dev/null
python3 tools/kernel_stats.py /tmp/trace_emu/h_results.db "$O/r05_emulated_rank_of_8_kernel_stats.csv" --skip-first 8 > /dev/null
cp "$O/r05_traffic.json" profiles/r05_traffic.json     # bench.py reads the newest traffic file from here
python3 bench.py --steps 20 --warmup 5 > "$O/r05_bench.json" 2> "$O/bench.err"
python3 bench.py --steps 20 --warmup 5 --workload pubmed > "$O/r05_pubmed_uvdist_bench.json" 2>> "$O/bench.err"
python3 bench.py --steps 20 --warmup 5 --workload pubmed --edgelist-api modules --cpu-rows -1 > "$O/r05_pubmed_uvdist_separate_modules_bench.json" 2>> "$O/bench.err"
python3 bench.py --steps 20 --warmup 5 --workload pubmed --edge-mode u-v-deg --cpu-rows -1 > "$O/r05_pubmed_uvdeg_bench.json" 2>> "$O/bench.err"
python3 bench.py --steps 20 --warmup 5 --workload pubmed --graph cora --cpu-rows -1 > "$O/r05_cora_uvdist_bench.json" 2>> "$O/bench.err"
python3 bench.py --steps 20 --warmup 5 --workload pubmed --graph cora --edge-mode u-v-deg --cpu-rows -1 > "$O/r05_cora_uvdeg_bench.json" 2>> "$O/bench.err"
python3 bench.py --steps 20 --warmup 5 --workload pubmed --edge-mode u-v-deg --edgelist-api modules --cpu-rows -1 > "$O/r05_pubmed_uvdeg_separate_modules_bench.json" 2>> "$O/bench.err"
python3 bench.py --steps 10 --warmup 3 --workload ppi --bf16 --graphs 20 > "$O/r05_ppi_bf16_bench.json" 2>> "$O/bench.err"
python3 bench.py --steps 10 --warmup 3 --prior 100,164 --no-variants --no-configs > "$O/r05_k128_chunked_rows_bench.json" 2>> "$O/bench.err"
python3 bench.py --emulate-world 8 --nodes 62500 --no-variants --no-configs --cpu-rows -1 > "$O/r05_emulated_rank_of_8_strong_500k.json" 2>> "$O/bench.err"
python3 bench.py --nodes 500000 --no-variants --no-configs > "$O/r05_bench_n500k_1gpu.json" 2>> "$O/bench.err"
python3 bench.py --steps 20 --warmup 5 --nodes 500000 --no-variants --cpu-rows -1 > "$O/r05_bench_n500k_1gpu.json" 2>> "$O/bench.err"
python3 bench.py --steps 20 --warmup 5 --emulate-world 8 --nodes 62500 --no-variants --cpu-rows -1 > "$O/r05_emulated_rank_of_8_strong_500k.json" 2>> "$O/bench.err"
python3 bench.py --steps 20 --warmup 5 --emulate-world 8 --nodes 62500 --exchange replicate --no-variants --cpu-rows -1 > "$O/r05_emulated_rank_of_8_strong_500k_replicate.json" 2>> "$O/bench.err"
ls -la "$O"; tail -c 400 "$O/bench.err"
python3 - <<'PY'
import json
j=json.load(open('gpurun_out/r05/r05_bench.json'))
print(j['ms_per_step'], j['roofline']['frac'], {k:(v.get('ms_per_step'), v.get('pair_kernel_ms'), (v.get('roofline') or {}).get('frac')) for k,v in j['variants'].items()})
PY
# single-stream trace of the headline step (per-kernel durations that add up to the step: the default step runs two kernels beside their neighbours)
cd /tmp
DGG_OVERLAP=0 rocprofv3 --kernel-trace --stats -d /tmp/trace_ss -o h -- python3 "$R/bench.py" --steps 20 --warmup 5 --repeats 3 --cpu-rows -1 --no-variants > /dev/null 2>&1
cd "$R"
python3 tools/kernel_stats.py /tmp/trace_ss/h_results.db "$O/r05_kernel_stats_single_stream.csv" --skip-first 8 > /dev/null
DGG_OVERLAP=0 python3 bench.py --steps 20 --warmup 5 --no-variants --cpu-rows -1 > "$O/r05_bench_single_stream.json" 2>> "$O/bench.err"
