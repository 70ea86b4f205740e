# round-6 profile collection (run on the GPU box): bash tools/prof_r06.sh  -> gpurun_out/r06/
set -eu
R="${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
O="$R/gpurun_out/r06"
rm -rf -- "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
BA="--steps 3 --warmup 1 --repeats 1 --cpu-rows -1 --no-hipgraph --no-variants --no-configs"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch -- python3 "$R/bench.py" $BA > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write -- python3 "$R/bench.py" $BA > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_mfma -- python3 "$R/bench.py" $BA > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_sq -- python3 "$R/bench.py" $BA > /dev/null 2>&1
TR="--steps 20 --warmup 5 --repeats 3 --cpu-rows -1 --no-variants --no-configs"
rocprofv3 --kernel-trace --stats -d /tmp/trace -o h -- python3 "$R/bench.py" $TR > "$O/r06_bench_under_profiler.json" 2>/dev/null
DGG_OVERLAP=0 rocprofv3 --kernel-trace --stats -d /tmp/trace_ss -o h -- python3 "$R/bench.py" $TR > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/trace_k128 -o h -- python3 "$R/bench.py" --prior 100,164 --steps 10 --warmup 3 --repeats 3 --cpu-rows -1 --no-variants --no-configs > /dev/null 2>&1
for nz in none rsym; do
  rocprofv3 --kernel-trace --stats -d /tmp/trace_k128_$nz -o h -- python3 "$R/bench.py" --prior 100,164 --noise $nz --steps 5 --warmup 2 --repeats 2 --cpu-rows -1 --no-variants --no-configs > /dev/null 2>&1
  rocprofv3 --kernel-trace --stats -d /tmp/trace_$nz -o h -- python3 "$R/bench.py" --noise $nz --steps 10 --warmup 3 --repeats 2 --cpu-rows -1 --no-variants --no-configs > /dev/null 2>&1
done
rocprofv3 --kernel-trace --stats -d /tmp/trace_x16 -o h -- python3 "$R/bench.py" --feat-scale 16 --steps 5 --warmup 3 --repeats 2 --cpu-rows -1 --no-variants --no-configs > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/trace_ppi -o h -- python3 "$R/bench.py" --steps 4 --warmup 2 --workload ppi --bf16 --graphs 20 --cpu-rows -1 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/trace_pub -o h -- python3 "$R/bench.py" --steps 20 --warmup 5 --workload pubmed --cpu-rows -1 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/trace_pubdeg -o h -- python3 "$R/bench.py" --steps 20 --warmup 5 --workload pubmed --edge-mode u-v-deg --cpu-rows -1 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/trace_emu -o h -- python3 "$R/bench.py" --steps 20 --warmup 5 --repeats 2 --emulate-world 8 --nodes 62500 --cpu-rows -1 --no-variants > /dev/null 2>&1
cd "$R"
python3 tools/pmc_traffic.py /tmp/pmc_fetch /tmp/pmc_write "$O/r06_traffic.json" 100000 128 64 > /dev/null
python3 tools/mfma_busy.py /tmp/pmc_mfma "$O/r06_mfma_busy.csv" > /dev/null
python3 tools/sq_breakdown.py /tmp/pmc_sq "$O/r06_sq_breakdown.csv" > /dev/null
python3 tools/kernel_stats.py /tmp/trace/h_results.db "$O/r06_kernel_stats.csv" --skip-first 8 > /dev/null
python3 tools/kernel_stats.py /tmp/trace_ss/h_results.db "$O/r06_kernel_stats_single_stream.csv" --skip-first 8 > /dev/null
python3 tools/kernel_stats.py /tmp/trace_k128/h_results.db "$O/r06_k128_chunked_rows_kernel_stats.csv" --skip-first 4 > /dev/null
python3 tools/kernel_stats.py /tmp/trace_k128_none/h_results.db "$O/r06_k128_unperturbed_kernel_stats.csv" --skip-first 2 > /dev/null
python3 tools/kernel_stats.py /tmp/trace_k128_rsym/h_results.db "$O/r06_k128_symmetric_kernel_stats.csv" --skip-first 2 > /dev/null
python3 tools/kernel_stats.py /tmp/trace_none/h_results.db "$O/r06_unperturbed_kernel_stats.csv" --skip-first 3 > /dev/null
python3 tools/kernel_stats.py /tmp/trace_rsym/h_results.db "$O/r06_symmetric_kernel_stats.csv" --skip-first 3 > /dev/null
python3 tools/kernel_stats.py /tmp/trace_x16/h_results.db "$O/r06_features_x16_kernel_stats.csv" --skip-first 3 > /dev/null
python3 tools/kernel_stats.py /tmp/trace_ppi/h_results.db "$O/r06_ppi_bf16_kernel_stats.csv" --skip-first 0 > /dev/null
python3 tools/kernel_stats.py /tmp/trace_pub/h_results.db "$O/r06_pubmed_kernel_stats.csv" --skip-first 0 > /dev/null
python3 tools/kernel_stats.py /tmp/trace_pubdeg/h_results.db "$O/r06_pubmed_uvdeg_kernel_stats.csv" --skip-first 0 > /dev/null
python3 tools/kernel_stats.py /tmp/trace_emu/h_results.db "$O/r06_emulated_rank_of_8_kernel_stats.csv" --skip-first 8 > /dev/null
cp "$O/r06_traffic.json" profiles/r06_traffic.json     # bench.py reads the newest traffic file from here
# the driver's command, its one line and its detail file; then the other workloads (detail files: the full result dicts)
DGG_BENCH_DETAIL=gpurun_out/r06/r06_bench.json python3 bench.py > "$O/r06_bench_line.json" 2> "$O/bench.err"
DGG_OVERLAP=0 DGG_BENCH_DETAIL=gpurun_out/r06/r06_bench_single_stream.json python3 bench.py --no-variants --no-configs --cpu-rows -1 > /dev/null 2>> "$O/bench.err"
DGG_BENCH_DETAIL=gpurun_out/r06/r06_pubmed_uvdist_bench.json python3 bench.py --steps 20 --warmup 5 --workload pubmed > /dev/null 2>> "$O/bench.err"
DGG_BENCH_DETAIL=gpurun_out/r06/r06_pubmed_uvdeg_bench.json python3 bench.py --steps 20 --warmup 5 --workload pubmed --edge-mode u-v-deg --cpu-rows -1 > /dev/null 2>> "$O/bench.err"
DGG_BENCH_DETAIL=gpurun_out/r06/r06_cora_uvdist_bench.json python3 bench.py --steps 20 --warmup 5 --workload pubmed --graph cora > /dev/null 2>> "$O/bench.err"
DGG_BENCH_DETAIL=gpurun_out/r06/r06_ppi_bf16_bench.json python3 bench.py --steps 10 --warmup 3 --workload ppi --bf16 --graphs 20 > /dev/null 2>> "$O/bench.err"
DGG_BENCH_DETAIL=gpurun_out/r06/r06_bench_n500k_1gpu.json python3 bench.py --nodes 500000 --no-variants --no-configs > /dev/null 2>> "$O/bench.err"
DGG_BENCH_DETAIL=gpurun_out/r06/r06_emulated_rank_of_8_strong_500k.json python3 bench.py --emulate-world 8 --nodes 62500 --no-variants --no-configs --cpu-rows -1 > /dev/null 2>> "$O/bench.err"
# the N > 1 step's collectives on the one GPU (one-rank RCCL group): captured (the default since round 6) and eager
DGG_FORCE_COLLECTIVES=1 DGG_BENCH_DETAIL=gpurun_out/r06/r06_forced_collectives_captured.json python3 bench.py --nodes 62500 --no-variants --no-configs --cpu-rows -1 > /dev/null 2>> "$O/bench.err"
DGG_FORCE_COLLECTIVES=1 DGG_BENCH_GRAPH_DIST=0 DGG_BENCH_DETAIL=gpurun_out/r06/r06_forced_collectives_eager.json python3 bench.py --nodes 62500 --no-variants --no-configs --cpu-rows -1 > /dev/null 2>> "$O/bench.err"
ls -la "$O"; grep -v "^BENCH_DETAIL" "$O/bench.err" | tail -c 600
cat "$O/r06_bench_line.json"
