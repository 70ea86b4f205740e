# round-3 profile collection (run on the GPU box): bash tools/prof_r03.sh  -> gpurun_out/r03/
set -eu
R="${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
O="$R/gpurun_out/r03"
rm -rf -- "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
BA="--steps 3 --warmup 1 --repeats 1 --cpu-rows -1 --no-hipgraph --no-variants"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch -- python3 $R/bench.py $BA > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write -- python3 $R/bench.py $BA > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_mfma -- python3 $R/bench.py $BA > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_sq -- python3 $R/bench.py $BA > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/trace -o h -- python3 $R/bench.py --steps 20 --warmup 5 --repeats 3 --cpu-rows -1 --no-variants > $O/bench_under_profiler.json 2>/dev/null
for nz in none sym rsym hash; do
  rocprofv3 --kernel-trace --stats -d /tmp/trace_$nz -o h -- python3 $R/bench.py --noise $nz --steps 10 --warmup 3 --repeats 2 --cpu-rows -1 --no-variants > /dev/null 2>&1
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_sq_none -- python3 $R/bench.py --noise none $BA > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_mfma_none -- python3 $R/bench.py --noise none $BA > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_sq_sym -- python3 $R/bench.py --noise rsym $BA > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/trace_ppi -o h -- python3 $R/bench.py --steps 4 --warmup 2 --workload ppi --bf16 --cpu-rows -1 > /dev/null 2>&1
cd $R
python3 tools/pmc_traffic.py /tmp/pmc_fetch /tmp/pmc_write $O/r03_traffic.json 100000 128 64 > /dev/null
python3 tools/mfma_busy.py /tmp/pmc_mfma $O/r03_mfma_busy.csv > /dev/null
python3 tools/sq_breakdown.py /tmp/pmc_sq $O/r03_sq_breakdown.csv > /dev/null
python3 tools/kernel_stats.py /tmp/trace/h_results.db $O/r03_kernel_stats.csv --skip-first 8 > /dev/null
python3 tools/kernel_stats.py /tmp/trace_none/h_results.db $O/r03_unperturbed_kernel_stats.csv --skip-first 3 > /dev/null
python3 tools/kernel_stats.py /tmp/trace_rsym/h_results.db $O/r03_symmetric_kernel_stats.csv --skip-first 3 > /dev/null
python3 tools/kernel_stats.py /tmp/trace_sym/h_results.db $O/r03_hash_symmetric_kernel_stats.csv --skip-first 3 > /dev/null
python3 tools/kernel_stats.py /tmp/trace_ppi/h_results.db $O/r03_ppi_bf16_kernel_stats.csv --skip-first 0 > /dev/null
python3 tools/kernel_stats.py /tmp/trace_hash/h_results.db $O/r03_hash_kernel_stats.csv --skip-first 3 > /dev/null
python3 tools/sq_breakdown.py /tmp/pmc_sq_none $O/r03_unperturbed_sq_breakdown.csv > /dev/null
python3 tools/mfma_busy.py /tmp/pmc_mfma_none $O/r03_unperturbed_mfma_busy.csv > /dev/null
python3 tools/sq_breakdown.py /tmp/pmc_sq_sym $O/r03_symmetric_sq_breakdown.csv > /dev/null
cp $O/r03_traffic.json profiles/r03_traffic.json     # bench.py reads the newest traffic file from here
python3 bench.py --steps 20 --warmup 5 > $O/r03_bench.json 2> $O/bench.err
python3 bench.py --steps 20 --warmup 5 --workload pubmed > $O/r03_pubmed_uvdist_bench.json 2>> $O/bench.err
python3 bench.py --steps 20 --warmup 5 --workload pubmed --edge-mode u-v-deg > $O/r03_pubmed_uvdeg_bench.json 2>> $O/bench.err
python3 bench.py --steps 10 --warmup 3 --workload ppi > $O/r03_ppi_bench.json 2>> $O/bench.err
python3 bench.py --steps 10 --warmup 3 --workload ppi --bf16 > $O/r03_ppi_bf16_bench.json 2>> $O/bench.err
python3 bench.py --steps 20 --warmup 5 --nodes 500000 --no-variants --cpu-rows -1 > $O/r03_bench_n500k_1gpu.json 2>> $O/bench.err
python3 bench.py --steps 20 --warmup 5 --emulate-world 8 --nodes 62500 --no-variants --cpu-rows -1 > $O/r03_emulated_rank_of_8_strong_500k.json 2>> $O/bench.err
python3 bench.py --steps 20 --warmup 5 --emulate-world 8 --no-variants --cpu-rows -1 > $O/r03_emulated_rank_of_8_weak_800k.json 2>> $O/bench.err
ls -la $O; tail -c 400 $O/bench.err
python3 - <<'PY'
import json
j=json.load(open('gpurun_out/r03/r03_bench.json'))
print(j['ms_per_step'], j['roofline']['frac'], {k:(v.get('ms_per_step'), v.get('pair_kernel_ms'), (v.get('roofline') or {}).get('frac')) for k,v in j['variants'].items()})
PY
