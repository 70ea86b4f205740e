cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/ppi
rocprofv3 --kernel-trace --stats -d /tmp/trace_ppi -o h -- python3 $R/bench.py --steps 4 --warmup 2 --workload ppi --bf16 --graphs 20 --cpu-rows -1 > $R/gpurun_out/ppi/bench.json 2> $R/gpurun_out/ppi/err.log
cd $R
python3 tools/kernel_stats.py /tmp/trace_ppi/h_results.db gpurun_out/ppi/ppi_bf16_kernel_stats.csv --skip-first 0 > /dev/null
head -40 gpurun_out/ppi/ppi_bf16_kernel_stats.csv | cut -c1-200
