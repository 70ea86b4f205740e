for M in 28 32; do DGG_SWEEP_M=$M DGG_SWEEP_STATS=1 timeout 300 python tools/time_sweep.py 100000 64 2 2>&1 | tail -1; done
DGG_SWEEP_STATS=1 timeout 300 python tools/time_sweep.py 100000 128 2 2>&1 | tail -1
DGG_SWEEP_STATS=1 timeout 300 python tools/time_sweep.py 30000 32 2 2>&1 | tail -1
DGG_SWEEP_STATS=1 timeout 300 python tools/time_sweep.py 500000 64 2 2>&1 | tail -1
DGG_SWEEP_STATS=1 timeout 300 python tools/time_sweep.py 20000 16 2 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o sweep -- python3 $GRAFT_REPO_ROOT/tools/time_sweep.py 100000 64 2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/kernel_stats.py /tmp/prof/sweep_results.db gpurun_out/r3g/sweep_kernel_stats2.csv --skip-first 2
