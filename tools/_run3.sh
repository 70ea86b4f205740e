mkdir -p gpurun_out/r3e
cd /tmp && export TMPDIR=/tmp
for v in nohit notest nostage nostagetest; do
  if [ $v = base ]; then unset DGG_HIP_SO; else export DGG_HIP_SO=$GRAFT_REPO_ROOT/tools/_bin/libdgg_$v.so; fi
  rm -rf /tmp/prof_$v
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o sweep -- python3 $GRAFT_REPO_ROOT/tools/time_sweep.py 100000 64 2 > /tmp/prof_$v.log 2>&1
  python3 $GRAFT_REPO_ROOT/tools/kernel_stats.py /tmp/prof_$v/sweep_results.db $GRAFT_REPO_ROOT/gpurun_out/r3e/$v.csv --skip-first 2
  echo "== $v"; grep -E "sw_sweep|sw_finalize" $GRAFT_REPO_ROOT/gpurun_out/r3e/$v.csv | sed 's/_ZN12_GLOBAL__N_1//' | cut -c1-30,95-
done
