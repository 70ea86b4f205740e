set -x
D=gpurun_out/r3n
mkdir -p $D
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_conv_reorder.py -q -m gpu -k "test_allpairs_topk_bit_exact or test_full_size_unperturbed_sweep or test_allpairs_topk_k_limit or row_range or unperturbed" -x 2>&1 | tail -4
for M in 28 32; do DGG_SWEEP_M=$M DGG_SWEEP_STATS=1 timeout 300 python tools/time_sweep.py 100000 64 2 2>&1 | tail -1; done
DGG_SWEEP_STATS=1 timeout 300 python tools/time_sweep.py 100000 128 2 2>&1 | tail -1
DGG_SWEEP_STATS=1 timeout 300 python tools/time_sweep.py 30000 32 2 2>&1 | tail -1
DGG_SWEEP_STATS=1 timeout 300 python tools/time_sweep.py 500000 64 2 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o sweep -- python3 $GRAFT_REPO_ROOT/tools/time_sweep.py 100000 64 2 > $GRAFT_REPO_ROOT/$D/prof.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/$D/pmc_sq -- python3 $GRAFT_REPO_ROOT/tools/time_sweep.py 100000 64 2 > $GRAFT_REPO_ROOT/$D/pmc_sq.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_MFMA --output-format csv -d $GRAFT_REPO_ROOT/$D/pmc_mfma -- python3 $GRAFT_REPO_ROOT/tools/time_sweep.py 100000 64 2 > $GRAFT_REPO_ROOT/$D/pmc_mfma.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/kernel_stats.py /tmp/prof/sweep_results.db $D/sweep_kernel_stats.csv --skip-first 2
python tools/sq_breakdown.py $D/pmc_sq $D/sq.csv; cat $D/sq.csv
python tools/mfma_busy.py $D/pmc_mfma $D/mfma.csv; cat $D/mfma.csv
