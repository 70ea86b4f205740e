cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu -k "unperturbed" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/tr -o h -- python3 $GRAFT_REPO_ROOT/tools/time_sweep.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python3 tools/kernel_stats.py /tmp/tr/h_results.db /tmp/ks.csv --skip-first 0 > /dev/null; head -12 /tmp/ks.csv | cut -c1-110
