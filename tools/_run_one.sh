cd $GRAFT_REPO_ROOT
python tools/time_rsym_clustered.py 2>&1 | tail -2
OUTLIERS=1 python tools/time_rsym_clustered.py 2>&1 | tail -2
