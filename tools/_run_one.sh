cd $GRAFT_REPO_ROOT
python tools/prof_ppi_copies.py 2>&1 | tail -25
