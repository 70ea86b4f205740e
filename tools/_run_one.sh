cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_hip_parity.py -q -m gpu -k "full_size_ranked_symmetric or ranked_symmetric_search_equals" -s 2>&1 | grep -E "^E  |passed|failed|FAILED|clustered=" | cut -c1-400 | head -20
