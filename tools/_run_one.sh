cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_hip_parity.py -q -m gpu -k "fallback_tiers or full_size_ranked_symmetric or module_matches or edge_cases or degenerate" 2>&1 | tail -3
