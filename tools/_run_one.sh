cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_hip_parity.py -q -m gpu -k "fallback_tiers" 2>&1 | grep -E "^E  |passed|failed|FAILED" | cut -c1-300 | head
