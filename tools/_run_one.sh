cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_hip_parity.py -q -m gpu -k "module_symmetric_noise" 2>&1 | grep -E "^E  |passed|failed|FAILED|Error" | cut -c1-300 | head
