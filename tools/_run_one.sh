cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_hip_conv_reorder.py -q -m gpu -k "variant_layer_without" 2>&1 | grep -E "^E  |passed|failed|FAILED|Error" | cut -c1-300 | head
