timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "literal_dgg_hard or dgg_hard_is_straight" -x 2>&1 | tail -25
