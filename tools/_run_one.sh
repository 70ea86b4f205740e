timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "wider_than_the_ell or module_matches or cora or ell_width" -x 2>&1 | tail -15
