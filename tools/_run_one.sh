timeout 900 python -m pytest tests/test_hip_parity.py tests/test_parallel_gpu.py -q -m gpu -k "rank_of_eight or two_ranks or rccl" -x 2>&1 | tail -15
