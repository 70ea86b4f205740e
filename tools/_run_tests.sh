mkdir -p gpurun_out/tests
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/tests/gpu_tests.txt
cat gpurun_out/tests/gpu_tests.txt
