mkdir -p gpurun_out/wpe
timeout 900 python -m pytest tests/test_gpu_wpe.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|Error" > gpurun_out/wpe/test.txt
