mkdir -p gpurun_out/syn
timeout 900 python -m pytest tests/test_gpu_fullsize_properties.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|Error|assert" > gpurun_out/syn/test.txt
