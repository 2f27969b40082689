mkdir -p gpurun_out/wpe
timeout 900 python -m pytest tests/test_gpu_wpe.py tests/test_gpu_configs.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|Error" > gpurun_out/wpe/test.txt
WPE_S=2 bash profiles/scripts/r02_wpe_profile.sh > gpurun_out/wpe/profile.txt 2>&1
