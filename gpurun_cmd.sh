mkdir -p gpurun_out/wpe
python bench_configs.py > gpurun_out/wpe/configs.json 2>/dev/null
python bench_stages.py > gpurun_out/wpe/stages.json 2>/dev/null
timeout 900 python -m pytest tests/test_gpu_wpe.py tests/test_gpu_configs.py tests/test_gpu_btk20_api.py tests/test_gpu_cpp_nodes.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|Error" > gpurun_out/wpe/test.txt
