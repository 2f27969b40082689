mkdir -p gpurun_out/cpp
BTK20_BACKEND=cpp timeout 1200 python -m pytest tests/test_gpu_btk20_api.py tests/test_gpu_tools.py -m gpu -q --tb=short 2>&1 | grep -v "^E   *$" > gpurun_out/cpp/test.txt
