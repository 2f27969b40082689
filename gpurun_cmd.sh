mkdir -p gpurun_out/cpp
timeout 1500 python -m pytest tests/ -m gpu -q --tb=short 2>&1 | grep -v "^E   *$" | tail -50 > gpurun_out/cpp/test.txt
