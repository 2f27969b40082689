mkdir -p gpurun_out/rls
timeout 600 python bench_stages.py > gpurun_out/rls/stages.txt 2>&1
