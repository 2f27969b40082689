mkdir -p gpurun_out/bench
python bench.py > gpurun_out/bench/bench.json 2> gpurun_out/bench/bench.err
