mkdir -p gpurun_out/bench
python bench.py > gpurun_out/bench/bench.json 2> gpurun_out/bench/bench.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/bench/smoke.txt 2>&1
