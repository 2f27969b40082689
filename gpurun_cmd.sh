mkdir -p gpurun_out/syn
timeout 900 python -m pytest tests/test_gpu_filterbank.py tests/test_gpu_edge_cases.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py tests/test_gpu_fullsize_properties.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|Error" > gpurun_out/syn/test.txt
python profiles/fb_ab.py 2>&1 | grep "M=" > gpurun_out/syn/fb_ab_new.txt
BTK_SYN_NARROW=1 python profiles/fb_ab.py 2>&1 | grep "M=" > gpurun_out/syn/fb_ab_old.txt
python bench_configs.py > gpurun_out/syn/configs.json 2>gpurun_out/syn/configs.err
