mkdir -p gpurun_out/syn
python profiles/syn_ab.py > gpurun_out/syn/syn_ab.txt 2>/dev/null
timeout 900 python -m pytest tests/test_gpu_filterbank.py tests/test_gpu_edge_cases.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize_properties.py tests/test_gpu_beamformer.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|Error" > gpurun_out/syn/test.txt
