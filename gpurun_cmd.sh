mkdir -p gpurun_out/wpe
BTK_WPE_LP_PHASES=1 WPE_S=2 python profiles/wpe_one.py 2>&1 | grep -E "lagprod|wpe_estimate" | tail -6 > gpurun_out/wpe/phases.txt
