bash profiles/scripts/r03_run_all.sh > gpurun_out/r03_run_all.log 2>&1
