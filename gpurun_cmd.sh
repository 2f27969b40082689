mkdir -p gpurun_out/wpe
cd profiles/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/mfma_war mfma_war.hip && /tmp/mfma_war > ../../gpurun_out/wpe/mfma_war.txt 2>&1
