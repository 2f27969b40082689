#!/bin/bash
# What clock and power does the fused kernel run at?  Runs profiles/fused_ab.py in a loop in the background and samples
# rocm-smi while it runs (one variant per BTK_FUSED_VAR in $VARS).
mkdir -p gpurun_out
for v in ${VARS:-15}; do
  ( for i in 1 2 3 4; do BTK_FUSED_VAR=$v python profiles/fused_ab.py 2>/dev/null | tail -1; done ) > gpurun_out/clock_probe_$v.log &
  pid=$!
  sleep 6
  echo "== BTK_FUSED_VAR=$v"
  for i in 1 2 3 4 5 6; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power|power" | tr -s ' ' | tr '\n' ';'; echo
    sleep 0.7
  done
  wait $pid
  tail -1 gpurun_out/clock_probe_$v.log
done
rocm-smi --showmaxpower 2>/dev/null | grep -i power
