#!/bin/bash
# A/B of the fused analysis+beamform kernel forms (BTK_FUSED_VAR is read once per process): ROUNDS alternating passes over the
# variants, best time per variant (clocks differ between boxes and drift within a call; only in-call comparisons count).
#   3 = LDS-DMA staging of the PCM span, 7 = polyphase window straight from HBM, 15 = 7 + window loads interleaved with the LDS traffic of the FFT
mkdir -p gpurun_out
: > gpurun_out/r03_fused_ab.log
for r in $(seq ${ROUNDS:-3}); do
  for v in ${VARS:-7 3}; do
    echo "var $v $(BTK_FUSED_VAR=$v python profiles/fused_ab.py 2>/dev/null | tail -1)" >> gpurun_out/r03_fused_ab.log
  done
done
python - <<'PY'
import json, collections
best = collections.defaultdict(lambda: (1e9, None))
for line in open("gpurun_out/r03_fused_ab.log"):
    _, v, js = line.split(" ", 2)
    d = json.loads(js)
    if d["ms"] < best[v][0]:
        best[v] = (d["ms"], d["rel_err_vs_staged"])
for v, (ms, err) in best.items():
    print("BTK_FUSED_VAR=%s best %.4f ms  (%.1f M frames/s, rel err vs staged %.2g)" % (v, ms, 65536 / ms / 1e3, err))
PY
