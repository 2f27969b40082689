mkdir -p gpurun_out/nlms
for a in 0 7 8; do
  BTK_NLMS_ALT=$a python bench_stages.py 2>/dev/null | python -c "
import sys, json
t=sys.stdin.read(); d=json.loads(t[t.index('{'):])
print('alt=$a', {k:(round(v['ms'],3), round(v['hbm_frac'],3)) for k,v in d.items() if k.startswith('nlms')})"
done > gpurun_out/nlms/alt.txt 2>&1
for a in 0 7 8; do BTK_NLMS_ALT=$a python bench.py --no-cpu --steps 50 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('alt=$a bench nlms', d['stages']['adaptive_nlms_canceller']['ms'])"; done >> gpurun_out/nlms/alt.txt 2>&1
