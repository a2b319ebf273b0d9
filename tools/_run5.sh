set -x
python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "conv or decode or trajectory or layout" > gpurun_out/t_r3e.log 2>&1
tail -4 gpurun_out/t_r3e.log
for lib in tools/_ab/libnqhip_prev.so "" tools/_ab/libnqhip_prev.so ""; do
  NQ_LIB=$lib python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-fp32 --no-nerv --no-phase1 --repeats 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=[$lib]', d['value'], d['ms_per_step'])"
  python - <<P
import json
rows=json.load(open('gpurun_out/bench_kernels_n1.json'))
print([ (r['kernel'], r['cin'], r['cout'], round(r['avg_ms']*1e3,1)) for r in rows[:8]])
P
done
