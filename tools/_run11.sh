for st in 0 30000 60000 100000 0 60000; do
NQ_IG3_STAGGER=$st python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-fp32 --no-nerv --no-phase1 --repeats 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stagger $st', d['value'], 'it/s', d['ms_per_step'])"
python - <<P
import json
rows=json.load(open('gpurun_out/bench_kernels_n1.json'))
print([ (r['cin'], r['cout'], round(r['avg_ms']*1e3,1)) for r in rows if r['kernel']=='conv_igemm3'][:6])
P
done
