NQ_IG3_NW=8 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "conv or decode or trajectory" 2>&1 | tail -3
for nw in 4 8 4 8; do
NQ_IG3_NW=$nw python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-fp32 --no-nerv --no-phase1 --repeats 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NW=$nw', d['value'], 'it/s', d['ms_per_step'])"
python - <<P
import json
rows=json.load(open('gpurun_out/bench_kernels_n1.json'))
print([ (r['cin'], r['cout'], round(r['avg_ms']*1e3,1)) for r in rows if r['kernel']=='conv_igemm3'][:8])
P
done
for nw in 4 8; do
NQ_IG3_NW=$nw python bench.py --workload nerv --steps 40 --warmup 6 --no-cpu-baseline --no-fp32 --no-nerv --no-phase1 --repeats 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nerv NW=$nw', d['value'], 'it/s', d['ms_per_step'])"
done
