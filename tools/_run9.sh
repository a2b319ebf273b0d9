for hs in 256 512 256 512; do
NQ_WGRAD3_HEAD_SPLITS=$hs python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-fp32 --no-nerv --no-phase1 --repeats 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('head splits $hs', d['value'], 'it/s', d['ms_per_step'])"
python - <<P
import json
rows=json.load(open('gpurun_out/bench_kernels_n1.json'))
print([ (r['kernel'], r['cin'], r['cout'], round(r['avg_ms']*1e3,1)) for r in rows if (r['cin'],r['cout']) in ((37,3),(3,37))])
P
done
python -m pytest tests/test_hip_parity.py tests/test_full_size.py -q -m gpu -k "wgrad or single_step or layout" 2>&1 | tail -3
