for cfg in "4 4" "5 4" "4 2" "5 2"; do
set -- $cfg
NQ_HEAD_RD=$1 NQ_HEAD_DG_R=$2 python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-fp32 --no-nerv --no-phase1 --repeats 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fwd R=$1 dgrad R=$2', d['value'], 'it/s', d['ms_per_step'])"
python - <<P
import json
rows=json.load(open('gpurun_out/bench_kernels_n1.json'))
print([ (r['kernel'], r['cin'], r['cout'], round(r['avg_ms']*1e3,1)) for r in rows if (r['cin'],r['cout']) in ((37,3),(3,37))])
P
done
