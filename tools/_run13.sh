for v in 0 1 0 1; do
NQ_IG3_NOSPLIT_SHORT=$v python bench.py --workload nerv --steps 40 --warmup 6 --no-cpu-baseline --no-fp32 --no-nerv --no-phase1 --repeats 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nerv keep_whole=$v', d['value'], 'it/s', d['ms_per_step'])"
done
python -m pytest tests/test_hip_parity.py -q -m gpu -k "conv or nerv or trajectory" 2>&1 | tail -2
