export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 NQ_DP_REHEARSAL=1
for wl in hnerv nerv; do for g in 0 1; do
NQ_DP_GRAPH=$g python bench.py --workload $wl --steps 40 --warmup 6 --no-cpu-baseline --no-fp32 --no-nerv --no-phase1 --repeats 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dp rehearsal $wl NQ_DP_GRAPH=$g', d['value'], 'it/s', d['ms_per_step'], 'ms; host enqueue', d['roofline']['host_enqueue_ms_per_step'])"
done; done
unset RANK WORLD_SIZE LOCAL_RANK NQ_DP_REHEARSAL
for wl in hnerv nerv; do
python bench.py --workload $wl --steps 40 --warmup 6 --no-cpu-baseline --no-fp32 --no-nerv --no-phase1 --repeats 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('single $wl', d['value'], 'it/s', d['ms_per_step'], 'ms; host enqueue', d['roofline']['host_enqueue_ms_per_step'])"
done
