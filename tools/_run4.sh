set -x
python -m pytest tests -q -m gpu -x > gpurun_out/t_r3d.log 2>&1
tail -6 gpurun_out/t_r3d.log
for cfg in "1 1 0 0" "0 0 1 1"; do
  set -- $cfg
  NQ_HEAD_FWD=$1 NQ_HEAD_DGRAD=$2 NQ_DEFER_REDUCE=$3 NQ_FUSED_ADAM=$4 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-fp32 --no-nerv --repeats 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['value'], d['ms_per_step'], d['repeats']['values'])"
done
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_b -- python3 $R/bench.py --steps 12 --warmup 5 --frames 8 --no-cpu-baseline --no-fp32 --no-nerv --no-phase1 --repeats 1 > $R/gpurun_out/trace_b.log 2>&1
cd $R
python3 tools/trace_step.py gpurun_out/trace_b gpurun_out/r03_b_step_sequence.txt > /dev/null; tail -3 gpurun_out/r03_b_step_sequence.txt
rm -rf gpurun_out/trace_b
