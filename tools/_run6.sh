set -x
python -m pytest tests -q -m gpu -x > gpurun_out/t_r3f.log 2>&1
tail -5 gpurun_out/t_r3f.log
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_n -- python3 $R/bench.py --workload nerv --steps 12 --warmup 5 --frames 8 --no-cpu-baseline --no-fp32 --no-nerv --no-phase1 --repeats 1 > $R/gpurun_out/trace_n.log 2>&1
cd $R
python3 tools/trace_step.py gpurun_out/trace_n gpurun_out/r03_nerv_a_step_sequence.txt > /dev/null; tail -3 gpurun_out/r03_nerv_a_step_sequence.txt
rm -rf gpurun_out/trace_n
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_r3c.log 2> gpurun_out/bench_r3c.err; python - <<'P'
import json
d=json.loads(open('gpurun_out/bench_r3c.log').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['repeats']['values'], d['roofline']['kernel'], d['roofline']['frac'], 'fp32', d['fp32']['value'], 'nerv', d['nerv']['value'], 'p1', d['phase1'])
P
