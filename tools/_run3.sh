set -x
python -m pytest tests/test_dp_gpu.py tests/test_hip_parity.py tests/test_full_size.py -q -m gpu -k "emulated or interleaved or head_ or single_step or 2_and_4 or decode or trajectory or config1" > gpurun_out/t_r3c.log 2>&1
tail -8 gpurun_out/t_r3c.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_r3b.log 2> gpurun_out/bench_r3b.err; python - <<'P'
import json
d=json.loads(open('gpurun_out/bench_r3b.log').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['repeats'], d['roofline']['kernel'], d['roofline']['frac'], d['fp32']['value'], d['nerv']['value'])
P
