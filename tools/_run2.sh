set -x
python tools/bench_head2.py > gpurun_out/head2.log 2>&1; tail -5 gpurun_out/head2.log
python -m pytest tests/test_dp_gpu.py tests/test_hip_parity.py tests/test_full_size.py -q -m gpu -k "emulated or interleaved or head_forward_streaming or single_step or 2_and_4" > gpurun_out/t_r3b.log 2>&1
tail -8 gpurun_out/t_r3b.log
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_a -- python3 $R/bench.py --steps 12 --warmup 5 --frames 8 --no-cpu-baseline --no-fp32 --no-nerv --no-phase1 --repeats 1 > $R/gpurun_out/trace_a.log 2>&1
cd $R
head -2 gpurun_out/trace_a/*/*_kernel_trace.csv | cut -c1-600
python3 tools/trace_step.py gpurun_out/trace_a gpurun_out/r03_a_step_sequence.txt > /dev/null; tail -3 gpurun_out/r03_a_step_sequence.txt
rm -rf gpurun_out/trace_a
