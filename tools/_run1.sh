set -x
python tools/precision_gate.py --frames bunny_real --ckpt tests/golden/hnerv3m_bunny8real_f16.npz --iters 21000 --seeds 905 --oracle-iters 0 --no-record --out gpurun_out/r03_gate21k_b.json > gpurun_out/r03_gate21k_b.log 2>&1
grep "seed 90" gpurun_out/r03_gate21k_b.log
python -m pytest tests/test_dp_gpu.py tests/test_hip_parity.py tests/test_full_size.py -x -q -m gpu -k "dp_gpu or interleaved or fused_loss or arena or single_step or 2_and_4" > gpurun_out/t_r3a.log 2>&1
tail -5 gpurun_out/t_r3a.log
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_a -- python3 $R/bench.py --steps 12 --warmup 5 --frames 8 --no-cpu-baseline --no-fp32 --no-nerv --no-phase1 --repeats 1 > $R/gpurun_out/trace_a.log 2>&1
cd $R
python3 tools/trace_step.py gpurun_out/trace_a gpurun_out/r03_a_step_sequence.txt > /dev/null; tail -3 gpurun_out/r03_a_step_sequence.txt
rm -rf gpurun_out/trace_a
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r3a.log 2> gpurun_out/bench_r3a.err; tail -c 3000 gpurun_out/bench_r3a.log
