set -x
python -m pytest tests -q -m gpu -x > gpurun_out/t_r3j.log 2>&1; tail -4 gpurun_out/t_r3j.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r3j.log 2> gpurun_out/bench_r3j.err; tail -c 2500 gpurun_out/bench_r3j.log
