set -x
python -m pytest tests/test_dp_gpu.py tests/test_hip_parity.py -q -m gpu -x -k "dp_gpu or uaq_multi or arena or captured or deferred or fused_adaround" > gpurun_out/t_r3g.log 2>&1
tail -6 gpurun_out/t_r3g.log
