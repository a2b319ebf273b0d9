set -x
R=/root/repo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_p1 -- python3 $R/bench.py --steps 12 --warmup 5 --frames 8 --no-cpu-baseline --no-fp32 --no-nerv --repeats 1 > $R/gpurun_out/trace_p1.log 2>&1
cd $R
python3 tools/trace_step.py gpurun_out/trace_p1 gpurun_out/r03_p1_step_sequence.txt > /dev/null
tail -50 gpurun_out/r03_p1_step_sequence.txt
rm -rf gpurun_out/trace_p1
