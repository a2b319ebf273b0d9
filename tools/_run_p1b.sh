NQ_BENCH_STEPLOG=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32 --no-nerv --repeats 1 2>&1 | grep steplog
