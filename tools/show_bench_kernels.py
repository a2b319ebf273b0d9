#!/usr/bin/env python3
"""Print the per-kernel table bench.py leaves in gpurun_out/bench_kernels_n1.json."""
import json, sys
rows = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/bench_kernels_n1.json"))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print(f"{r['kernel']:12s} k{r['k']} {r['cin']:4d}->{r['cout']:4d} {r['H']}x{r['W']}  avg {r['avg_ms'] * 1e3:7.1f} us  {r['tflops']:6.1f} TF")
