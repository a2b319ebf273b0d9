#!/usr/bin/env python3
"""Print the per-kernel table bench.py leaves in gpurun_out/bench_kernels_n1.json."""
import json, os, sys
p = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "bench_kernels_n1.json")
rows = json.load(open(p))
tot = sum(r["total_ms"] for r in rows)
steps = max(r["launches"] for r in rows)
for r in rows:
    print(f'{r["kernel"]:12s} k{r["k"]} {r["cin"]:4d}->{r["cout"]:4d} {r["H"]:4d}x{r["W"]:<4d} n={r["launches"]:3d} avg {r["avg_ms"]:.3f} ms  {r["tflops"]:6.1f} TF')
print(f"conv total per profiled step (/{steps}): {tot/steps:.3f} ms")
