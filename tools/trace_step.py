#!/usr/bin/env python3
"""One steady-state calibration iteration as the GPU ran it: kernel by kernel with duration and the idle gap in front of it.

    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 bench.py --steps 12 --warmup 4 --frames 8 --no-cpu-baseline --no-fp32
    python3 tools/trace_step.py <dir> [out.txt]

An iteration starts at `step_prologue_kernel` (captured iterations, quantization/calib_model.py); the tool prints the LAST
complete one of the trace (a graph replay) and the totals by kernel family.
"""
import csv
import glob
import re
import sys
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "..", "profiles"))
from summarize import kname  # noqa: E402


def main(d, out=None):
    f = glob.glob(d + "/*/*_kernel_trace.csv")[0]
    def grid(r):
        if r.get("Grid_Size"):
            return r["Grid_Size"]
        g = 1
        for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"):
            g *= int(r.get(k) or 1)
        return str(g)

    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kname(r["Kernel_Name"]), grid(r), r.get("VGPR_Count", ""),
             r.get("LDS_Block_Size", "")) for r in csv.DictReader(open(f))]
    rows.sort()
    starts = [i for i, r in enumerate(rows) if r[2].startswith("step_prologue")]
    assert len(starts) >= 3, "no captured iterations in the trace"
    a, b = starts[-3], starts[-2]
    it = rows[a:b]
    lines = [f"# one iteration = {len(it)} kernels, {(rows[b][0] - rows[a][0]) / 1e3:.1f} us from prologue to prologue",
             f"# {'kernel':60s} {'grid':>9s} {'vgpr':>5s} {'lds':>7s} {'dur_us':>8s} {'gap_us':>7s}"]
    prev_end, busy, small, nsmall = None, 0, 0, 0
    for s, e, n, g, v, l in it:
        gap = 0 if prev_end is None else (s - prev_end) / 1e3
        lines.append(f"{n:62s} {g:>9s} {v:>5s} {l:>7s} {(e - s) / 1e3:8.1f} {gap:7.1f}")
        prev_end = e
        busy += e - s
        if e - s < 30000:
            small += e - s
            nsmall += 1
    lines.append(f"# kernel time {busy / 1e3:.1f} us; {nsmall} launches below 30 us: {small / 1e3:.1f} us")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(*sys.argv[1:3])
