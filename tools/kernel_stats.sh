#!/bin/bash
# Per-kernel average durations of the hot loop (rocprofv3 --kernel-trace --stats over bench.py, graph replays on):
#   bash tools/kernel_stats.sh [substring ...]     -> name, calls, average microseconds (all kernels when no substring given)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/kstats
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --frames 8 --no-cpu-baseline --no-fp32 --no-trained --no-uvg --steps 40 --warmup 6 > $OUT/bench.log 2>&1
cd $R
python3 - "$@" <<'PY'
import csv, glob, sys
pats = sys.argv[1:]
rows = list(csv.DictReader(open(glob.glob("gpurun_out/kstats/*/*_kernel_stats.csv")[0])))
for r in rows:
    if not pats or any(p in r["Name"] for p in pats):
        print(f'{r["Name"][:90]:90s} {int(r["Calls"]):5d} {float(r["AverageNs"]) / 1e3:9.1f} us')
PY
grep '^{' $OUT/bench.log | cut -c88-112
rm -rf $OUT/*/
