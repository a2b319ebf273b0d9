#!/bin/bash
# Collect the per-round rocprofv3 evidence on the GPU box:  bash tools/profile_round.sh r02_c
# One --kernel-trace --stats pass and three SEPARATE --pmc passes (never combined with a trace domain), then the
# summaries go to profiles/<tag>_* via profiles/summarize.py.  Output is kept under gpurun_out/ (merged back by gpurun).
set -e
TAG=${1:-rXX}
WL=${2:-hnerv}          # hnerv (the headline workload) | nerv (BASELINE configs[2]: NeRV-3M + Hadamard) | uvg12m (configs[4] shape)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
MODE=${3:-full}         # full | trace (only the per-iteration kernel sequence: one short run)
PREC=${4:-bf16x3}       # bf16x3 | fp32 (exact fp32-input MFMA convolutions)
BENCH="python3 $R/bench.py --frames 8 --workload $WL --precision $PREC --no-cpu-baseline --no-fp32 --no-nerv --no-phase1 --no-trained --no-uvg --repeats 1"
if [ "$MODE" = "trace" ]; then
  rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- $BENCH --steps 12 --warmup 5 > $OUT/trace.log 2>&1
  cd $R
  python3 tools/trace_step.py $OUT/trace $OUT/${TAG}_step_sequence.txt
  grep '^{' $OUT/trace.log | tail -1 | cut -c1-200
  rm -rf $OUT/trace
  exit 0
fi
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH --steps 40 --warmup 6 > $OUT/stats.log 2>&1
export NQ_GRAPH=0
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/fetch -- $BENCH --steps 3 --warmup 1 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $BENCH --steps 3 --warmup 1 > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT/sq -- $BENCH --steps 3 --warmup 1 > $OUT/sq.log 2>&1
unset NQ_GRAPH
cd $R
python3 profiles/summarize.py $OUT/stats $OUT/fetch $OUT/write $OUT/sq $OUT/$TAG > $OUT/summary.log 2>&1
cp $(ls $OUT/stats/*/*_kernel_stats.csv | head -1) $OUT/${TAG}_rocprof_kernel_stats_raw.csv
grep '^{' $OUT/stats.log | tail -1
cat $OUT/summary.log
# one steady-state iteration kernel by kernel (graph replay), from a trace of its own
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- $BENCH --steps 12 --warmup 5 > $OUT/trace.log 2>&1
cd $R
python3 tools/trace_step.py $OUT/trace $OUT/${TAG}_step_sequence.txt > /dev/null 2>&1 || true
# keep the merge-back small: the raw traces are not needed once summarised
rm -rf $OUT/stats $OUT/fetch $OUT/write $OUT/sq $OUT/trace
