#!/bin/bash
# Kernel-trace and PMC passes of one bench workload on the GPU box; summaries land in gpurun_out/<tag>_*.txt.
#   usage: tools/profile_step.sh <tag> <workload: orvit|steve|hr> [trace|pmc|all]
# PMC passes are separate runs (FETCH_SIZE and WRITE_SIZE cannot share a pass; never combined with sys/hip traces).
set -e
TAG=$1; WL=$2; WHAT=${3:-all}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
HEAD=$(cat "$ROOT/focus_amd/lib/BUILD_HEAD" 2>/dev/null || echo unknown)
STEPS=3; WARM=1
ARGS="bench.py --workload $WL --steps $STEPS --warmup $WARM --no-cpu-baseline --no-roofline"
# the STEVE record is traced eagerly: warm-up + timed steps = STEPS + WARM executions of the step, nothing else
[ "$WL" = steve ] && ARGS="$ARGS --steve-eager"
[ "$WL" = hr ] && ARGS="$ARGS --hr-large-batch 0"
cd "$ROOT"
if [ "$WHAT" = trace ] || [ "$WHAT" = all ]; then
  rm -rf /tmp/prof_trace
  rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_trace -- python3 $ARGS > "$OUT/${TAG}_trace_bench.log" 2>&1
  { echo "# rocprofv3 --kernel-trace --output-format csv -- python3 $ARGS   (build $HEAD)";
    python3 profiles/trace_summary.py /tmp/prof_trace $((STEPS + WARM)) 70; } > "$OUT/${TAG}_kernel_summary_${WL}.txt"
fi
if [ "$WHAT" = pmc ] || [ "$WHAT" = all ]; then
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/prof_$C
    rocprofv3 --kernel-trace --pmc $C -d /tmp/prof_$C --output-format csv -- python3 $ARGS > "$OUT/${TAG}_pmc_${C}.log" 2>&1
  done
  python3 tools/pmc_traffic.py /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE "$HEAD" "$WL" $((STEPS + WARM)) > "$OUT/${TAG}_pmc_hbm_traffic_${WL}.txt"
  rm -rf /tmp/prof_sq
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVE_CYCLES -d /tmp/prof_sq --output-format csv \
      -- python3 $ARGS > "$OUT/${TAG}_pmc_sq.log" 2>&1
  { echo "# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVE_CYCLES -- python3 $ARGS   (build $HEAD)";
    python3 tools/pmc_summary.py /tmp/prof_sq; } > "$OUT/${TAG}_pmc_mfma_util_${WL}.txt"
fi
echo "profile_step $TAG $WL $WHAT done"
