#!/bin/bash
# SQ-only counter pass for one short bench run.  usage: pmc_sq.sh <outdir> [bench args]
set -e
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p "$OUT"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary "$@" > "$OUT/sq.json" 2> "$OUT/sq.err"
echo done
