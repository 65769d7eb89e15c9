#!/bin/bash
# Memory-pipeline counters (TA / TCP) for one short bench run.
set -e
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p "$OUT"
ARGS="--steps 1 --warmup 0 --spp-per-step 4 --no-cpu-baseline $@"
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum GRBM_GUI_ACTIVE --output-format csv -d "$OUT/mem1" -- python3 bench.py $ARGS > "$OUT/mem1.json" 2> "$OUT/mem1.err"
timeout -s KILL 120 rocprofv3 --pmc TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum --output-format csv -d "$OUT/mem2" -- python3 bench.py $ARGS > "$OUT/mem2.json" 2> "$OUT/mem2.err"
echo done
