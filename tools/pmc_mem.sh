#!/bin/bash
# Memory-pipeline counters (TA / TD / TCP) for one short bench run.  usage: pmc_mem.sh <outdir> [bench args]
set -e
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p "$OUT"
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-secondary $@"
timeout -s KILL 200 rocprofv3 --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum GRBM_GUI_ACTIVE --output-format csv -d "$OUT/mem1" -- python3 bench.py $ARGS > "$OUT/mem1.json" 2> "$OUT/mem1.err"
timeout -s KILL 200 rocprofv3 --pmc TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum --output-format csv -d "$OUT/mem2" -- python3 bench.py $ARGS > "$OUT/mem2.json" 2> "$OUT/mem2.err"
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_IFETCH --output-format csv -d "$OUT/sq3" -- python3 bench.py $ARGS > "$OUT/sq3.json" 2> "$OUT/sq3.err"
echo done
