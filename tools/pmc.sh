#!/bin/bash
# Collects rocprofv3 PMC counters for one short bench run, one counter group per pass.
# usage: tools_pmc.sh <outdir> [bench args...]
set -e
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p "$OUT"
ARGS="--steps 1 --warmup 0 --spp-per-step 4 --no-cpu-baseline --no-secondary $@"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/sq" -- python3 bench.py $ARGS > "$OUT/sq.json" 2> "$OUT/sq.err"
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq2" -- python3 bench.py $ARGS > "$OUT/sq2.json" 2> "$OUT/sq2.err"
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d "$OUT/tcc" -- python3 bench.py $ARGS > "$OUT/tcc.json" 2> "$OUT/tcc.err"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 bench.py $ARGS > "$OUT/fetch.json" 2> "$OUT/fetch.err"
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 bench.py $ARGS > "$OUT/write.json" 2> "$OUT/write.err"
echo done
