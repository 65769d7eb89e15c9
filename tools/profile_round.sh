#!/bin/bash
# Produces the rocprofv3 evidence for one round: kernel-trace statistics of the default bench
# command and separate PMC passes (HBM bytes, SQ activity, L2 hits) of a shortened one.
# usage: tools/profile_round.sh <tag> [bench args]   (run on the GPU box; writes gpurun_out/profile_<tag>/)
set -e
export TMPDIR=/tmp
TAG=${1:-r01}
shift || true
EXTRA="$@"
OUT=gpurun_out/profile_$TAG
mkdir -p "$OUT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 3 --warmup 1 --no-secondary $EXTRA > "$OUT/bench_traced.json" 2> "$OUT/trace.err"
SHORT="--steps 1 --warmup 0 --no-cpu-baseline --no-secondary $EXTRA"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 bench.py $SHORT > "$OUT/fetch.json" 2> "$OUT/fetch.err"
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 bench.py $SHORT > "$OUT/write.json" 2> "$OUT/write.err"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq" -- python3 bench.py $SHORT > "$OUT/sq.json" 2> "$OUT/sq.err"
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d "$OUT/tcc" -- python3 bench.py $SHORT > "$OUT/tcc.json" 2> "$OUT/tcc.err"
# L1 gather path (texture addresser / data return) next to vector issue
timeout -s KILL 200 rocprofv3 --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d "$OUT/l1a" -- python3 bench.py $SHORT > "$OUT/l1a.json" 2> "$OUT/l1a.err"
timeout -s KILL 200 rocprofv3 --pmc TD_TD_BUSY_sum TD_TC_STALL_sum SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d "$OUT/l1b" -- python3 bench.py $SHORT > "$OUT/l1b.json" 2> "$OUT/l1b.err"
echo profile done
