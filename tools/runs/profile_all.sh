#!/bin/bash
# Evidence in one GPU call: kernel-trace statistics and PMC passes of the default workload and of the open
# scene, the weak/strong scaling prediction from per-rank shares, and a bench line with the secondary runs.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3b
bash tools/profile_round.sh r03 > gpurun_out/r3b/profile_r03.log 2>&1 || { echo "profile r03 failed"; tail -5 gpurun_out/r3b/profile_r03.log; exit 1; }
echo "profile r03 done"
bash tools/profile_round.sh r03_outside --scene outside_synth > gpurun_out/r3b/profile_r03_outside.log 2>&1 || { echo "profile outside failed"; tail -5 gpurun_out/r3b/profile_r03_outside.log; exit 1; }
echo "profile r03_outside done"
timeout -k 10 600 python tools/simulate_ranks.py --out gpurun_out/r3b/scaling_prediction.json > gpurun_out/r3b/simulate.log 2>&1 || { echo "simulate failed"; tail -5 gpurun_out/r3b/simulate.log; exit 1; }
echo "simulate done"
