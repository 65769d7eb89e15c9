#!/bin/bash
# Batch-size and hand-over sweep of the default workload: steps of 32/64/128 frames, the wavefront organisation handing over
# to its persistent launch at bounce 3..7 or never, and one run each of the open scene and of the 4K resolution.
mkdir -p gpurun_out/r3a
for t in 0 3 4 5 6 7; do
  echo "== tail $t"
  LPT_WAVEFRONT_TAIL=$t timeout -k 10 200 bash tools/bench_pipe.sh 0 || exit 1
done
for s in 32 128; do
  echo "== spp $s"
  timeout -k 10 200 bash tools/bench_pipe.sh 0 --spp-per-step $s || exit 1
done
echo "== outside"
timeout -k 10 200 bash tools/bench_pipe.sh 3 --scene outside_synth || exit 1
echo "== 4k"
timeout -k 10 200 bash tools/bench_pipe.sh 3 --width 3840 --height 2160 --spp-per-step 32 || exit 1
