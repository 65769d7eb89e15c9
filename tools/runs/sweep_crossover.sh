#!/bin/bash
# Wavefront against hybrid organisation by batch size (frames of 1920x1080 per step), to place the switch of LPT_PIPELINE_AUTO.
for spp in 2 4 6 8 10 12 16 24; do
  for p in 0 4; do
    echo -n "spp $spp: "; timeout -k 10 120 bash tools/bench_pipe.sh $p --spp-per-step $spp || exit 1
  done
done
