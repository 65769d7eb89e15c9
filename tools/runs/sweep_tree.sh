#!/bin/bash
# Host mesh-tree builder parameters (read from the environment by the scene loader) on both scenes, wavefront organisation.
mkdir -p gpurun_out/r3e
run() { echo "== $*"; env "$@" timeout -k 10 200 bash tools/bench_pipe.sh 3 || exit 1; env "$@" timeout -k 10 200 bash tools/bench_pipe.sh 3 --scene outside_synth || exit 1; }
for spec in "$@"; do run $spec; done
