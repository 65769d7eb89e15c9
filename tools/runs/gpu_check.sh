#!/bin/bash
# GPU tests named on the command line (default: all), then a parity sweep over fresh random-scene seeds.
mkdir -p gpurun_out/r3d
timeout -k 10 1000 python -u -m pytest ${@:-tests} -m gpu -x -q --timeout 400 --timeout-method thread > gpurun_out/r3d/gputests.txt 2>&1
rc=$?; tail -6 gpurun_out/r3d/gputests.txt; exit $rc
