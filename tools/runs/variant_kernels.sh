#!/bin/bash
# Builds variants of liblpt_hip.so (extra -D flags) on the CPU here, then in ONE GPU call runs a kernel trace of a
# short bench of each and prints the average duration of the main kernels.
#   SCENE_ARGS="--scene outside_synth" tools/runs/variant_kernels.sh name1:"-DX=1" name2:"..." ...
set -e
cd "$(dirname "$0")/../.."
V=logipathtracer_amd/lib/variants
rm -rf $V; mkdir -p $V
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -Wno-pass-failed -Wno-unused-function -Iinclude -Ilogipathtracer_amd/csrc"
names=(); pids=()
for spec in "$@"; do
  n=${spec%%:*}; f=${spec#*:}
  names+=($n)
  /opt/rocm/bin/hipcc $FLAGS $f -shared -o $V/$n.so logipathtracer_amd/csrc/lpt_hip.hip logipathtracer_amd/csrc/lpt_lbvh.hip 2>$V/$n.err &
  pids+=($!)
  if [ ${#pids[@]} -ge 4 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
for n in "${names[@]}"; do if [ ! -f $V/$n.so ]; then echo "build of $n failed"; head -20 $V/$n.err; exit 1; fi; done
cmd="export TMPDIR=/tmp && mkdir -p gpurun_out/vk"
for n in "${names[@]}"; do
  cmd="$cmd && cp $V/$n.so logipathtracer_amd/lib/liblpt_hip.so && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/vk/$n -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary $SCENE_ARGS > gpurun_out/vk/$n.json 2> gpurun_out/vk/$n.err; rc=\$?; [ \$rc -eq 0 -o \$rc -eq 3 ] && echo '== $n' && python3 tools/runs/kernel_avgs.py gpurun_out/vk/$n gpurun_out/vk/$n.json"
done
rm -rf gpurun_out/vk
/usr/local/graft/bin/gpurun --timeout 1100 -- "$cmd" 2>&1 | grep -vE "^\[gpurun\] sending"
