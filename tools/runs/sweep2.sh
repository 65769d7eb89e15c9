#!/bin/bash
# Builds variants of liblpt_hip.so (extra -D flags) on the CPU here, then in ONE GPU call benches each on the Cornell
# box and on the open scene (wavefront organisation).   tools/runs/sweep2.sh name1:"-DX=1" name2:"..." ...
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
cmd="cp logipathtracer_amd/lib/liblpt_hip.so /tmp/keep.so"
for n in "${names[@]}"; do
  cmd="$cmd && cp $V/$n.so logipathtracer_amd/lib/liblpt_hip.so && echo -n '$n cornell: ' && timeout -k 10 120 bash tools/bench_pipe.sh ${PIPE:-0} $ARGS_A && echo -n '$n outside: ' && timeout -k 10 120 bash tools/bench_pipe.sh ${PIPE:-3} --scene outside_synth $ARGS_B"
done
/usr/local/graft/bin/gpurun --timeout 1100 -- "$cmd" 2>&1 | grep -E "cornell: |outside: |status=|rror"
