#!/bin/bash
# Builds variants of liblpt_hip.so (extra -D flags), then in ONE GPU call: for each variant the Cornell parity
# tests and a short bench of the given pipeline on Cornell and on outside_synth.
#   tools/variant_test.sh <pipeline> name1:"-DX=1" name2:"..." ...
set -e
cd "$(dirname "$0")/.."
P=$1; shift
V=logipathtracer_amd/lib/variants
rm -rf $V; mkdir -p $V
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -Wno-pass-failed -Wno-unused-function -Iinclude -Ilogipathtracer_amd/csrc"
names=()
pids=()
for spec in "$@"; do
  n=${spec%%:*}; f=${spec#*:}
  names+=($n)
  /opt/rocm/bin/hipcc $FLAGS $f -shared -o $V/$n.so logipathtracer_amd/csrc/lpt_hip.hip logipathtracer_amd/csrc/lpt_lbvh.hip 2>$V/$n.err &
  pids+=($!)
  if [ ${#pids[@]} -ge 4 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
for n in "${names[@]}"; do if [ ! -f $V/$n.so ]; then echo "build of $n failed"; cat $V/$n.err | head -20; exit 1; fi; done
cmd="mkdir -p gpurun_out"
for n in "${names[@]}"; do
  cmd="$cmd && cp $V/$n.so logipathtracer_amd/lib/liblpt_hip.so && echo '== $n' && timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_synthetic_scenes.py -m gpu -x -q --timeout 200 --timeout-method thread 2>&1 | tail -1 && echo -n 'cornell: ' && timeout -k 10 120 bash tools/bench_pipe.sh $P && echo -n 'outside: ' && timeout -k 10 120 bash tools/bench_pipe.sh $P --scene outside_synth --spp-per-step 32"
done
/usr/local/graft/bin/gpurun --timeout 1100 -- "$cmd" 2>&1 | grep -vE "^\[gpurun\] (sending|merged)"
