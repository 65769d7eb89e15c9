#!/bin/bash
# Build variants of liblpt_hip.so with extra -D flags (here, on the CPU), then bench them all in ONE GPU call.
#   tools/sweep_variants.sh <pipeline> name1:"-DX=1 -DY=2" name2:"..." ...
# The variants live in logipathtracer_amd/lib/variants/ (git-ignored with the rest of lib/).
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
cmd="set -e; cp logipathtracer_amd/lib/liblpt_hip.so /tmp/keep.so"
for n in "${names[@]}"; do
  cmd="$cmd; cp $V/$n.so logipathtracer_amd/lib/liblpt_hip.so; echo -n '$n: '; timeout -k 10 120 bash tools/bench_pipe.sh $P $SWEEP_BENCH_ARGS"
done
/usr/local/graft/bin/gpurun --timeout 900 -- "$cmd" 2>&1 | grep -E "^[A-Za-z0-9_]+: |status=|error|Error"
