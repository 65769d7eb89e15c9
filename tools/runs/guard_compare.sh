#!/bin/bash
# Guarded 16-entry search kernel against the instance that holds the scene's stack bound, per BVH builder (0 host SAH with
# spatial splits, 1 host SAH, 2 GPU LBVH, 3 GPU PLOC) on the open scene.
for b in 0 2 3; do
  for exact in 0 1; do
    echo -n "builder $b exact $exact: "
    LPT_EXTEND_EXACT_STACK=$exact timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --scene outside_synth --builder $b 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('%.1f Msamples/s  %.2f ms/step  k_extend %.2f ms x %d  stack %s' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['launches'], {k:v for k,v in r['traversal_stack'].items() if k!='note'}))" || exit 1
  done
done
