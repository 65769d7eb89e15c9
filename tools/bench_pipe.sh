#!/bin/bash
# Short bench of one pipeline (0 wavefront, 1 megakernel, 2 persistent, 3 automatic, 4 hybrid); prints the numbers that matter.
P=${1:-2}; shift
timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --pipeline $P "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin)
print(d['config']['pipeline'], 'Msamples/s %.1f  ms/step %.2f  device_ms %.2f  kernel_ms %.2f (%d launches)' % (d['value'], d['ms_per_step'], d['device_ms_rank0'], d['extend_ms_rank0'], d['extend_launches_rank0']))"
