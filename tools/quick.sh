#!/bin/bash
# Parity tests that exercise the kernels, then a short bench; prints the numbers that matter.
set -e
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_components.py -m gpu -x -q --timeout 200 --timeout-method thread -k "oracle or golden or flight or hit_for_hit" 2>&1 | tail -3
timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin)
print('Msamples/s %.1f  ms/step %.2f  device_ms %.2f  extend_ms %.2f (%d launches, avg %.3f ms)' % (d['value'], d['ms_per_step'], d['device_ms_rank0'], d['extend_ms_rank0'], d['extend_launches_rank0'], d['extend_ms_rank0']/max(1,d['extend_launches_rank0'])))"
