#!/bin/bash
# A/B of the two feature_nn engines (same library, env switch), interleaved
for round in 1 2; do for k in 4x4 16x16; do
  BNN_CHAOS_KERNEL=$k python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('engine $k', '%.4g'%r['value'], '%.2f ms'%r['roofline']['kernel_ms'], '%.3f'%r['roofline']['frac'])"
done; done
