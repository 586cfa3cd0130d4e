#!/bin/bash
# bench.py in each launch mode (interleaved), optional extra args
for round in 1 2; do for mode in "" "--single-launch" "--unfused"; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline $mode "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('mode[$mode]', '%.4g'%r['value'], '%.2f ms'%r['roofline']['kernel_ms'], '%.3f'%r['roofline']['frac'])"
done; done
