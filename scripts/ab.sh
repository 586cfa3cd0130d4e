#!/bin/bash
# A/B: bench.py (unfused) for each .so given as argument, interleaved rounds in separate processes
for round in 1 2; do for so in "$@"; do
  BNN_CHAOS_SO=$PWD/bnn_chaos_model_amd/csrc/$so python bench.py --steps 3 --warmup 1 --no-cpu-baseline --unfused 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$so', '%.4g'%r['value'], '%.2f ms'%r['roofline']['kernel_ms'], '%.3f'%r['roofline']['frac'])"
done; done
