#!/bin/bash
# How much of the matrix-pipe time do two co-resident waves per SIMD recover over one?  Runs bench.py on an experiment build
# (-DBNN_EXP=4: BNN_EXP_LDS_PAD adds dynamic LDS so that only one workgroup = one wave per SIMD fits on a CU).
for pad in 0 40000; do
  BNN_EXP_LDS_PAD=$pad BNN_CHAOS_SO=$PWD/bnn_chaos_model_amd/csrc/libexp4.so python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null |
    python -c "import sys,json; r=json.loads(sys.stdin.read()); print('pad $pad', '%.4g'%r['value'], '%.2f ms'%r['roofline']['kernel_ms'], '%.3f'%r['roofline']['frac'])"
done
