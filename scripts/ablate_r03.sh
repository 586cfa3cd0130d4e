#!/bin/bash
# Per-cause cost table of the headline kernel by ABLATION (profiles/r03_ablation_c3.txt): the library is built with parts of the
# tile loop / tail removed (-DBNN_ABLATE=bits, bnn_forward.hip.h: 1 ReLU, 2 Welford pool, 4 per-tile x loads, 8 everything after the
# tile loop) and configs[2] is timed for each build on ONE box, interleaved.  Results of the ablated builds are wrong by
# construction; only their kernel time is used.  Build the variants first (in the build container):
#   for v in 1 2 4 8 15; do python -m bnn_chaos_model_amd.csrc.build -DBNN_ABLATE=$v -o libabl_$v.so; done
python scripts/ab_variants.py --reps ${1:-2} --steps 3 libbnn_chaos_hip.so libabl_1.so libabl_2.so libabl_4.so libabl_8.so libabl_15.so
