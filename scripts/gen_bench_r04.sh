#!/bin/bash
# Generic-engine bench lines (round 4): the pretrained network forced onto the generic engine next to its own kernels, and four other
# hparams-built networks.  Appends JSON lines to gpurun_out/r4_gen_bench.jsonl and prints a one-line summary each.
set -o pipefail
mkdir -p gpurun_out
for a in "--workload c2 --engine generic" "--workload c2" "--workload c2 --net 64,16,1,1" "--workload c2 --net 20,10,1,1" \
         "--workload c2 --net 128,32,1,1 --steps 3" "--workload c2 --net 40,20,2,2" "--workload c2 --net 40,20,1,1,82" "--workload noisy --engine generic" "--workload noisy"; do
  timeout -k 10 300 python bench.py $a --no-cpu-baseline > gpurun_out/r4_gen_bench.tmp 2> gpurun_out/r4_gen_bench.err || { echo "FAILED: $a"; tail -5 gpurun_out/r4_gen_bench.err; exit 1; }
  python - "$a" <<'PY'
import json, sys
r = json.loads(open("gpurun_out/r4_gen_bench.tmp").read().strip().splitlines()[-1])
print(sys.argv[1], "| %.3e evals/s  %.1f ms  frac %.3f exec %.3f" % (r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["frac_executed"]))
PY
  cat gpurun_out/r4_gen_bench.tmp >> gpurun_out/r4_gen_bench.jsonl
done
