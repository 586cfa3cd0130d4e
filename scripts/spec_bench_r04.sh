#!/bin/bash
# Specialised-form bench lines (round 4): each network through the ahead-of-time generic engine and through its run-time-compiled form.
# Appends JSON lines to gpurun_out/r4_spec_bench.jsonl and prints a one-line summary each.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p gpurun_out
export BNN_SPEC_CACHE=${BNN_SPEC_CACHE:-$R/bnn_chaos_model_amd/csrc/_spec}   # in-tree (private to the checkout), never a predictable world-writable /tmp path
run() {
  timeout -k 10 400 python bench.py $1 --no-cpu-baseline > gpurun_out/r4_spec_bench.tmp 2> gpurun_out/r4_spec_bench.err || { echo "FAILED: $1"; tail -5 gpurun_out/r4_spec_bench.err; exit 1; }
  python - "$1" <<'PY'
import json, sys
r = json.loads(open("gpurun_out/r4_spec_bench.tmp").read().strip().splitlines()[-1])
print(sys.argv[1], "| %.3e evals/s  %.1f ms  frac %.3f exec %.3f" % (r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["frac_executed"]))
PY
  cat gpurun_out/r4_spec_bench.tmp >> gpurun_out/r4_spec_bench.jsonl
}
for a in "${@:-all}"; do :; done
if [ "$1" = "quick" ]; then
  run "--workload c2 --engine spec --spec-w8 0"
  run "--workload c2 --engine spec --spec-w8 1"
  run "--workload c2 --engine generic"
  run "--workload c2"
  exit 0
fi
for net in "" "--net 64,16,1,1" "--net 20,10,1,1" "--net 40,20,2,2" "--net 40,20,1,1,82" "--net 64,32,1,1" "--net 80,20,1,1" "--net 100,30,1,1 --steps 5" "--net 128,32,1,1 --steps 3"; do
  run "--workload c2 $net --engine generic"
  run "--workload c2 $net --engine spec"            # the form the specialiser's measured selection attaches
  case "$net" in ""|*64,16*) run "--workload c2 $net --engine spec --spec-w8 0"; run "--workload c2 $net --engine spec --spec-w8 1";; esac
done
run "--workload noisy --engine generic"
run "--workload noisy --engine spec"
run "--workload noisy"
run "--workload c2 --timesteps 99"
run "--workload c2 --timesteps 99 --engine generic"
