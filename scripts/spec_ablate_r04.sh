#!/bin/bash
# Where the specialised form's non-MFMA time goes: the pretrained shapes (configs[1] size) with one piece of the tile removed at a time
# (BNN_GEN_ABLATE, bnn_generic.hip.h; results are wrong by construction -- only the time is read).  Appends to gpurun_out/r4_spec_ablate.txt
set -o pipefail
mkdir -p gpurun_out
for d in "" "BNN_GEN_ABLATE=1" "BNN_GEN_ABLATE=2" "BNN_GEN_ABLATE=4" "BNN_GEN_ABLATE=8" "BNN_GEN_ABLATE=16" "BNN_GEN_ABLATE=31" $EXTRA_VARIANTS; do
  BNN_SPEC_DEFINES="$d" timeout -k 10 300 python bench.py --workload c2 --engine spec --no-cpu-baseline $ABLATE_ARGS > gpurun_out/r4_spec_ablate.tmp 2> gpurun_out/r4_spec_ablate.err || { echo "FAILED: $d"; tail -5 gpurun_out/r4_spec_ablate.err; exit 1; }
  python - "$d" <<'PY' | tee -a gpurun_out/r4_spec_ablate.txt
import json, sys
r = json.loads(open("gpurun_out/r4_spec_ablate.tmp").read().strip().splitlines()[-1])
print("%-24s %.3e evals/s  %.1f ms" % (sys.argv[1] or "(product)", r["value"], r["ms_per_step"]))
PY
done
