#!/bin/bash
# rocprofv3 kernel-trace stats of the c5 bench (configs[4] share) in every arithmetic, and PMC passes on the bf16 kernel.
export TMPDIR=/tmp
OUT=gpurun_out/prof_r03_lowp; rm -rf $OUT; mkdir -p $OUT gpurun_out/profiles_r03
for pr in f32 bf16 bf16x3 bf16x6 f16 f16x3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$pr -o trace -- python3 bench.py --workload c5 --precision $pr --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${pr}_bench.json 2> $OUT/${pr}.err
  echo "trace $pr rc=$?"
done
python3 - <<'PY'
import csv, os
out = open("gpurun_out/profiles_r03/r03_kernel_stats_c5_precisions.csv", "w")
out.write("arithmetic,Name,Calls,TotalDurationNs,AverageNs,PercentageOfAllKernels\n")
for pr in ("f32", "bf16", "bf16x3", "bf16x6", "f16", "f16x3"):
    f = f"gpurun_out/prof_r03_lowp/{pr}/trace_kernel_stats.csv"
    if not os.path.exists(f):
        continue
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in rows:
        n = r["Name"].split("(")[0].replace("void ", "")
        if "bnn" in n:
            out.write(f"{pr},\"{n}\",{r['Calls']},{r['TotalDurationNs']},{r['AverageNs']},{100 * float(r['TotalDurationNs']) / tot:.3f}\n")
out.close()
print(open("gpurun_out/profiles_r03/r03_kernel_stats_c5_precisions.csv").read())
PY
bash scripts/lowp_pmc.sh bf16 10 > gpurun_out/profiles_r03/r03_pmc_lowp_bf16.txt 2>&1
for pr in f16 f16x3; do python3 bench.py --workload c5 --precision $pr --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/profiles_r03/r03_bench_c5_$pr.json 2>/dev/null; done
tail -25 gpurun_out/profiles_r03/r03_pmc_lowp_bf16.txt
