#!/bin/bash
# SQ / GRBM PMC passes on `bench.py --workload $1 [more bench args]`, per-kernel averages printed.  usage: bash scripts/pmc_workload.sh noisy
export TMPDIR=/tmp
WL=${1:-noisy}; shift
OUT=gpurun_out/pmc_$WL; rm -rf $OUT; mkdir -p $OUT
i=0
for ctr in "GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python3 bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/p$i.json 2> $OUT/p$i.err || echo "pass $i failed"
done
python3 - "$WL" <<'PY'
import csv, glob, collections, sys
wl = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(f'gpurun_out/pmc_{wl}/p*/pmc_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'forward' in k:
            acc[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
            acc[(k, '_dur_ns')].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
for (k, c), v in sorted(acc.items()):
    print(k, c, '%.5g' % (sum(v) / len(v)))
PY
