#!/bin/bash
# PMC passes (separate counter groups, kernel-trace only) over scripts/lowp_one.py: bash scripts/lowp_pmc.sh bf16 10
export TMPDIR=/tmp
OUT=gpurun_out/pmc_lowp; rm -rf $OUT; mkdir -p $OUT
i=0
for ctr in "GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python3 scripts/lowp_one.py "$@" > $OUT/p$i.log 2> $OUT/p$i.err || echo "pass $i failed"
done
python3 - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc_lowp/p*/pmc_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','')
        if 'lowp' in k or 'forward_kernel' in k:
            acc[(k,r['Counter_Name'])].append(float(r['Counter_Value']))
            acc[(k,'_dur_ns')].append(float(r['End_Timestamp'])-float(r['Start_Timestamp']))
for (k,c),v in sorted(acc.items()): print(k, c, '%.5g'%(sum(v)/len(v)))
PY
