#!/bin/bash
# PMC passes over the generic engine (v50 shapes forced onto it, configs[1] size) -> gpurun_out/r4_gen_pmc/
# ENGINE=spec PMC_OUT=r4_spec_pmc: the same over the network's specialised form
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${PMC_OUT:-r4_gen_pmc}; rm -rf $OUT; mkdir -p $OUT
ARGS="--workload c2 --engine ${ENGINE:-generic} --steps 2 --warmup 1 --no-cpu-baseline $GEN_ARGS"
G1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES"
G2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE"
G3="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM"
i=0
for ctr in "$G1" "$G2" "$G3" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python3 $R/bench.py $ARGS > $OUT/p$i.json 2> $OUT/p$i.err
  echo "pmc [$ctr] rc=$?"
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "generic" in r["Kernel_Name"] or "bnn_spec_forward" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot):
    print(f"{k:28s} {tot[k] / n[k]:.4e}  (per dispatch, {n[k]} dispatches)")
PY
