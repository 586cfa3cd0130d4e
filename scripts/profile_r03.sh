#!/bin/bash
# rocprofv3 passes for profiles/ (round 3).  Kernel-trace stats of every bench workload, then PMC passes (one counter group per
# pass, never combined with other trace domains; TCC and FETCH/WRITE in passes of their own) on the default bench (configs[2]) and
# on the noisy workload.  usage (on the GPU box, from the repo root): bash scripts/profile_r03.sh
export TMPDIR=/tmp
OUT=gpurun_out/prof_r03
rm -rf $OUT; mkdir -p $OUT
ARGS="--warmup 1 --no-cpu-baseline"
for wl in c3 c2 noisy c5 c4; do
  steps=3; [ $wl = c4 ] && steps=1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$wl -o trace -- python3 bench.py $ARGS --steps $steps --workload $wl > $OUT/${wl}_bench.json 2> $OUT/${wl}_trace.err
  echo "trace $wl rc=$?"
done
G1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES"
G2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE"
G3="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM"
i=0
for ctr in FETCH_SIZE WRITE_SIZE "$G1" "$G2" "$G3" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_c3_$i -o pmc -- python3 bench.py $ARGS --steps 2 --workload c3 > $OUT/pmc_c3_$i.json 2> $OUT/pmc_c3_$i.err
  echo "pmc c3 [$ctr] rc=$?"
done
i=0
for ctr in "$G1" "$G2" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_noisy_$i -o pmc -- python3 bench.py $ARGS --steps 3 --workload noisy > $OUT/pmc_noisy_$i.json 2> $OUT/pmc_noisy_$i.err
  echo "pmc noisy [$ctr] rc=$?"
done
python3 scripts/summarize_profile_r03.py $OUT gpurun_out/profiles_r03
