#!/bin/bash
# rocprofv3 passes for profiles/ (round 2): kernel-trace stats of the default bench (configs[2]) and of the other workloads,
# then one PMC pass per counter group on the default bench (never combined with other traces; TCC counters in passes of their own).
# usage (on the GPU box, from the repo root): bash scripts/profile_r02.sh
export TMPDIR=/tmp
OUT=gpurun_out/prof_r02
rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --no-cpu-baseline"
for wl in c3 c2 noisy; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$wl -o trace -- python3 bench.py $ARGS --workload $wl > $OUT/${wl}_bench.json 2> $OUT/${wl}_trace.err
  echo "trace $wl rc=$?"
done
for ctr in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $ctr | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_$tag -o pmc -- python3 bench.py $ARGS --workload c3 > $OUT/pmc_${tag}.json 2> $OUT/pmc_${tag}.err
  echo "pmc $ctr rc=$?"
done
python3 scripts/summarize_profile_r02.py $OUT gpurun_out/profiles_r02
