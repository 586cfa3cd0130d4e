#!/bin/bash
# rocprofv3 passes for profiles/: kernel-trace stats, then one PMC pass per counter group (never combined with other traces).
export TMPDIR=/tmp
OUT=gpurun_out/prof_r01
mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 -L > $OUT/counters_list.txt 2>&1
for mode in workspace single_launch; do
  extra=""; [ $mode = single_launch ] && extra="--single-launch"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$mode -o trace -- python3 bench.py $ARGS $extra > $OUT/${mode}_bench.json 2> $OUT/${mode}_trace.err
done
for ctr in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $ctr | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_$tag -o pmc -- python3 bench.py $ARGS > $OUT/pmc_${tag}.json 2> $OUT/pmc_${tag}.err
  echo "pmc $ctr rc=$?"
done
find $OUT -name "*.csv" | head -50
