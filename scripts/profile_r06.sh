#!/bin/bash
# The round-6 evidence set, taken on ONE lease (one gpurun call): the un-profiled default bench line, then the rocprofv3 kernel-stats pass
# of the same command, then the PMC passes (one counter group per pass, never combined with other trace domains; FETCH_SIZE and WRITE_SIZE
# in passes of their own), then the other workloads, the generic engine and the small kernels.  Everything lands in gpurun_out/prof_r06 and
# is condensed by scripts/summarize_profile_r06.py into gpurun_out/profiles_r06 (copy that to profiles/).
# usage (on the GPU box): bash scripts/profile_r06.sh            (PART=a: the headline set only -- default line, c3 trace, c3 PMC, default line
#                                                                 again; PART=b: the other workloads, engines and small kernels; default both)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_r06
PART=${PART:-ab}
[ "$PART" != "b" ] && rm -rf $OUT
mkdir -p $OUT
B="python3 $R/bench.py"
case $PART in *a*) echo "== 1. un-profiled default line"; $B > $OUT/default_bench.json 2> $OUT/default_bench.err; echo rc=$?;; esac
ARGS="--warmup 1 --no-cpu-baseline"
echo "== 2. kernel-trace stats"
for wl in c3 c2 noisy c5 c4; do
  case $PART in ab) ;; a) [ $wl != c3 ] && continue;; b) [ $wl = c3 ] && continue;; esac
  steps=3; [ $wl = c4 ] && steps=1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$wl -o trace -- $B $ARGS --steps $steps --workload $wl > $OUT/${wl}_bench.json 2> $OUT/${wl}_trace.err
  echo "trace $wl rc=$?"
done
case $PART in *a*) rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/smallcall -o trace -- python3 $R/scripts/dev/small_call_trace.py > $OUT/smallcall.log 2> $OUT/smallcall_trace.err; echo "trace smallcall rc=$?";; esac
export BNN_SPEC_CACHE=${BNN_SPEC_CACHE:-$R/bnn_chaos_model_amd/csrc/_spec}
for tag in "gen_v50:--workload c2 --engine generic" "gen_h64l16:--workload c2 --net 64,16,1,1 --engine generic" "gen_noisy:--workload noisy --engine generic" \
           "spec_v50:--workload c2 --engine spec" "spec_h64l16:--workload c2 --net 64,16,1,1 --engine spec" "spec_noisy:--workload noisy --engine spec" \
           "spec_t99:--workload c2 --timesteps 99"; do
  case $PART in *b*) ;; *) continue;; esac
  name=${tag%%:*}; a=${tag#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o trace -- $B $ARGS --steps 3 $a > $OUT/${name}_bench.json 2> $OUT/${name}_trace.err
  echo "trace $name rc=$?"
done
case $PART in *b*) ;; *) false;; esac && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/small -o trace -- python3 $R/scripts/small_kernels_r05.py > $OUT/small_events.jsonl 2> $OUT/small_trace.err; echo "trace small rc=$?"
echo "== 3. PMC passes"
G1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES"
G2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE"
G3="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM"
i=0
for ctr in FETCH_SIZE WRITE_SIZE "$G1" "$G2" "$G3" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  case $PART in *a*) ;; *) continue;; esac
  i=$((i+1))
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_c3_$i -o pmc -- $B $ARGS --steps 2 --workload c3 > $OUT/pmc_c3_$i.json 2> $OUT/pmc_c3_$i.err
  echo "pmc c3 [$ctr] rc=$?"
done
for wl in noisy gen_v50 spec_v50; do
  case $PART in *b*) ;; *) continue;; esac
  a="--workload noisy"; [ $wl = gen_v50 ] && a="--workload c2 --engine generic"; [ $wl = spec_v50 ] && a="--workload c2 --engine spec"
  i=0
  for ctr in "$G1" "$G2" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_${wl}_$i -o pmc -- $B $ARGS --steps 3 $a > $OUT/pmc_${wl}_$i.json 2> $OUT/pmc_${wl}_$i.err
    echo "pmc $wl [$ctr] rc=$?"
  done
done
case $PART in *a*) echo "== 4. un-profiled default line again (box drift over the session)"; $B --no-cpu-baseline > $OUT/default_bench_after.json 2> $OUT/default_bench_after.err; echo rc=$?;; esac
python3 $R/scripts/summarize_profile_r06.py $OUT $R/gpurun_out/profiles_r06
