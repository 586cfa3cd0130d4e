#!/bin/bash
# Round-6 GPU call runner: each step under its own timeout, logs under gpurun_out/; a step that fails an assertion does not stop the
# call, a step that is KILLED (timeout: 124 / 137) does -- nothing further touches the GPU after that.
#   scripts/r06_call.sh <step> [<step> ...]      steps: scale micro nodes trace smalltiming cprofile smoke surface edges budget gputests bench
#   (abheadline / abnoisy need the variant libraries built first: `python -m bnn_chaos_model_amd.csrc.build -DBNN_NIN16=1 -o libbnn_nin16.so`;
#    libbnn_r05.so = the round-5 library, built from a worktree of commit 6de3a5d)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
mkdir -p gpurun_out
VARIANT=$R/bnn_chaos_model_amd/csrc/libbnn_nfmemset.so
step() {   # step <name> <seconds> <command...>
  local name=$1 secs=$2; shift 2
  echo "=== $name: $*" | tee -a gpurun_out/r06_steps.log
  timeout -k 10 "$secs" "$@" > "gpurun_out/r06_$name.log" 2>&1
  local rc=$?
  echo "=== $name: rc=$rc" | tee -a gpurun_out/r06_steps.log
  tail -n "${TAILN:-12}" "gpurun_out/r06_$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "KILLED: stopping the call" | tee -a gpurun_out/r06_steps.log; exit $rc; fi
  return 0
}
for s in "$@"; do
  case $s in
    scale)   step scale_test 900 python -m pytest tests/test_scale_parity.py -x -q -m gpu -s ;;
    micro)   step probe_micro 200 python scripts/dev/graph_nf_probe3.py micro default
             DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 step probe_micro_nocapture 200 python scripts/dev/graph_nf_probe3.py micro packet-capture-off ;;
    # (the product probe with the memset-header library + a damaged x FAULTS the GPU -- garbage header, out-of-bounds append: run once in
    #  round 6, never again; graph_nf_probe3.py refuses that combination)
    nodes)   step probe_nodes 300 python scripts/dev/graph_nf_probe3.py nodes ;;
    trace)   rm -rf gpurun_out/r06_trace_small
             (cd /tmp && export TMPDIR=/tmp && cd "$R" && step trace_small 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r06_trace_small -- python3 scripts/dev/small_call_trace.py)
             f=$(find gpurun_out/r06_trace_small -name "*kernel_trace.csv" | head -1)
             [ -n "$f" ] && python scripts/dev/trace_gaps.py "$f" | tee gpurun_out/r06_trace_gaps_${TAG:-x}.txt ;;
    smallkernels) TAILN=14 step small_kernels 400 python scripts/small_kernels_r05.py ;;
    stats)   step stats_tests 600 python -m pytest tests/test_stats_stream.py tests/test_hip_parity.py -x -q -m gpu ;;
    manychunks) step many_chunks 400 python scripts/dev/many_small_chunks_timing.py ;;
    smalltiming) step small_timing 300 python scripts/dev/small_forward_timing.py ;;
    abheadline) step ab_headline 900 python scripts/ab_variants.py --workload c3 --reps 3 --steps 4 libbnn_r05.so libbnn_chaos_hip.so ;;
    cprofile) TAILN=60 step cprofile 300 python scripts/dev/dropin_cprofile.py ;;
    example) step example 600 python examples/five_planet_pipeline.py ;;
    smoke)   step smoke 600 python __graft_entry__.py --smoke ;;
    abnoisy) step ab_noisy 900 python scripts/ab_variants.py --workload noisy --reps 3 --steps 6 libbnn_chaos_hip.so libbnn_nin16.so ;;
    surface) step surface 900 python -m pytest tests/test_surface_gpu.py tests/test_scale_parity.py -x -q -m gpu ;;
    edges)   step edges 900 python -m pytest tests/test_hip_edges.py -x -q -m gpu ;;
    budget)  step budget_${TAG:-x} 600 python scripts/dev/dropin_budget.py "${TAG:-x}" ;;
    gputests) TAILN=30 step gputests 1100 python -m pytest tests -x -q -m gpu ;;
    bench)   step bench 600 python bench.py ;;
    *) echo "unknown step $s"; exit 2 ;;
  esac
done
