#!/bin/bash
# rocprofv3 kernel statistics of the small kernels at the configs[4] share (scripts/small_kernels_r04.py) -> gpurun_out/r4_small/
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r4_small
python $R/scripts/small_kernels_r04.py > $R/gpurun_out/r4_small/events.jsonl
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4_small/prof -o small -- python $R/scripts/small_kernels_r04.py > $R/gpurun_out/r4_small/under_rocprof.jsonl
find $R/gpurun_out/r4_small/prof -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r4_small/kernel_stats.csv \;
head -12 $R/gpurun_out/r4_small/kernel_stats.csv
cat $R/gpurun_out/r4_small/events.jsonl
