#!/bin/bash
# rocprofv3 kernel stats of the bench step -> gpurun_out/prof_<tag>_kernel_stats.csv ; usage: tools/prof_step.sh TAG [bench args]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-profile "$@" > $R/gpurun_out/prof_${TAG}_run.log 2>&1
F=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1)
cp "$F" $R/gpurun_out/prof_${TAG}_kernel_stats.csv
head -40 $R/gpurun_out/prof_${TAG}_kernel_stats.csv | cut -c1-160
