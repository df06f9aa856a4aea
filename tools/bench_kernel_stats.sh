#!/bin/bash
# rocprofv3 --kernel-trace --stats of the headline bench leg only -> gpurun_out/<tag>/bench_kernel_stats.csv + bench.out
# usage (on the GPU box): bash tools/bench_kernel_stats.sh <tag> [steps]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-ks}; STEPS=${2:-3}; O=$R/gpurun_out/$TAG
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/b -- python $R/bench.py --steps $STEPS --warmup 1 --no-extra-legs --no-k64 --no-cpu-baseline > $O/bench.out 2> $O/bench.err
K=$(find $O/b -name "*kernel_stats.csv" | head -1); cp "$K" $O/bench_kernel_stats.csv; rm -rf $O/b
tail -c 200 $O/bench.out
