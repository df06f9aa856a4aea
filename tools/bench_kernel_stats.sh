#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench's headline step -> gpurun_out/<tag>_bench_kernel_stats.csv
# usage (gpurun): bash tools/bench_kernel_stats.sh <tag> [extra bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r03}; shift
O=$R/gpurun_out/prof_$TAG; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-k64 --no-extra-legs "$@" > $O/bench.out 2> $O/bench.err
K=$(find $O/bench -name "*kernel_stats.csv" | head -1); cp "$K" $O/bench_kernel_stats.csv; rm -rf $O/bench
tail -c 300 $O/bench.err; head -40 $O/bench_kernel_stats.csv | cut -c1-200
