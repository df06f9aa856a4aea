#!/bin/bash
# Round profiles on the GPU box: kernel-trace stats of the bench step + PMC passes of the dominant kernel.
# usage (gpurun): bash tools/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r02}; O=$R/gpurun_out/prof_$TAG
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-k64 --no-realistic > $O/bench.json 2> $O/bench.err
K=$(find $O/bench -name "*kernel_stats.csv" | head -1); cp "$K" $O/bench_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/real -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-k64 > $O/real.json 2> $O/real.err
K=$(find $O/real -name "*kernel_stats.csv" | head -1); cp "$K" $O/real_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$C -- python $R/tools/ms_iter_only.py 64 50 128 f16 > $O/pmc_$C.log 2>&1
done
bash $R/tools/pmc_f16.sh f16 prof_$TAG/pmc_sq > /dev/null 2>&1
python $R/tools/pmc_summary.py f16p_kernel $O/pmc_f16_summary.md $(find $O/pmc_sq $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE -name "*counter_collection.csv" -printf "%h\n" | sort -u)
rm -rf $O/bench $O/real                      # raw traces are large; the stats csv files are kept
find $O -name "*.csv" -size +20M -delete
ls -la $O; cat $O/bench.json | head -c 600; echo; head -30 $O/bench_kernel_stats.csv
