#!/bin/bash
# Round-3 profiles on the GPU box -> gpurun_out/prof_r03/ ; then (in the build container) python tools/profile_r03_digest.py
#   1 kernel-trace stats of the headline bench leg        4 PMC passes of the dense 64-query-per-wave kernel (unstructured rows)
#   2 FETCH_SIZE / WRITE_SIZE passes of that leg          5 kernel-trace stats of the training step (fp32 and bf16)
#   3 SQ counter pass of that leg (per-kernel table)      6 the default bench line
# (counters are collected in their own runs with --kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r03
rm -rf $O; mkdir -p $O
HEAD="python $R/bench.py --no-extra-legs --no-k64 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/b -- $HEAD --steps 3 --warmup 1 > $O/bench.out 2> $O/bench.err
cp $(find $O/b -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv; rm -rf $O/b
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$C -- $HEAD --steps 1 --warmup 1 > $O/pmc_$C.log 2>&1
  cp $(find $O/pmc_$C -name "*counter_collection.csv" | head -1) $O/bench_$C.csv; rm -rf $O/pmc_$C
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pk -- $HEAD --steps 1 --warmup 1 > $O/pk.log 2>&1
cp $(find $O/pk -name "*counter_collection.csv" | head -1) $O/bench_sq.csv; rm -rf $O/pk
bash $R/tools/pmc_f16.sh f16 prof_r03/pmc_dense > /dev/null 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_dense/$C -- python $R/tools/ms_iter_only.py 64 50 128 f16 > $O/pmc_dense_$C.log 2>&1
done
python $R/tools/pmc_summary.py "ms_iterate_f16w_kernel<4, false, true>" $O/pmc_dense_summary.md $(find $O/pmc_dense -name "*counter_collection.csv" -printf "%h\n" | sort -u) > /dev/null
rm -rf $O/pmc_dense
for mode in "" "--bf16"; do
  n=train${mode:+_bf16}
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python $R/tools/train_bench.py 32 10000 64 3 $mode > $O/$n.out 2> $O/$n.err
  cp $(find $O/t -name "*kernel_stats.csv" | head -1) $O/${n}_kernel_stats.csv; rm -rf $O/t
done
python $R/bench.py > $O/bench_default.out 2> $O/bench_default.err
find $O -name "*.csv" -size +30M -delete
ls -la $O; tail -c 300 $O/bench.out
