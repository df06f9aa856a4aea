#!/bin/bash
# Round-6 profiles on the GPU box -> gpurun_out/prof_r06/ ; then (in the build container) python tools/profile_r06_digest.py
#   1 kernel-trace stats of the headline bench leg        4 kernel-trace stats + SQ counters of the HPNet-on flow
#   2 FETCH_SIZE / WRITE_SIZE passes of that leg          5 the default bench line
#   3 SQ counter passes of that leg (per-kernel table incl. scratch, LDS conflicts, clock)
# (counters are collected in their own runs with --kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r06
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
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM --kernel-trace --output-format csv -d $O/pk2 -- $HEAD --steps 1 --warmup 1 > $O/pk2.log 2>&1
cp $(find $O/pk2 -name "*counter_collection.csv" | head -1) $O/bench_sq2.csv; rm -rf $O/pk2
rocprofv3 --kernel-trace --stats --output-format csv -d $O/h -- $HEAD --hpnet --steps 3 --warmup 1 > $O/hpnet.out 2> $O/hpnet.err
cp $(find $O/h -name "*kernel_stats.csv" | head -1) $O/hpnet_kernel_stats.csv; rm -rf $O/h
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/hk -- $HEAD --hpnet --steps 1 --warmup 1 > $O/hk.log 2>&1
cp $(find $O/hk -name "*counter_collection.csv" | head -1) $O/hpnet_sq.csv; rm -rf $O/hk
python $R/bench.py > $O/bench_default.out 2> $O/bench_default.err
find $O -name "*.csv" -size +30M -delete
ls -la $O; tail -c 300 $O/bench.out
