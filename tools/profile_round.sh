#!/bin/bash
# Round profiles on the GPU box: kernel-trace stats of the bench step (headline and realistic legs), of the clustering stage on
# planted embeddings and of the training step; PMC passes of the dominant kernels; per-kernel PMC table of one bench step.
# usage (gpurun): bash tools/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/ ; then python tools/profile_digest.py <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r02}; O=$R/gpurun_out/prof_$TAG
mkdir -p $O
stats() {   # name, command...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- "$@" > $O/$name.out 2> $O/$name.err
  K=$(find $O/$name -name "*kernel_stats.csv" | head -1); cp "$K" $O/${name}_kernel_stats.csv; rm -rf $O/$name
}
stats bench python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-k64 --no-realistic
stats real python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-k64
stats msstage python $R/tools/ms_stage_only.py 64 2
stats train python $R/tools/train_bench.py 32 10000 64 3 --bf16
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$C -- python $R/tools/ms_iter_only.py 64 50 128 f16 > $O/pmc_$C.log 2>&1
done
bash $R/tools/pmc_f16.sh f16 prof_$TAG/pmc_sq > /dev/null 2>&1
python $R/tools/pmc_summary.py "f16r_kernel<false, false>" $O/pmc_f16_summary.md $(find $O/pmc_sq $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE -name "*counter_collection.csv" -printf "%h\n" | sort -u)
# per-kernel PMC table: one bench step (headline) and one clustering stage on planted embeddings (block-sparse kernel)
PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $O/pk_bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-k64 --no-realistic > $O/pk_bench.out 2> $O/pk_bench.err
python $R/tools/pmc_kernels.py $(find $O/pk_bench -name "*counter_collection.csv" | head -1) $O/pmc_kernels_bench.md > /dev/null
rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $O/pk_ms -- python $R/tools/ms_stage_only.py 64 1 > $O/pk_ms.out 2> $O/pk_ms.err
python $R/tools/pmc_kernels.py $(find $O/pk_ms -name "*counter_collection.csv" | head -1) $O/pmc_kernels_msstage.md > /dev/null
rm -rf $O/pk_bench $O/pk_ms
python $R/tools/ms_sparse_f16_check.py 64 > $O/sparse_check.out 2>&1
python $R/tools/ms_sparse_breakdown.py 64 > $O/sparse_breakdown.out 2>&1
python $R/tools/pointwise_bench.py > $O/pointwise_bench.out 2>&1
python $R/tools/ms_f16h_check.py 64 > $O/f16h_check.out 2>&1
find $O -name "*.csv" -size +20M -delete
ls -la $O; tail -c 400 $O/bench.out; echo; head -12 $O/bench_kernel_stats.csv | cut -c1-160
