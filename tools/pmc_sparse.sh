#!/bin/bash
# PMC passes over the block-sparse mean-shift kernel on the bench's trained embeddings: tools/pmc_sparse.sh <outdir-under-gpurun_out> [form]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-pmc_sparse}; F=${2:-0}
mkdir -p $O
python $R/tools/sparse_ab.py warm $F > $O/warm.log 2>&1      # fills the /tmp cache (embeddings, dense rows) outside the profiler
run() { rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$P -- python $R/tools/sparse_ab.py pmc$P $F > $O/$P.log 2>&1; }
P=p1; run SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
P=p2; run SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_MISC
P=p3; run SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVES SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM
P=p4; run FETCH_SIZE
P=p5; run WRITE_SIZE
python $R/tools/pmc_summary.py ms_sparse_f16_kernel $O/summary.md $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 > /dev/null 2>&1
cat $O/summary.md
