#!/bin/bash
# PMC passes over the split-fp16 mean-shift kernel: tools/pmc_f16.sh <variant> <outdir-under-gpurun_out>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; V=${1:-f16}; O=$R/gpurun_out/${2:-pmc_f16}
mkdir -p $O
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p1 -- python $R/tools/ms_iter_only.py 64 10 128 $V > $O/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $O/p2 -- python $R/tools/ms_iter_only.py 64 10 128 $V > $O/p2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_WAVE32_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVES --kernel-trace --output-format csv -d $O/p3 -- python $R/tools/ms_iter_only.py 64 10 128 $V > $O/p3.log 2>&1
find $O -name "*counter_collection.csv" | head; tail -2 $O/p1.log
