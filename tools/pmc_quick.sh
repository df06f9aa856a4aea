#!/bin/bash
# SQ counter passes of the headline leg only -> gpurun_out/pmcq/{sq,sq2}.csv  (then: python tools/pmc_table.py ...)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcq
rm -rf $O; mkdir -p $O
HEAD="python $R/bench.py --no-extra-legs --no-k64 --no-cpu-baseline"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pk -- $HEAD --steps 1 --warmup 1 > $O/pk.log 2>&1
cp $(find $O/pk -name "*counter_collection.csv" | head -1) $O/sq.csv; rm -rf $O/pk
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM --kernel-trace --output-format csv -d $O/pk2 -- $HEAD --steps 1 --warmup 1 > $O/pk2.log 2>&1
cp $(find $O/pk2 -name "*counter_collection.csv" | head -1) $O/sq2.csv; rm -rf $O/pk2
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $O/pk3 -- $HEAD --steps 1 --warmup 1 > $O/pk3.log 2>&1
cp $(find $O/pk3 -name "*counter_collection.csv" | head -1) $O/sq3.csv; rm -rf $O/pk3
find $O -name "*.csv" -size +30M -delete
ls -la $O
