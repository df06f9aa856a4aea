#!/bin/bash
# SQ counters of the ordered kNN sweeps alone (tools/experiments/knn_time.py) -> stdout
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { rocprofv3 --pmc $1 --kernel-trace --output-format csv -d /tmp/pk$2 -- python $R/tools/experiments/knn_time.py > /tmp/pk$2.log 2>&1; }
run "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" 1
run "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" 2
run "SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_MFMA" 3
python3 - <<P
import csv,glob,collections
for i in (1,2,3):
    f=glob.glob(f"/tmp/pk{i}/**/*counter_collection.csv",recursive=True)
    if not f: print("no file",i); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"]
        if "knn_ord" not in k: continue
        k=k[28:54]; acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
    for k,v in acc.items(): print(i,k,{a:round(b/1e6,2) for a,b in v.items()})
P
