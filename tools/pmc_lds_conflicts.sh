cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/conf -- python $R/tools/ms_iter_only.py 64 10 128 f16 > /tmp/conf.log 2>&1
python $R/tools/pmc_summary.py "f16r_kernel<false, false>" /tmp/conf.md /tmp/conf/* | tail -12
