cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_tnx -- python tools/gemm_probe.py 5 > $O/prof_tnx.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/pmc_tnx -- python tools/gemm_probe.py 2 > $O/pmc_tnx.log 2>&1
