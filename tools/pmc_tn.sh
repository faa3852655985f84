cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM -d $O/pmc6 -- python tools/gemm_probe.py 2 > $O/pmc6.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $O/pmc7 -- python tools/gemm_probe.py 2 > $O/pmc7.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $O/pmc8 -- python tools/gemm_probe.py 2 > $O/pmc8.log 2>&1
tail -2 $O/pmc6.log
