cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
rocprofv3 --list-avail 2>/dev/null | grep -i -E "ICACHE|IFETCH|INST_LEVEL|SQ_INSTS_SMEM|SQ_WAIT_INST_ANY|SQ_ACTIVE_INST_MISC|XDL|MFMA" | head -40 > $O/pmc_avail.txt
timeout 200 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $O/pmc_ic -- python tools/gemm_probe.py 2 > $O/pmc_ic.log 2>&1
tail -3 $O/pmc_ic.log
