# usage (GPU box): bash tools/micro/prof_dw.sh <tag>  -> kernel durations, wave-state / matrix-pipe fractions, LDS conflicts and
# instruction-cache counters of tools/micro/dw_probe.py (both forms of the first layer's weight gradient)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; tag=$1; cd $R
CMD="python tools/micro/dw_probe.py"
rm -rf $O/pd_$tag*
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $O/pd_${tag}_kt -- $CMD > $O/pd_${tag}_kt.log 2>&1
python tools/rocpd_stats.py $(find $O/pd_${tag}_kt -name "*.db" | head -1) | grep -E "kernel|dw_tr|x6w|reduce4" | cut -c1-60,88-160 > $O/dw_${tag}_stats.txt
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $O/pd_${tag}_sq -- $CMD > $O/pd_${tag}_sq.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/pd_${tag}_gui -- $CMD > $O/pd_${tag}_gui.log 2>&1
python tools/pmc_mfma_summary.py $(find $O/pd_${tag}_sq -name "*.db" | head -1) $(find $O/pd_${tag}_gui -name "*.db" | head -1) | grep -E "kernel|dw_tr|x6w|reduce4" >> $O/dw_${tag}_stats.txt
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU -d $O/pd_${tag}_lds -- $CMD > $O/pd_${tag}_lds.log 2>&1
python tools/pmc_summary.py $(find $O/pd_${tag}_lds -name "*.db" | head -1) dw_tr >> $O/dw_${tag}_stats.txt
python tools/pmc_summary.py $(find $O/pd_${tag}_lds -name "*.db" | head -1) x6w >> $O/dw_${tag}_stats.txt
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_VMEM -d $O/pd_${tag}_ic -- $CMD > $O/pd_${tag}_ic.log 2>&1
python tools/pmc_summary.py $(find $O/pd_${tag}_ic -name "*.db" | head -1) dw_tr >> $O/dw_${tag}_stats.txt
python tools/pmc_summary.py $(find $O/pd_${tag}_ic -name "*.db" | head -1) x6w >> $O/dw_${tag}_stats.txt
tail -3 $O/pd_${tag}_lds.log $O/pd_${tag}_ic.log | grep -i -E "error|invalid|not" | head -5 >> $O/dw_${tag}_stats.txt
rm -rf $O/pd_${tag}_kt $O/pd_${tag}_sq $O/pd_${tag}_gui $O/pd_${tag}_lds $O/pd_${tag}_ic
cat $O/dw_${tag}_stats.txt
