# usage (GPU box): bash tools/micro/pmc_bin.sh <tag> <binary> [args]  -> wave-state / matrix-pipe fractions per kernel of a stand-alone probe
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; tag=$1; shift
rm -rf $O/pmc_${tag}_sq $O/pmc_${tag}_gui
timeout -k 5 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $O/pmc_${tag}_sq -- "$@" > $O/pmc_${tag}_sq.log 2>&1
timeout -k 5 120 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/pmc_${tag}_gui -- "$@" > $O/pmc_${tag}_gui.log 2>&1
cd $R
python tools/pmc_mfma_summary.py $(find $O/pmc_${tag}_sq -name "*.db" | head -1) $(find $O/pmc_${tag}_gui -name "*.db" | head -1) | tee $O/pmc_${tag}.txt
rm -rf $O/pmc_${tag}_sq $O/pmc_${tag}_gui
