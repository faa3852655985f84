# usage (GPU box): bash tools/pmc_step.sh <tag> [bench args]  -> gpurun_out/pmc_<tag>.txt: matrix-pipe / wave-state fractions per kernel of the step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; tag=$1; shift
rm -rf $O/pmcs_${tag}_sq $O/pmcs_${tag}_gui
cd $R
timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $O/pmcs_${tag}_sq -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline "$@" > $O/pmcs_${tag}_sq.log 2>&1
timeout -k 5 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/pmcs_${tag}_gui -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline "$@" > $O/pmcs_${tag}_gui.log 2>&1
python tools/pmc_mfma_summary.py $(find $O/pmcs_${tag}_sq -name "*.db" | head -1) $(find $O/pmcs_${tag}_gui -name "*.db" | head -1) | tee $O/pmc_${tag}.txt
rm -rf $O/pmcs_${tag}_sq $O/pmcs_${tag}_gui
