# usage (on the GPU box, via gpurun): bash tools/prof_round2.sh <tag>
# Everything the round's profiles/ files come from, in one call:
#   1. kernel trace + stats of the default bench command            -> <tag>_kernel_stats_graph.txt, _per_step_breakdown.txt, _one_step_timeline.txt
#   2. two PMC passes (FETCH_SIZE / WRITE_SIZE; never combined with other trace domains) -> <tag>_pmc_hbm.{json,txt}
#   3. one PMC pass of the SQ counters over the same command         -> <tag>_pmc_mfma.txt (MFMA-busy / wave-wait fractions per kernel)
#   4. the full bench line (reads profiles/pmc_hbm_latest.json)      -> <tag>_bench.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-x}; cd $R
CMD="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strong-shard"
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $O/prof_$T -- $CMD > $O/prof_$T.log 2>&1
DB=$(find $O/prof_$T -name "*.db" | head -1)
python tools/rocpd_stats.py $DB > $O/${T}_kernel_stats_graph.txt
python tools/rocpd_step.py $DB 10 > $O/${T}_per_step_breakdown.txt
python tools/rocpd_one_step.py $DB > $O/${T}_one_step_timeline.txt
rm -rf $O/prof_$T
timeout -k 5 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_${T}_fetch -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-strong-shard > $O/pmc_${T}_fetch.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_${T}_write -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-strong-shard > $O/pmc_${T}_write.log 2>&1
python tools/pmc_to_json.py $(find $O/pmc_${T}_fetch -name "*.db" | head -1) $(find $O/pmc_${T}_write -name "*.db" | head -1) $O/${T}_pmc_hbm.json > $O/${T}_pmc_hbm.txt
rm -rf $O/pmc_${T}_fetch $O/pmc_${T}_write
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $O/pmc_${T}_sq -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-strong-shard > $O/pmc_${T}_sq.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/pmc_${T}_gui -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-strong-shard > $O/pmc_${T}_gui.log 2>&1
python tools/pmc_mfma_summary.py $(find $O/pmc_${T}_sq -name "*.db" | head -1) $(find $O/pmc_${T}_gui -name "*.db" | head -1) --json $O/${T}_pmc_mfma.json > $O/${T}_pmc_mfma.txt
cp $O/${T}_pmc_mfma.json profiles/pmc_mfma_latest.json
rm -rf $O/pmc_${T}_sq $O/pmc_${T}_gui
cp $O/${T}_pmc_hbm.json profiles/pmc_hbm_latest.json
timeout 400 python bench.py --steps 20 --warmup 5 > $O/${T}_bench.json 2> $O/${T}_bench.err
tail -c 600 $O/${T}_bench.json; head -12 $O/${T}_pmc_mfma.txt
