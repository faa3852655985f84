# usage: bash tools/prof_config.sh <config> -> gpurun_out/prof_cfg<config>/ (kernel trace of a short bench run of another BASELINE configuration)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
rm -rf $O/prof_cfg$1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_cfg$1 -- python bench.py --config $1 --steps 10 --warmup 2 --no-cpu-baseline --no-roofline > $O/prof_cfg$1.log 2>&1
grep "timed region" $O/prof_cfg$1.log
