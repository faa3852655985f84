# usage: bash tools/prof_quick.sh <tag>   -> gpurun_out/prof_<tag>/  (kernel trace of a short bench run)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$1 -- python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-roofline > $O/prof_$1.log 2>&1
tail -1 $O/prof_$1.log | cut -c1-300
