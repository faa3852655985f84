# usage: bash tools/prof_dp1.sh <tag>  -> kernel trace of the N > 1 code path at world size 1 (RCCL, two graphs + all-gather)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
SWR_BENCH_FORCE_DP=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$1 -- python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-roofline > $O/prof_$1.log 2>&1
grep "timed region" $O/prof_$1.log
