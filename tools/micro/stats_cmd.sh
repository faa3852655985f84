# usage (GPU box): bash tools/micro/stats_cmd.sh <tag> <bench args...>  -> per-kernel stats of `python bench.py <args>` (top 14)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1; shift; cd $R
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $O/prof_$T -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline "$@" > $O/prof_$T.log 2>&1
DB=$(find $O/prof_$T -name "*.db" | head -1)
python tools/rocpd_step.py $DB 10 | grep -i "reduce_kernel\|direct_kernel\|finalize_kernel\|median" | head -8
rm -rf $O/prof_$T
