cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_s4 -- python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-roofline --no-graph > $O/prof_s4.log 2>&1
tail -1 $O/prof_s4.log | cut -c1-300
