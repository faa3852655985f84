cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $O/prof_probe -- python tools/micro/fl_probe.py > $O/prof_probe.log 2>&1
DB=$(find $O/prof_probe -name "*.db" | head -1)
python tools/rocpd_stats.py $DB | head -12
rm -rf $O/prof_probe
