cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1; cd $R
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $O/prof_$T -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/prof_$T.log 2>&1
DB=$(find $O/prof_$T -name "*.db" | head -1)
python tools/rocpd_one_step.py $DB > $O/${T}_one_step.txt
python tools/rocpd_step.py $DB 10 > $O/${T}_breakdown.txt
rm -rf $O/prof_$T
