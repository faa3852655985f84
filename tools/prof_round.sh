# usage (on the GPU box, via gpurun): bash tools/prof_round.sh <tag>
# kernel trace + stats, two PMC passes (HBM fetch / write; never combined with other trace domains) and the full bench line
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-x}; cd $R
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$T -- python bench.py --steps 20 --warmup 2 --no-cpu-baseline > $O/prof_$T.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_${T}_fetch -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/pmc_${T}_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_${T}_write -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/pmc_${T}_write.log 2>&1
timeout 400 python bench.py > $O/bench_$T.json 2> $O/bench_$T.err
tail -c 1500 $O/bench_$T.json
