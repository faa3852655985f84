# usage (GPU box): bash tools/prof_shard_dp1.sh <tag> [config=2] [batch=8192] -> kernel timeline of ONE data-parallel step (world 1 over RCCL)
# of the strong-scaling shard: <tag>_c<C>_b<B>_dp1_one_step.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-x}; C=${2:-2}; B=${3:-8192}; cd $R
SWR_BENCH_FORCE_DP=1 timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $O/prof_${T}_sdp1 -- python bench.py --config $C --batch $B --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-strong-shard > $O/prof_${T}_sdp1.log 2>&1
DB=$(find $O/prof_${T}_sdp1 -name "*.db" | head -1)
python tools/rocpd_one_step.py $DB > $O/${T}_c${C}_b${B}_dp1_one_step.txt
rm -rf $O/prof_${T}_sdp1
