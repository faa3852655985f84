# usage (GPU box): bash tools/prof_shard.sh <tag> [config=2] [batch=8192] -> the strong-scaling shard on one GPU:
#   <tag>_shard_bench.json (graphed step), <tag>_shard_dp1_bench.json (world 1 over RCCL), <tag>_shard_one_step.txt (kernel timeline)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-x}; C=${2:-2}; B=${3:-8192}; cd $R
python bench.py --config $C --batch $B --steps 300 --no-cpu-baseline --no-roofline > $O/${T}_c${C}_b${B}_bench.json 2> $O/${T}_c${C}_b${B}_bench.err
SWR_BENCH_FORCE_DP=1 python bench.py --config $C --batch $B --steps 300 --no-cpu-baseline --no-roofline > $O/${T}_c${C}_b${B}_dp1_bench.json 2> $O/${T}_c${C}_b${B}_dp1_bench.err
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $O/prof_${T}_shard -- python bench.py --config $C --batch $B --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/prof_${T}_shard.log 2>&1
DB=$(find $O/prof_${T}_shard -name "*.db" | head -1)
python tools/rocpd_one_step.py $DB > $O/${T}_c${C}_b${B}_one_step.txt
rm -rf $O/prof_${T}_shard
python -c "
import json
for n in ('','_dp1'):
    d=json.loads([l for l in open('$O/${T}_c${C}_b${B}'+n+'_bench.json') if l.startswith('{')][-1]); print('$C', '$B', n, d['ms_per_step'], d['value'])
"
