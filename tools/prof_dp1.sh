# usage (GPU box): bash tools/prof_dp1.sh <tag>  -> the N > 1 code path at world size 1 over RCCL beside the single-GPU step, same box:
#   <tag>_dp1_bench.json (SWR_BENCH_FORCE_DP=1), <tag>_n1_bench.json, <tag>_dp1_one_step.txt (kernel timeline of one data-parallel step)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-x}; cd $R
SWR_BENCH_FORCE_DP=1 python bench.py --steps 200 --no-cpu-baseline --no-roofline --no-strong-shard > $O/${T}_dp1_bench.json 2> $O/${T}_dp1_bench.err
python bench.py --steps 200 --no-cpu-baseline --no-roofline --no-strong-shard > $O/${T}_n1_bench.json 2> $O/${T}_n1_bench.err
SWR_BENCH_FORCE_DP=1 timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $O/prof_${T}_dp1 -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-strong-shard > $O/prof_${T}_dp1.log 2>&1
DB=$(find $O/prof_${T}_dp1 -name "*.db" | head -1)
python tools/rocpd_one_step.py $DB > $O/${T}_dp1_one_step.txt
rm -rf $O/prof_${T}_dp1
python -c "
import json
for n in ('dp1','n1'):
    d=json.loads([l for l in open('$O/${T}_'+n+'_bench.json') if l.startswith('{')][-1]); print(n, d['ms_per_step'], d['value'])
"
