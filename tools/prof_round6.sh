# usage (GPU box): bash tools/prof_round6.sh <tag>  -> the evidence set of round 6 in ONE call:
#   tools/prof_round3_c.sh (config 2: kernel stats / timeline / PMC HBM + matrix-pipe passes / full bench line; uniform ids; bf16 perf mode;
#   K3 probe; configs 1, 3-6 with their CPU baselines), the strong-scaling shards (config 2 at 8 192 rows: graphed, world-1 RCCL, timelines;
#   configs 4 and 1 at their short batches), and the default bench line exactly as the driver runs it.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-x}; cd $R
bash tools/prof_round3_c.sh $T
cd $R
bash tools/prof_shard.sh $T 2 8192
bash tools/prof_shard_dp1.sh $T 2 8192
bash tools/prof_shard.sh $T 4 8192
bash tools/prof_shard.sh $T 1 4096
bash tools/prof_dp1.sh $T
cd $R
( time python bench.py > $O/${T}_bench_default.json 2> $O/${T}_bench_default.err ) 2> $O/${T}_bench_default.time
tail -3 $O/${T}_bench_default.time
python -c "
import json
d=json.loads(open('$O/${T}_bench_default.json').read().strip().splitlines()[-1])
print('default line:', d['ms_per_step'], d['value'], d['roofline']['frac'] if d.get('roofline') else None, d['config'].get('strong_shard'))
"
