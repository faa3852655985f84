# usage (GPU box): bash tools/prof_round3_c.sh <tag>  -> evidence set c of round 3 in ONE call: tools/prof_round2.sh (step profile,
# PMC HBM / wave-state passes, full bench line), the uniform-id and bf16 perf-mode lines, the K3 stand-alone probe, and the other
# configurations (tools/prof_configs_all.sh)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-x}; cd $R
bash tools/prof_round2.sh $T
cd $R
python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-roofline --no-strong-shard 2>/dev/null | grep '^{' > $O/${T}_bench_200steps.json
python bench.py --uniform-ids --steps 100 --warmup 10 --no-cpu-baseline --no-strong-shard 2>/dev/null | grep '^{' > $O/${T}_bench_uniform_ids.json
python bench.py --config 5 --uniform-ids --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/${T}_bench_cfg5_uniform_ids.json
SWR_GEMM=bf16 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-strong-shard 2>/dev/null | grep '^{' > $O/${T}_bench_bf16_perf_mode.json
timeout -k 5 120 python tools/micro/k3_probe.py > $O/${T}_k3_probe.txt 2>&1
for f in 200steps uniform_ids cfg5_uniform_ids bf16_perf_mode; do python -c "import json; d=json.load(open('$O/${T}_bench_$f.json')); print('$f', round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,2), 'M samples/s')"; done
bash tools/prof_configs_all.sh $T
