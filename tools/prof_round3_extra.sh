# usage (GPU box): bash tools/prof_round3_extra.sh <tag>  -> the round-3 evidence beside tools/prof_round2.sh:
#   uniform-id bench lines (configs 2 and 5), the bf16 perf-mode line, the stand-alone probes (K3 MFMA vs direct, forward
#   product on bf16 planes + its wave-state counters, VALU issue rates)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-x}; cd $R
python bench.py --uniform-ids --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/${T}_bench_uniform_ids.json
python bench.py --config 5 --uniform-ids --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/${T}_bench_cfg5_uniform_ids.json
SWR_GEMM=bf16 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/${T}_bench_bf16_perf_mode.json
SWR_K3_MFMA=1 timeout -k 5 120 python tools/micro/k3_probe.py > $O/${T}_k3_probe.txt 2>&1
timeout -k 5 60 tools/micro/bin/rows3_probe > $O/${T}_rows3_probe.txt 2>&1
bash tools/micro/pmc_bin.sh rows3 $R/tools/micro/bin/rows3_probe >> $O/${T}_rows3_probe.txt 2>&1
timeout -k 5 60 tools/micro/bin/valu_rate > $O/${T}_valu_rate.txt 2>&1
for f in uniform_ids cfg5_uniform_ids bf16_perf_mode; do python -c "import json; d=json.load(open('$O/${T}_bench_$f.json')); print('$f', round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,2), 'M samples/s')"; done
tail -4 $O/${T}_rows3_probe.txt
