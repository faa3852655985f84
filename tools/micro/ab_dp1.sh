# usage (GPU box): bash tools/micro/ab_dp1.sh "VAR=a" "VAR=b" ...  -> ms/step of the N > 1 code path at world size 1 over RCCL
for rep in 1 2; do
for v in "$@"; do
  env SWR_BENCH_FORCE_DP=1 $v python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dp1 $v', round(d['ms_per_step'],4))"
done
done
