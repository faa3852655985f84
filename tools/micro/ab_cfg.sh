# usage (GPU box): bash tools/micro/ab_cfg.sh <config> "VAR=a" "VAR=b" ...  -> ms/step of bench.py --config <config> under each setting, twice
c=$1; shift
for rep in 1 2; do
for v in "$@"; do
  env $v python bench.py --config $c --steps 100 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg$c $v', round(d['ms_per_step'],4))"
done
done
