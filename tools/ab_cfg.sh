# usage (GPU box): bash tools/ab_cfg.sh <config> "VAR=a" "VAR=b" ... -> ms/step of `bench.py --config C` under each environment ("-" = defaults), two rounds
C=$1; shift
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = "-" ]; then e=""; else e="$v"; fi
  env $e python bench.py --config $C --steps 60 --warmup 10 --no-cpu-baseline --no-roofline --no-strong-shard 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg$C [$v]', round(d['ms_per_step'],4))"
done
done
