# usage (GPU box): bash tools/ab_shard.sh <config> <batch> "VAR=a VAR2=b" "VAR=c" ...  -> ms/step of `bench.py --config C --batch B` under each
# environment setting (two rounds, so that drift over the call shows); "-" = the defaults
C=$1; B=$2; shift 2
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = "-" ]; then e=""; else e="$v"; fi
  env $e python bench.py --config $C --batch $B --steps 300 --warmup 10 --no-cpu-baseline --no-roofline --no-strong-shard 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg$C b$B [$v]', round(d['ms_per_step'],4))"
done
done
