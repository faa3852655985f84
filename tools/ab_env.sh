# usage (GPU box): bash tools/ab_env.sh "VAR=a" "VAR=b" ...  -> ms/step of the default bench under each environment setting
# (two rounds, so that drift over the call shows)
for rep in 1 2; do
for v in "$@"; do
  env $v python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4))"
done
done
