# usage (GPU box): bash tools/micro/ab_variants.sh base v1 v2 ...  -> ms/step of the default bench for libswr.so and the builds
# scenario_wise_rec/_lib/variants/libswr_<v>.so of tools/build_variant.py (two rounds, so that drift over the call shows)
for rep in 1 2; do
for v in "$@"; do
  if [ $v = base ]; then unset SWR_LIB; else export SWR_LIB=$PWD/scenario-wise-rec_amd/scenario_wise_rec/_lib/variants/libswr_$v.so; fi
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4))"
done
done
