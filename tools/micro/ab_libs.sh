# usage (GPU box): bash tools/micro/ab_libs.sh base v1 v2 ...  -> ms/step of the default bench for libswr.so and tools/micro/bin/libswr_<v>.so
for v in "$@"; do
  if [ $v = base ]; then unset SWR_LIB; else export SWR_LIB=$PWD/tools/micro/bin/libswr_$v.so; fi
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4))"
done
