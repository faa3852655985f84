for rep in 1 2; do for c in 5 6 3; do for v in base epiold; do
  if [ $v = base ]; then unset SWR_LIB; else export SWR_LIB=$PWD/scenario-wise-rec_amd/scenario_wise_rec/_lib/variants/libswr_$v.so; fi
  python bench.py --config $c --steps 100 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$v', d['config']['workload'], round(d['ms_per_step'],4))"
done; done; done
