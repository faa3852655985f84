python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "rowmat" 2>&1 | tail -2
for rep in 1 2; do for v in 0 1; do SWR_ROWMAT_PIPE=$v python bench.py --config 5 --steps 50 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('pipe=$v', d['config']['workload'], round(d['ms_per_step'],4))"; done; done
