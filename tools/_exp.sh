cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py tests/test_parity_gpu.py tests/test_baseline_shapes_gpu.py -x -q 2>&1 | tail -3
python bench.py --config 5 --steps 50 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg5', d['ms_per_step'])"
python bench.py --config 3 --steps 50 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg3', d['ms_per_step'])"
