cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c -- python bench.py --steps 20 --warmup 2 --no-cpu-baseline > $O/prof_c.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_c_fetch -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/pmc_c_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_c_write -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/pmc_c_write.log 2>&1
timeout 400 python bench.py > $O/bench_r01_c.json 2> $O/bench_r01_c.err
tail -c 600 $O/bench_r01_c.json
ls -la $O/prof_c/* $O/pmc_c_fetch/* | head
