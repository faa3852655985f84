# usage: bash tools/pmc_kernel.sh <tag> <counters...>  -> gpurun_out/pmc_<tag>/ (PMC pass over a short bench run)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
tag=$1; shift
timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $O/pmc_$tag -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_$tag.log 2>&1
tail -1 $O/pmc_$tag.log | cut -c1-200
