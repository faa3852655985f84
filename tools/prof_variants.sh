# usage: bash tools/prof_variants.sh <suffix> ...   -> kernel traces of short bench runs with _lib/libswr_<suffix>.so (SWR_LIB)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
for v in "$@"; do
  L=$R/scenario-wise-rec_amd/scenario_wise_rec/_lib/libswr_$v.so; [ "$v" = base ] && L=$R/scenario-wise-rec_amd/scenario_wise_rec/_lib/libswr.so
  rm -rf $O/prof_var_$v
  SWR_LIB=$L timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_var_$v -- python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-roofline > $O/prof_var_$v.log 2>&1
done
