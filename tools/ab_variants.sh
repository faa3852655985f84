# usage (GPU box): bash tools/ab_variants.sh "<kernel-name regex>" <variant> [<variant> ...]      ("base" = the regular build)
# Per variant: ms/step of the default bench (100 steps) and the in-step average of the matching kernels (rocprofv3 stats).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
PAT="$1"; shift
for v in "$@"; do
  if [ "$v" = base ]; then unset SWR_LIB; else export SWR_LIB=$R/scenario-wise-rec_amd/scenario_wise_rec/_lib/variants/libswr_$v.so; fi
  ms=$(python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],4))")
  rm -rf $O/ab_$v
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/ab_$v -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/ab_$v.log 2>&1
  DB=$(find $O/ab_$v -name "*.db" | head -1)
  echo "== $v: $ms ms/step"
  python tools/rocpd_step.py $DB 10 | grep -E "$PAT|median span" | cut -c1-110
  rm -rf $O/ab_$v
done
