# usage (GPU box): bash tools/prof_configs_all.sh <tag>  -> for every other BASELINE configuration: the bench line
# (gpurun_out/<tag>_bench_cfgN.json) and the per-step kernel breakdown (gpurun_out/<tag>_cfgN_per_step.txt)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-x}; cd $R
for c in ${CONFIGS:-1 3 4 5 6}; do
  # HBM bytes per kernel of THIS configuration's step (two separate PMC passes) -> profiles/pmc_hbm_cfgN.json, which the bench
  # line below reads for its `traffic` fields
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_cfg${c}_f -- python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $O/pmc_cfg${c}_f.log 2>&1
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_cfg${c}_w -- python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $O/pmc_cfg${c}_w.log 2>&1
  python tools/pmc_to_json.py $(find $O/pmc_cfg${c}_f -name "*.db" | head -1) $(find $O/pmc_cfg${c}_w -name "*.db" | head -1) $O/pmc_hbm_cfg$c.json > /dev/null
  cp $O/pmc_hbm_cfg$c.json profiles/pmc_hbm_cfg$c.json
  rm -rf $O/pmc_cfg${c}_f $O/pmc_cfg${c}_w
  timeout 900 python bench.py --config $c --steps 50 --warmup 10 2>/dev/null | grep '^{' > $O/${T}_bench_cfg$c.json
  rm -rf $O/prof_cfg$c
  timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $O/prof_cfg$c -- python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/prof_cfg$c.log 2>&1
  python tools/rocpd_step.py $(find $O/prof_cfg$c -name "*.db" | head -1) 5 > $O/${T}_cfg${c}_per_step.txt
  rm -rf $O/prof_cfg$c
  python -c "import json; d=json.load(open('$O/${T}_bench_cfg$c.json')); print('cfg$c', round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,2), 'M samples/s')"
done
