# usage (GPU box): bash tools/micro/pmc_raw.sh <tag> "<counters>" <binary> [args]  -> mean counter values per kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; tag=$1; ctrs=$2; shift; shift
rm -rf $O/pmcr_$tag
timeout -k 5 120 rocprofv3 --kernel-trace --pmc $ctrs -d $O/pmcr_$tag -- "$@" > $O/pmcr_$tag.log 2>&1
cd $R
python - <<PY
import sqlite3, glob, collections
db = sqlite3.connect(glob.glob("$O/pmcr_$tag/**/*.db", recursive=True)[0])
t = [r[0] for r in db.execute("select name from sqlite_master where type='table' and name like 'rocpd_pmc_event%'")]
rows = db.execute("select name, dispatch_id, counter_name, sum(counter_value) from pmc_events group by dispatch_id, counter_name").fetchall()
per = collections.defaultdict(lambda: collections.defaultdict(list))
for name, _d, c, v in rows:
    per[name.split("(")[0]][c].append(v)
for k, cs in per.items():
    if any(s in k for s in ("rows3", "segsum", "tn3", "bnmix", "tower", "act_bwd")):
        print(k[:48], {c: round(sum(v) / len(v)) for c, v in sorted(cs.items())})
PY
rm -rf $O/pmcr_$tag
