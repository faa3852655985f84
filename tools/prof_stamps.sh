# usage (GPU box): bash tools/prof_stamps.sh <tag>  -> <tag>_stamps.txt: where the branches of the replayed step really are (device wall
# clock written by one-lane kernels at the fork / join points, SWR_STAMPS) and which branch is critical (a spin delay on one at a time)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-x}; cd $R
{
echo "# SWR_STAMPS=all python bench.py --steps 200 (us since the step's first stamp; the stamps themselves add ~15 us to the step)"
SWR_STAMPS=all python bench.py --steps 200 --no-cpu-baseline --no-roofline 2>&1 | grep "stamps\|issued" | head -2
echo "# m_begin/m_end only"
SWR_STAMPS=m_begin,m_end python bench.py --steps 200 --no-cpu-baseline --no-roofline 2>&1 | grep "stamps"
for v in "SWR_SORT_DELAY_US=0" "SWR_SORT_DELAY_US=80" "SWR_DELAY_DW_US=20" "SWR_DELAY_EMBED_US=20" "SWR_RIDERS_FIRST=0"; do
  echo "# $v"
  env $v python bench.py --steps 200 --no-cpu-baseline --no-roofline 2>&1 | grep "issued" | head -1
done
} > $O/${T}_stamps.txt 2>&1
cat $O/${T}_stamps.txt
