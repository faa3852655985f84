# usage (GPU box): bash tools/ab_roofline.sh base v1 v2 ...  -> for libswr.so and scenario_wise_rec/_lib/variants/libswr_<v>.so: ms/step of the
# default bench and the stand-alone launch times of the first-layer products (the bench line's roofline block)
D=$PWD/scenario-wise-rec_amd/scenario_wise_rec/_lib/variants
for v in "$@"; do
  if [ $v = base ]; then unset SWR_LIB; else export SWR_LIB=$D/libswr_$v.so; fi
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; a=r['also']
print('$v', 'step', round(d['ms_per_step'],4), 'dW+reduce', round(r['avg_launch_ms']*1e3,1), 'fwd', round(a['gemm_rows_x6_kernel(forward)']['avg_launch_ms']*1e3,1), 'dX', round(a['gemm_rows_x6_kernel(dX)']['avg_launch_ms']*1e3,1), 'gather', round(a['embed_gather_kernel']['avg_launch_ms']*1e3,1), 'k3', round(a['k3_direct_sums']['avg_launch_ms']*1e3,1))"
done
