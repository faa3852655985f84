# usage (GPU box): bash tools/micro/ab_flf.sh base nomma noepi ...  -> stand-alone time of swr_fl_fwd at config 2 for libswr.so and the
# ablation builds _lib/variants/libswr_flf_<v>.so (python tools/build_variant.py flf_<v> first_layer.hip -DFLF_NO_...)
for v in "$@"; do
  if [ $v = base ]; then unset SWR_LIB; else export SWR_LIB=$PWD/scenario-wise-rec_amd/scenario_wise_rec/_lib/variants/libswr_flf_$v.so; fi
  FL_ONLY=fwd python tools/micro/fl_probe.py 2>/dev/null | tail -1
done
