# usage (GPU box): bash tools/micro/ab_dw.sh base v1 v2 ...  -> stand-alone time of swr_fl_dw_bn (transpose-read form + reduce) for
# libswr.so and tools/micro/bin/libswr_<v>.so (ablation builds: results of the variants are wrong by design)
for v in "$@"; do
  if [ $v = base ]; then unset SWR_LIB; else export SWR_LIB=$PWD/tools/micro/bin/libswr_$v.so; fi
  DW_TIME_ONLY=1 python tools/micro/dw_probe.py 2>&1 | grep "swr_fl_dw_bn"
done
