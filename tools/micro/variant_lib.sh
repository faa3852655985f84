#!/bin/bash
# Builds a variant of libswr.so with ONE source recompiled under extra flags (A/B runs through SWR_LIB=...):
#   tools/micro/variant_lib.sh tools/micro/bin/libswr_v.so bnmix.hip -DBM_ROWS=32
set -e
out=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/../.." && pwd)
pkg=$root/scenario-wise-rec_amd
obj=/tmp/variant_$$.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$root/include -I$pkg/csrc -Wno-unused-result -DNDEBUG "$@" -c $pkg/csrc/$src -o $obj
objs=$(ls $pkg/build/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $obj -o $out
rm -f $obj
echo "built $out"
