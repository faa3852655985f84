#!/bin/bash
# A/B against the committed version of ONE source: builds scenario_wise_rec/_lib/variants/libswr_<name>.so from `git show <rev>:<file>`
# linked with the current objects of the other sources.   usage: tools/build_old_variant.sh <name> <file.hip> [rev]
set -e
name=$1; src=$2; rev=${3:-HEAD}
root=$(cd "$(dirname "$0")/.." && pwd); pkg=$root/scenario-wise-rec_amd
mkdir -p $pkg/scenario_wise_rec/_lib/variants
tmp=$pkg/csrc/_old_$$_$src
git -C $root show $rev:scenario-wise-rec_amd/csrc/$src > $tmp
obj=/tmp/old_$$.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$root/include -I$pkg/csrc -Wno-unused-result -DNDEBUG -c -x hip $tmp -o $obj
rm -f $tmp
objs=$(ls $pkg/build/*.o | grep -v "/${src%.hip}.o" | grep -v variant_)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $obj -o $pkg/scenario_wise_rec/_lib/variants/libswr_$name.so
rm -f $obj
echo built $pkg/scenario_wise_rec/_lib/variants/libswr_$name.so
