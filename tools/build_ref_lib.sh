#!/bin/bash
# Build libldot from the kernel sources of ANOTHER commit into tools/bin/libldot_<name>.so (git-ignored; it travels to the GPU box), for
# side-by-side runs against the working tree's library in one gpurun call:
#   tools/build_ref_lib.sh 4338591 r6 && gpurun -- 'tools/ab_lib.sh "tools/bin/libldot_r6.so lightningdot_amd/libldot.so" 3 20'
# (the Python side of the working tree drives both: the commit must export the same C ABI)
set -e
C=${1:?commit}; N=${2:?name}; X=${3:-}   # third argument "ablation": compile with -DLDOT_ABLATION (the LDOT_DEBUG_* hooks)
DEF=; [ "$X" = ablation ] && DEF=-DLDOT_ABLATION
ROOT=$(cd "$(dirname "$0")/.." && pwd); W=$ROOT/tools/bin/ref_$N
rm -rf $W; mkdir -p $W/lightningdot_amd/csrc $W/include $W/obj
cd $ROOT
for f in $(git ls-tree --name-only $C lightningdot_amd/csrc/); do git show $C:$f > $W/lightningdot_amd/csrc/$(basename $f); done
git show $C:include/ldot.h > $W/include/ldot.h
cd $W
for s in lightningdot_amd/csrc/*.hip; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-function $DEF -c $s -o obj/$(basename ${s%.hip}).o ) &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/bin/libldot_$N.so obj/*.o
ls -la $ROOT/tools/bin/libldot_$N.so
