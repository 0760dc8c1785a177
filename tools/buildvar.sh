#!/bin/bash
# A variant of the library built from the same sources with extra compiler flags, for A/B runs on one box:
#   bash tools/buildvar.sh <name> <flags ...>      -> build/libswmarlin_<name>.so   (select it with SWM_LIB_PATH)
# e.g. bash tools/buildvar.sh p0 -DSWM_LIGHT_PRIO=0 -DSWM_TAIL_PRIO=0   (the library without issue priorities: collect_profiles.sh)
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
d=$root/build/var_$name; mkdir -p $d
cd $root/simpleworks_amd/csrc
pids=()
for f in capi msm ntt vec spmv marlin pedersen; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -I../../include "$@" -c $f.hip -o $d/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p || exit 1; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $root/build/libswmarlin_$name.so $d/capi.o $d/msm.o $d/ntt.o $d/vec.o $d/spmv.o $d/marlin.o $d/pedersen.o && rm -rf $d && echo built $name
