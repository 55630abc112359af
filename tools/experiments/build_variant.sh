#!/bin/bash
# Builds a variant of libbenerf_hip.so with extra compiler flags into build/lib_<name>.so (git-ignored; travels with gpurun):
#   tools/experiments/build_variant.sh nodma -DDWS_NO_DMA
#   BENERF_HIP_LIB=build/lib_nodma.so python tools/experiments/time_mlp_kernels.py
# Only the MLP translation units are recompiled with the flags; the other objects are the shipped build's.
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/../.." && pwd)
src=$root/benerf_amd/csrc
out=$root/build/var_$name
mkdir -p $out
(cd $src && make -s all)
objs=""
for f in api pose rays composite sample_pdf loss events_adam helpers; do objs="$objs $src/$f.o"; done
for f in mlp_pack mlp_fwd mlp_fwd_h mlp_bwd mlp_bwd_h mlp_bwd_s mlp_dw mlp_dw_h mlp_dw_s; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c $src/$f.hip -o $out/$f.o &
done
wait
for f in mlp_pack mlp_fwd mlp_fwd_h mlp_bwd mlp_bwd_h mlp_bwd_s mlp_dw mlp_dw_h mlp_dw_s; do objs="$objs $out/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/build/lib_$name.so $objs
echo built $root/build/lib_$name.so
