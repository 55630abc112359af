#!/bin/bash
# Builds variants of the split forward kernel with parts compiled out (BENERF_ABL_*: no weight-fragment loads, no epilogue
# work, no LDS operand reads) next to the shipped library and times the inference launch of each, pipelined schedule only.
# Attribution of the forward kernel's time behind DESIGN.md section 4.  Run here (builds) then on the GPU box (times):
#   bash tools/experiments/ablate_fwd.sh build ; gpurun -- bash tools/experiments/ablate_fwd.sh run
cd "$(dirname "$0")/../.." || exit 1
D=tools/experiments/abl
if [ "$1" = build ]; then
  mkdir -p $D
  for v in "" NOLOAD NOEPI "NOLOAD NOEPI" "NOLOAD NOEPI NOLDS"; do
    tag=$(echo "base $v" | tr ' ' '_'); flags=""; for f in $v; do flags="$flags -DBENERF_ABL_$f"; done
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $flags -c benerf_amd/csrc/mlp_fwd_h.hip -o $D/fwd_$tag.o || exit 1
    objs=$(ls benerf_amd/csrc/*.o | grep -v mlp_fwd_h.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/lib_$tag.so $objs $D/fwd_$tag.o || exit 1
  done
  rm -f $D/*.o; ls -la $D
else
  for lib in $D/lib_*.so; do
    for p in 1 0; do
      echo -n "$(basename $lib) PIPE=$p: "; BENERF_FWD_PIPE=$p BENERF_HIP_LIB=$PWD/$lib python tools/experiments/time_fwd_modes.py 2>&1 | tail -1
    done
  done
fi
