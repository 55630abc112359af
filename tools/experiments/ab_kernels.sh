#!/bin/bash
# A/B of the three K3 launches alone (tools/experiments/time_mlp_kernels.py) between two builds of the library on ONE box,
# alternating processes:   tools/experiments/ab_kernels.sh build/lib_base.so benerf_amd/libbenerf_hip.so [rounds] [n_rays n_samples]
# prints one line per process; compare the medians by eye (box-to-box spread of this pool: +-2 %, within a box: +-0.5 %).
A=$1; B=$2; R=${3:-3}; shift; shift; shift
for i in $(seq 1 $R); do
  BENERF_HIP_LIB=$A python tools/experiments/time_mlp_kernels.py "$@"
  BENERF_HIP_LIB=$B python tools/experiments/time_mlp_kernels.py "$@"
done
