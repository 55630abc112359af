#!/bin/bash
# Per-kernel durations of the weight-gradient launch (thin kernel, big kernel, reduce) and of the forward / dX launches, alone at one
# network's point count, for several builds of the library on ONE box (rocprofv3 --kernel-trace --stats around time_mlp_kernels.py):
#   tools/experiments/ab_dw_parts.sh "4081 128 10" benerf_amd/libbenerf_hip.so build/lib_x.so ...
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
ARGS=$1; shift
for L in "$@"; do
  O=gpurun_out/ab_parts/$(basename $L .so); rm -rf $O; mkdir -p $O
  BENERF_HIP_LIB=$L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python tools/experiments/time_mlp_kernels.py $ARGS 2>/dev/null | tail -1
  python - $O <<'P'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if "mlp_" in n:
            print("      %-60s calls %4s  avg %9.1f us  min %9.1f  max %9.1f" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
P
  find $O -name "*.csv" | xargs rm -f
done
