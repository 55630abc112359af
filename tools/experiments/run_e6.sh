cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
for L in default skew; do
if [ $L = default ]; then unset BENERF_HIP_LIB; else export BENERF_HIP_LIB=build/lib_$L.so; fi
python tools/experiments/time_mlp_kernels.py 4081 128 10 2>/dev/null | grep "^lib="
done; done
