set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/e3
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_f64_truth_gpu.py tests/test_path_gpu.py -x -q -m gpu > gpurun_out/e3/tests.log 2>&1
tail -8 gpurun_out/e3/tests.log
for i in 1 2 3; do
BENERF_HIP_LIB=build/lib_base.so timeout 300 python bench.py --primary-only --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/e3/bench_base_$i.json
timeout 300 python bench.py --primary-only --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/e3/bench_new_$i.json
done
BENERF_HIP_LIB=build/lib_base.so python tools/experiments/time_mlp_kernels.py 4081 128 10 | grep "^lib="
python tools/experiments/time_mlp_kernels.py 4081 128 10 | grep "^lib="
BENERF_HIP_LIB=build/lib_base.so python tools/experiments/time_mlp_kernels.py 4081 128 10 | grep "^lib="
python tools/experiments/time_mlp_kernels.py 4081 128 10 | grep "^lib="
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/e3/bench_*.json')):
    try:
        d=json.loads(open(f).read()); print(f, d['value'], d['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
