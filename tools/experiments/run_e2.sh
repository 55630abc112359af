set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/e2
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_f64_truth_gpu.py -x -q -m gpu > gpurun_out/e2/tests.log 2>&1
tail -15 gpurun_out/e2/tests.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/e2/alone -o t -- python tools/experiments/time_mlp_kernels.py 4081 192 10 > gpurun_out/e2/alone.log 2>&1
grep "^lib=" gpurun_out/e2/alone.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/e2/alone_c -o t -- python tools/experiments/time_mlp_kernels.py 4081 64 10 > gpurun_out/e2/alone_c.log 2>&1
grep "^lib=" gpurun_out/e2/alone_c.log
for i in 1 2; do
timeout 300 python bench.py --primary-only --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/e2/bench_$i.json
done
find gpurun_out/e2 -name "*.csv" | grep -v "kernel_stats" | xargs rm -f
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/e2/bench_*.json')):
    try:
        d=json.loads(open(f).read()); print(f, d['value'], d['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
head -7 gpurun_out/e2/alone/t_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
head -7 gpurun_out/e2/alone_c/t_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
