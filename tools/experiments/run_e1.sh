set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/e1
timeout 600 python tools/experiments/overlap_probe.py > gpurun_out/e1/overlap.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/e1/alone -o t -- python tools/experiments/time_mlp_kernels.py 4081 192 10 > gpurun_out/e1/alone.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/e1/alone_c -o t -- python tools/experiments/time_mlp_kernels.py 4081 64 10 > gpurun_out/e1/alone_c.log 2>&1
for i in 1 2; do
timeout 300 python bench.py --primary-only --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/e1/bench_side_$i.json
BENERF_DW_STREAM=main timeout 300 python bench.py --primary-only --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/e1/bench_main_$i.json
done
find gpurun_out/e1 -name "*.csv" | grep -v "kernel_stats" | xargs rm -f
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/e1/bench_*.json')):
    try:
        d=json.loads(open(f).read()); print(f, d['value'], d['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
cat gpurun_out/e1/overlap.log | tail -25
