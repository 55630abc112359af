cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/e5
timeout 1500 python -m pytest tests/test_api_gpu.py -x -q -m gpu > gpurun_out/e5/tests.log 2>&1
tail -8 gpurun_out/e5/tests.log
for i in 1 2 3; do
timeout 300 python bench.py --primary-only --no-cpu-baseline --no-pipeline 2>/dev/null | tail -1 > gpurun_out/e5/bench_serial_$i.json
timeout 300 python bench.py --primary-only --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/e5/bench_pipe_$i.json
done
for W in C2 C4 C5; do
timeout 300 python bench.py --workload $W --batch-fraction 8 --primary-only --no-cpu-baseline --no-pipeline 2>/dev/null | tail -1 > gpurun_out/e5/bench8_${W}_serial.json
timeout 300 python bench.py --workload $W --batch-fraction 8 --primary-only --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/e5/bench8_${W}_pipe.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/e5/bench*.json')):
    try:
        d=json.loads(open(f).read()); print(f, d['value'], d['ms_per_step'], d['config'].get('median_ms_per_step'))
    except Exception as e: print(f, 'ERR', e)
PY
