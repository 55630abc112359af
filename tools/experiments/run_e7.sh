cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/e7
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/e7/tests.log 2>&1
tail -5 gpurun_out/e7/tests.log
(time timeout 900 python bench.py 2>gpurun_out/e7/bench_err.log | tail -1 > gpurun_out/e7/bench_C2.json) 2>&1 | grep real
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/e7/trace8 -o t -- python bench.py --batch-fraction 8 --primary-only --no-cpu-baseline --steps 8 --warmup 3 > /dev/null 2>&1
python tools/step_timeline.py $(find gpurun_out/e7/trace8 -name "t_kernel_trace.csv") > gpurun_out/e7/timeline_C2_one_eighth.txt 2>&1
rm -rf gpurun_out/e7/trace8
python - <<'PY'
import json
d=json.loads(open('gpurun_out/e7/bench_C2.json').read())
print(d['value'], d['ms_per_step'], d['roofline']['traffic_source'], d['roofline'].get('traffic'), d['roofline']['hbm'])
PY
head -60 gpurun_out/e7/timeline_C2_one_eighth.txt
