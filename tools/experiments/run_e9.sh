cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/e9
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/e9/tests.log 2>&1
tail -3 gpurun_out/e9/tests.log
cp gpurun_out/parity_report.txt gpurun_out/e9/parity_report.txt 2>/dev/null
(time timeout 900 python bench.py 2>gpurun_out/e9/bench_err.log | tail -1 > gpurun_out/e9/bench_C2.json) 2>&1 | grep real
python - <<'PY'
import json
d=json.loads(open('gpurun_out/e9/bench_C2.json').read())
print(d['value'], d['ms_per_step'], d['roofline']['traffic_source'][:60], d['exact_f32'], {k:v['frac_of_mfma_peak'] for k,v in d['roofline_f32']['per_kernel'].items()}, d['roofline_f32']['frac'], d['roofline_f32']['kernel'])
PY
