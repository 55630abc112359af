cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/e8
for i in 1 2; do
timeout 300 python bench.py --primary-only --no-cpu-baseline --mlp-precision f32 --steps 20 2>/dev/null | tail -1 > gpurun_out/e8/f32_side_$i.json
BENERF_DW_STREAM=main timeout 300 python bench.py --primary-only --no-cpu-baseline --mlp-precision f32 --steps 20 2>/dev/null | tail -1 > gpurun_out/e8/f32_main_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/e8/f32_*.json')):
    d=json.loads(open(f).read()); pk=d['roofline']['per_kernel']
    print(f, d['value'], d['ms_per_step'], {k:(v['avg_ms'],v['frac_of_mfma_peak']) for k,v in pk.items()})
PY
ROUND=r04 bash tools/profile_round.sh > gpurun_out/e8/profile_round.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/e8/trace -o t -- python bench.py --primary-only --no-cpu-baseline --steps 8 --warmup 3 > /dev/null 2>&1
python tools/step_timeline.py $(find gpurun_out/e8/trace -name "t_kernel_trace.csv") > gpurun_out/e8/step_timeline_C2.txt 2>&1
rm -rf gpurun_out/e8/trace
ls gpurun_out/bench_r04 gpurun_out/prof_r04_split | head -30
