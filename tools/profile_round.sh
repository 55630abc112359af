# Runs on the GPU box (gpurun): bench lines of every workload + rocprofv3 passes of the C2 bench command.
#   ROUND=r02 bash tools/profile_round.sh ; python tools/summarize_profile.py gpurun_out/prof_r02_split profiles r02_split ...
set -x
R=${ROUND:-r06}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bench_$R
timeout 900 python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_$R/bench_C2.json
for W in C3 C4 C5; do timeout 600 python bench.py --workload $W --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_$R/bench_$W.json; done
for W in C2 C4 C5; do timeout 600 python bench.py --workload $W --batch-fraction 8 --no-cpu-baseline --primary-only 2>/dev/null | tail -1 > gpurun_out/bench_$R/bench_${W}_one_eighth_batch.json; done
for MODE in split f32; do O=gpurun_out/prof_${R}_$MODE; mkdir -p $O; B="python bench.py --primary-only --mlp-precision $MODE"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- $B --steps 8 --warmup 2 2>/dev/null | tail -1 > $O.bench.json
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- $B --steps 2 --warmup 1 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- $B --steps 2 --warmup 1 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -o p -- $B --steps 2 --warmup 1 > /dev/null 2>&1
find $O -name "*.csv" | grep -v "kernel_trace\|counter_collection\|kernel_stats" | xargs rm -f
done
# round 6: the path an unchanged train.py runs (tools/dropin_driver.py): kernel trace -> per-launch timeline, kernel stats; host times
O=gpurun_out/prof_${R}_dropin; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python tools/dropin_driver.py --steps 8 --warmup 3 2>/dev/null | tail -1 > $O.bench_under_rocprof.json
timeout 300 python tools/dropin_driver.py --steps 30 --warmup 5 --timers --host-times 2>/dev/null | tail -1 > $O.bench.json
python tools/dropin_timeline.py $(find $O/trace -name "t_kernel_trace.csv") > $O.timeline.txt
# the per-rank step of a strong-scaled 8-GPU run (1/8 batch) under the kernel trace, for tools/step_timeline.py
O=gpurun_out/prof_${R}_eighth; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python bench.py --primary-only --batch-fraction 8 --steps 8 --warmup 2 2>/dev/null | tail -1 > $O.bench.json
python tools/step_timeline.py $(find $O/trace -name "t_kernel_trace.csv") > $O.timeline.txt
python tools/step_timeline.py $(find gpurun_out/prof_${R}_split/trace -name "t_kernel_trace.csv") > gpurun_out/prof_${R}_split.timeline.txt
find gpurun_out/prof_${R}_dropin gpurun_out/prof_${R}_eighth -name "*.csv" | grep -v "kernel_trace\|counter_collection\|kernel_stats" | xargs rm -f
du -sh gpurun_out
