# Runs on the GPU box (gpurun): bench lines of every workload + rocprofv3 passes of the C2 bench command.
#   ROUND=r02 bash tools/profile_round.sh ; python tools/summarize_profile.py gpurun_out/prof_r02_split profiles r02_split ...
set -x
R=${ROUND:-r05}
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
du -sh gpurun_out
