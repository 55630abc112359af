set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for MODE in split f32; do O=gpurun_out/prof_${ROUND:-r02}_$MODE; mkdir -p $O; B="python bench.py --primary-only --mlp-precision $MODE"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- $B --steps 6 --warmup 2 2>/dev/null | tail -1 > $O.bench.json
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- $B --steps 2 --warmup 1 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- $B --steps 2 --warmup 1 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -o p -- $B --steps 2 --warmup 1 > /dev/null 2>&1
find $O -name "*.csv" | grep -v "kernel_trace\|counter_collection\|kernel_stats" | xargs rm -f
done
du -sh gpurun_out
