mkdir -p gpurun_out/r05e
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_f64_truth_gpu.py -x -q -k "bwd or dw or grad or mlp or arithmetic" 2>&1 | tail -4
for i in 1 2 3; do for L in build/lib_base_p1.so benerf_amd/libbenerf_hip.so; do BENERF_HIP_LIB=$L python tools/experiments/time_mlp_kernels.py 4081 128 30 2>/dev/null | tail -1; done; done > gpurun_out/r05e/ab_dx_p1.txt
cat gpurun_out/r05e/ab_dx_p1.txt
BENERF_HIP_LIB=build/lib_trdx.so python tools/experiments/trace_phases.py dx 2>&1 | grep -v amdgpu.ids > gpurun_out/r05e/trace_dx_p1.txt; head -6 gpurun_out/r05e/trace_dx_p1.txt
