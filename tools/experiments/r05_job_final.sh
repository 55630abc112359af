mkdir -p gpurun_out/r05q
timeout 1400 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r05q/gputests_all.txt; tail -3 gpurun_out/r05q/gputests_all.txt
ROUND=r05 bash tools/profile_round.sh > gpurun_out/profile_round_r05.log 2>&1
bash tools/experiments/pmc_mlp.sh > gpurun_out/r05q/sq_counters.log 2>&1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for BF in 1 8; do O=gpurun_out/r05q/trace$BF; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python bench.py --primary-only --batch-fraction $BF --steps 6 --warmup 3 > /dev/null 2>&1
python tools/step_timeline.py $(find $O -name "*kernel_trace.csv" | head -1) > gpurun_out/r05q/step_timeline_bf$BF.txt 2>&1
find $O -name "*.csv" | xargs rm -f; done
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], "rays", d["config"]["rays_global"], "rays/s %.0f" % d["value"], "ms/step %.3f" % d["ms_per_step"], {k: v["avg_ms"] for k, v in d["roofline"]["per_kernel"].items()})'
for W in C2 C4 C5; do for F in 1 8; do python bench.py --workload $W --batch-fraction $F --no-cpu-baseline --primary-only 2>/dev/null | tail -1 | python -c "$P" "$W 1/$F"; done; done > gpurun_out/r05q/one_eighth.txt
cat gpurun_out/r05q/one_eighth.txt
