mkdir -p gpurun_out/r05l
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_api_gpu.py -x -q -k "mlp or bwd or dw or grad or fwd or resume or checkpoint or pack" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05l/trace8; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python bench.py --primary-only --batch-fraction 8 --steps 20 --warmup 3 > /dev/null 2>&1
python - $O <<'P'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if any(k in n for k in ("compose", "fuse_views", "dw_reduce", "pack_pair", "adam")):
            print("      %-50s calls %4s  avg %9.1f us" % (n[:50], r["Calls"], float(r["AverageNs"]) / 1e3))
P
find $O -name "*.csv" | xargs rm -f
P='import json,sys; d=json.loads(sys.stdin.read()); print(d["config"]["workload"][:2], "batch 1/%d" % d["config"]["batch_fraction"], "rays", d["config"]["rays_global"], "rays/s %.0f" % d["value"], "ms/step %.3f" % d["ms_per_step"])'
for F in 1 8; do python bench.py --batch-fraction $F --no-cpu-baseline --primary-only 2>/dev/null | tail -1 | python -c "$P"; done
