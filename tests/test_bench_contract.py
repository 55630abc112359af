"""The bench.py output contract (driver + SURVEY 8d), checked on CPU against the committed evidence line of the last
profile round and against bench.py's own bookkeeping helpers."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _latest_line():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_C2.json")))
    assert files, "no committed bench evidence"
    return json.loads(open(files[-1]).read())


def test_bench_line_has_the_contract_fields():
    d = _latest_line()
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == "training rays/s" and d["unit"] == "rays/s" and d["higher_is_better"] is True
    assert base["metric"].lower().startswith("training rays/s") or "rays/s" in base["metric"]
    for k in ("value", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "vs_baseline", "dtype", "data", "config", "roofline",
              "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] >= 50 and d["warmup"] >= 10 and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["config"]["workload"].startswith("C2") and "model" not in d["config"]
    # value = rays per step / time per step (wall clock over all timed steps)
    assert abs(d["value"] - d["config"]["rays_per_step_per_gpu"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 2e-3
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and 0 < r["frac"] < 1 and r["frac"] <= r["frac_executed"] < 1
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # achieved = algorithmic FLOPs per launch / measured launch duration
    assert abs(r["achieved"] - r["points_per_launch"] * r["flops_per_point"] / (r["avg_launch_ms"] * 1e-3) / 1e12) / r["achieved"] < 2e-3
    assert r["flops_per_point"] == 1186304 and (r["traffic"] is None or r["traffic"] > 0)
    # the dominant kernel fits the step twice (coarse + fine launch)
    assert 2 * r["avg_launch_ms"] < d["ms_per_step"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["unit"] == "rays/s" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]


def test_bench_bookkeeping_helpers():
    import bench
    from benerf_amd import workloads as WL
    wl = WL.WORKLOADS["C2"] if hasattr(WL, "WORKLOADS") else WL.get("C2")
    # training = 3 x forward FLOPs; executed MFMAs per product in split mode: 3 in each of fwd / dX / dW (hi/lo-split
    # operands everywhere); the opt-in reduced-precision backward: 3 + 2 + 1
    assert bench.EXECUTED_PER_PRODUCT["split"] == {"mlp_fwd": 3, "mlp_bwd_dx": 3, "mlp_bwd_dw": 3}
    assert bench.EXECUTED_PER_PRODUCT["split_f16bwd"] == {"mlp_fwd": 3, "mlp_bwd_dx": 2, "mlp_bwd_dw": 1}
    # h0..h7 + hv and dY0..dY7 + dhv (the linear feature layer's output and gradient are not saved: dW composes them from dhv^T h7),
    # PE / PE(dir) saved like every operand: hi + 8-bit code
    assert bench.DW_BYTES_PER_POINT["split"] == 3 * 2 * 2176 + 3 * 96 + 8
    assert WL.rays_per_step(wl) == 4081 and WL.rays_per_step(dict(wl, bins=4)) == 5 * 1024 + 19 * 107      # dense event bins: B + 1 event poses
    assert bench.physical_cores() >= 1
    b = bench.algorithmic_bytes_per_step(wl, wl["channels"])
    assert 3e7 < b < 1e8          # weights + gradients + Adam state + per-ray I/O: tens of MB, not the GBs of saved activations


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher in the environment re-runs itself as two ranks under
    torch.distributed.run (rendezvous on 127.0.0.1) and proves the group spans them: an all-reduce of ones sees 2 ranks and
    the step's three gradient buckets are exchanged and timed.  gloo stands in for RCCL here (no GPU): same launch path."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selfcheck-only", "--backend", "gloo",
                          "--workload", "C4"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["rccl_ranks_seen"] == 2 and d["allreduce_ms"] > 0
    assert d["allreduce_bytes"] == (2 * 595586 + 31) * 4            # fine net, coarse net, trajectory + range-guard verdict
    assert d["scaling"] == "strong", "C4 / C5 are quoted as ONE 8192-ray batch over the GPUs (SURVEY 8e): strong scaling by default"
    # asking for more ranks than a launcher provides is an error, not a silent single-rank run
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selfcheck-only", "--backend", "gloo"],
                         env=env2, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert res.returncode != 0 and "WORLD_SIZE=1" in res.stderr
