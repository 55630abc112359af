#!/usr/bin/env python
"""Training-throughput bench of the BeNeRF hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N > 1 without WORLD_SIZE in the environment: spawns its N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full training iteration (train.py:153-394 semantics) on a synthetic batch
of the BASELINE.json workload (default C2: benerf_unreal/livingroom_gray camera, gray,
2 x 1024 event rays + 19 x 107 blur rays = 4081 rays, 64 coarse + 128 fine samples):
trajectory spline -> rays -> PE + coarse MLP -> compositing -> sample_pdf -> PE + fine MLP ->
compositing -> event + blur loss -> full backward (both MLPs, rays, spline) -> gradient
all-reduce (N > 1) -> Adam with LR decay -> weight re-pack.  Inputs are resident in HBM when
the timed region starts.  Data-parallel: --scaling weak (default; every rank renders its own 4081
rays) or --scaling strong (the global batch of the workload is split over the ranks).

Prints ONE JSON line (rank 0) with the metric, the roofline of the dominant kernel (the
fused MLP against the MFMA roof with the algorithmic 1 186 304 FLOP per point, HIP-event timed
inside the run; HBM figures of the bandwidth-bound dW launch under `hbm`) and a CPU baseline (the
oracle's op-for-op torch-CPU step, host cores and CPU model stated: full C2 step + C1).
"""
import argparse
import json
import os
import sys
import time

import numpy as np      # noqa: E402
import torch            # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import benerf_amd       # noqa: E402
benerf_amd.configure_runtime()      # GPU_MAX_HW_QUEUES = 8 unless the user chose a value; before the HIP runtime initialises (no device call yet)

F32_MFMA_PEAK_TFLOPS = 157.3    # MI355X f32 matrix peak (MI355X_MICROARCH.md)
F16_MFMA_PEAK_TFLOPS = 2516.6   # dense f16 matrix peak: 256 CUs x 4 SIMDs x 1024 flop/clk x 2.4 GHz
HBM_PEAK_GBS = 8000.0
# f16 MFMAs the split-mode kernels execute per algorithmic product block: hi/lo-split operands, hi x hi + hi x lo +
# lo x hi, in the forward pass and in both backward GEMMs; the opt-in reduced-precision backward (split_f16bwd): dX chain 2 (f16
# gradient x hi/lo weight), dW 1 (f16 x f16); exact-f32 mode: 1 f32 MFMA chain everywhere
EXECUTED_PER_PRODUCT = {"split": {"mlp_fwd": 3, "mlp_bwd_dx": 3, "mlp_bwd_dw": 3},
                        "split_f16bwd": {"mlp_fwd": 3, "mlp_bwd_dx": 2, "mlp_bwd_dw": 1}}
# bytes per sample point the split-mode dW launch streams once (saved activations h0..h7, hv and as many gradients -
# f16 + 8-bit residual code, 3 bytes per value; since round 5 neither the linear feature layer's output nor its gradient is saved
# (split_f16bwd: h0..h7, feature, hv, one f16 per value) -, the PE / PE(dir) operands (split: saved like every other operand, 3
# bytes per value; split_f16bwd: f32 rows), one d_raw row)
DW_BYTES_PER_POINT = {"split": 3 * ((8 * 256 + 128) + (8 * 256 + 128)) + 3 * (64 + 32) + 8,
                      "split_f16bwd": 2 * ((8 * 256 + 256 + 128) + (8 * 256 + 256 + 128)) + 4 * (64 + 32) + 8}
DTYPE_NOTE = {
    "split": "f32 storage; every MLP GEMM (forward, dX, dW) as 3 f16 MFMAs on hi/lo-split operands (22-bit in flight, 19-bit saved "
             "operands), f32 accumulate: fp32-equivalent, measured against float64 (tests/test_f64_truth_gpu.py)",
    "split_f16bwd": "f32 storage; forward 3 f16 MFMAs on hi/lo-split operands (22-bit); REDUCED-PRECISION backward: f16 gradient x "
                    "hi/lo weight (dX), f16 x f16 (dW), f32 accumulate",
    "f32": "f32 (v_mfma_f32_32x32x2_f32: bit-exact f32 products, f32 accumulate)"}


def algorithmic_bytes_per_step(wl, C):
    """SURVEY 8d 'Algorithmic bytes': what ANY implementation must move per step - weights read in forward and
    backward, gradients written, Adam state read + written, per-ray inputs and outputs.  Saved activations are design
    traffic, not algorithmic."""
    n_w = sum(o * i + o for o, i in ((256, 63), (256, 256), (256, 256), (256, 256), (256, 256), (256, 319), (256, 256),
                                     (256, 256), (128, 283), (256, 256), (1, 256), (C, 128)))
    weights = 2 * n_w * 4 * (2 + 1 + 6)                       # fwd read + bwd read, grad write, m/v/p read + write
    rays = 2 * wl["Re"] + wl["n"] * wl["Rr"]
    per_ray = 8 + (2 * C + 4) * 4                               # pixel index in, rgb_map/rgb0 + disp/acc x 2 out (sigma skipped in training)
    return weights + rays * per_ray


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="weak: every rank renders the workload's batch; strong: the workload's batch is the GLOBAL batch.  Default: "
                         "weak for C1-C3, strong for C4 / C5 (BASELINE.json quotes those as 8192 rays over 8 GPUs, SURVEY 8e)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend: nccl = RCCL over xGMI (the product path); gloo only with --selfcheck-only (CPU test)")
    ap.add_argument("--selfcheck-only", action="store_true",
                    help="rendezvous + collective self-check only (ranks seen, all-reduce time), no training step")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="development: every rank uses device 0 and the gloo transport (the whole multi-rank flow of this script on a "
                         "one-GPU box; the number it prints is not a measurement)")
    ap.add_argument("--rccl-loopback", action="store_true",
                    help="one GPU, but every collective of the step is issued on a one-rank RCCL communicator (a sum over one rank is "
                         "the identity): the fixed per-step cost of the data-parallel plumbing - RCCL launches, stream hand-overs - "
                         "without wire time")
    ap.add_argument("--batch-fraction", type=int, default=1,
                    help="render 1/F of the workload's pixels per rank (F = 8 on one GPU: the per-rank step of a strong-scaled 8-GPU run)")
    ap.add_argument("--proxy-rank", type=int, default=-1,
                    help="with --batch-fraction F: WHICH rank's share of the F-way ray-balanced split (dist.balanced_shard_bounds) this one "
                         "GPU renders.  -1 (default): the rank with the most rays - the one the step of a strong-scaled job waits for")
    ap.add_argument("--pixel-shards", action="store_true",
                    help="with --batch-fraction F: the round-4 split instead (event and blur pixels each dealt evenly, left-overs to the low "
                         "ranks; rank 0's share) - for the A/B in profiles/r05_one_eighth_batch_same_box.log")
    ap.add_argument("--event-bins", type=int, default=1,
                    help="dense event bins (BASELINE.json configs[4]; an extension, the reference has one bin per step): the event "
                         "window is cut into B contiguous equal bins, the event pixels are rendered at the B + 1 bin boundaries, every "
                         "bin contributes the reference's event-loss term on its pose pair (engine.TrainStep(event_bins=B))")
    ap.add_argument("--n-events", type=int, default=2_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", default="sweep",
                    help="host threads of the cpu_baseline leg: 'sweep' (default: one warm full-size step at 16, 64 and the physical core "
                         "count, the best is timed and reported with the sweep) or a number")
    ap.add_argument("--primary-only", action="store_true", help="timed training steps only (profiling runs): no secondary legs")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--mlp-precision", default="split", choices=["f32", "split", "split_f16bwd"],
                    help="arithmetic of the fused MLP kernels (include/benerf_hip.h, K3 `precision`)")
    a = ap.parse_args()
    if a.scaling is None:
        a.scaling = "strong" if a.workload in ("C4", "C5") else "weak"
    return a


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: re-run this script as N ranks under torch.distributed.run
    (one process per GPU, rendezvous on 127.0.0.1) and pass its exit code on.  Refuses when the node has fewer devices."""
    import socket
    import subprocess
    if a.backend == "nccl" and not a.oversubscribe and torch.cuda.device_count() < a.gpus:
        print("bench.py: --gpus %d but only %d device(s) visible" % (a.gpus, torch.cuda.device_count()), file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def collective_selfcheck(world, device, n_net):
    """Every rank contributes a vector of ones: the sum must equal the world size everywhere (the process group really spans
    `world` ranks); then the step's three gradient buckets (fine net, coarse net, trajectory) are all-reduced 20 times,
    timed with events on the stream that waits for them."""
    ones = torch.ones(64, dtype=torch.float32, device=device)
    torch.distributed.all_reduce(ones)
    seen = int(round(float(ones.min().item())))
    assert float(ones.max().item()) == float(ones.min().item()) == world, "all-reduce of ones gave %r on a world of %d" % (ones.tolist()[:4], world)
    buckets = [torch.zeros(n, dtype=torch.float32, device=device) for n in (n_net, n_net, 31)]
    cuda = device.type == "cuda"

    def once():
        hs = [torch.distributed.all_reduce(b, async_op=True) for b in buckets]
        for h in hs:
            h.wait()
    for _ in range(3):
        once()
    if cuda:
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    t0 = time.perf_counter()
    for _ in range(20):
        once()
    if cuda:
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
    else:
        ms = (time.perf_counter() - t0) * 1e3 / 20
    nbytes = sum(b.numel() for b in buckets) * 4
    return {"rccl_ranks_seen": seen, "allreduce_ms": round(ms, 4), "allreduce_bytes": nbytes,
            "allreduce_bus_gbs": round(2 * (world - 1) / world * nbytes / (ms * 1e-3) / 1e9, 2)}


def power_leg(one_step, device_index, seconds):
    """Average / peak socket power, power cap and shader clock over `seconds` of back-to-back steps (None without amdsmi)."""
    try:
        import threading
        import amdsmi
        amdsmi.amdsmi_init()
        h = amdsmi.amdsmi_get_processor_handles()[device_index]
        cap = amdsmi.amdsmi_get_power_cap_info(h).get("power_cap")
    except Exception:      # noqa: BLE001 - diagnostics only
        return None
    rows, stop = [], [False]

    def sample():
        while not stop[0]:
            try:
                p = amdsmi.amdsmi_get_power_info(h)
                c = amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX)
                rows.append((p.get("current_socket_power", p.get("socket_power")), c.get("clk")))
            except Exception:      # noqa: BLE001
                pass
            time.sleep(0.01)
    for _ in range(5):
        one_step()
    torch.cuda.synchronize()
    th = threading.Thread(target=sample, daemon=True)
    th.start()
    t0 = time.perf_counter()
    n_steps = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(10):
            one_step()
        n_steps += 10
        torch.cuda.synchronize()
    sustained_ms = (time.perf_counter() - t0) / n_steps * 1e3
    stop[0] = True
    th.join()
    pw = [float(r[0]) for r in rows if isinstance(r[0], (int, float))]
    ck = [float(r[1]) for r in rows if isinstance(r[1], (int, float))]
    if not pw:
        return None
    return {"cap_w": round(cap / 1e6, 1) if isinstance(cap, (int, float)) else None, "avg_w": round(sum(pw) / len(pw), 1), "max_w": max(pw),
            "gfx_mhz_avg": round(sum(ck) / len(ck), 1) if ck else None, "samples": len(pw),
            "ms_per_step_sustained": round(sustained_ms, 3), "steps_sustained": n_steps,
            "note": "socket power / shader clock sampled every 10 ms over back-to-back training steps (small-kernel phases included)"}


def hbm_traffic_leg(a, timeout_s=120):
    """HBM bytes of the K3 launches, measured in THIS run: rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, one pass each, with
    --kernel-trace only - the combination the GPU guide prescribes) over two training steps of this very command
    (`bench.py --primary-only`, same workload / batch / arithmetic mode) in a child process, reduced like tools/summarize_profile.py
    does for the committed summaries: per launch = sum over the launch's kernels, averaged over the coarse and the fine launch;
    FETCH_SIZE x 2 (gfx950 counts 64 B per 128-B request on wide streaming reads: MI355X_MICROARCH.md) and x 1024 (KB units).
    Returns ({group: {"hbm_read_bytes_per_launch", "hbm_write_bytes_per_launch"}}, seconds) or (None, reason)."""
    import collections
    import csv
    import glob
    import importlib.util
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if rp is None:
        return None, "rocprofv3 not found"
    spec = importlib.util.spec_from_file_location("summarize_profile", os.path.join(ROOT, "tools", "summarize_profile.py"))
    sp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sp)
    t0 = time.perf_counter()
    tmp = tempfile.mkdtemp(prefix="benerf_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", BENERF_BENCH_NO_PMC="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--primary-only", "--no-cpu-baseline", "--steps", "2", "--warmup", "1",
           "--workload", a.workload, "--batch-fraction", str(a.batch_fraction), "--mlp-precision", a.mlp_precision, "--seed", str(a.seed),
           "--n-events", str(a.n_events), "--event-bins", str(a.event_bins)]
    tot = collections.defaultdict(lambda: collections.defaultdict(list))      # (group, part) -> counter -> per-dispatch values
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out_dir = os.path.join(tmp, counter)
            r = subprocess.run([rp, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out_dir, "-o", "p", "--"] + cmd,
                               cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
            hits = glob.glob(os.path.join(out_dir, "**", "p_counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not hits:
                return None, "rocprofv3 --pmc %s failed (rc %d)" % (counter, r.returncode)
            for row in csv.DictReader(open(hits[0])):
                c = sp.classify(row["Kernel_Name"])
                if c and row["Counter_Name"] == counter:
                    tot[c][counter].append(float(row["Counter_Value"]))
    except (subprocess.TimeoutExpired, OSError, KeyError, ValueError) as e:
        return None, "PMC leg: %s" % type(e).__name__
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    res = {}
    for g in ("mlp_fwd", "mlp_bwd_dx", "mlp_bwd_dw"):
        rd = sum(sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]) for c, v in tot.items() if c[0] == g and v.get("FETCH_SIZE"))
        wr = sum(sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"]) for c, v in tot.items() if c[0] == g and v.get("WRITE_SIZE"))
        if rd == 0 and wr == 0:
            return None, "no %s dispatches in the PMC passes" % g
        res[g] = {"hbm_read_bytes_per_launch": 2 * rd * 1024, "hbm_write_bytes_per_launch": wr * 1024}
    return res, time.perf_counter() - t0


def dropin_leg(a, timeout_s=240):
    """The path an UNCHANGED train.py runs: tools/dropin_driver.py - train.py:153-394 restated with the reference's names (graph.forward,
    the loss lines on torch tensors, every logger.write(.item()), loss.backward(), the optimizer_*.step() calls, the five learning-rate
    updates) on the modules `benerf_amd.dropin.install()` registers - at the same workload and arithmetic mode, in a child process (the
    driver switches torch's default tensor type to cuda like train.py:472 and aliases top-level module names): >= 20 timed
    iterations after warm-up, wall clock around them, HIP-event durations of the K3 launches.  A second child under
    `rocprofv3 --kernel-trace` (6 iterations) gives launches per iteration and where the device idles (tools/dropin_timeline.py)."""
    import glob
    import shutil
    import subprocess
    import tempfile
    drv = os.path.join(ROOT, "tools", "dropin_driver.py")
    cmd = [sys.executable, drv, "--workload", a.workload, "--mlp-precision", a.mlp_precision, "--timers"]
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        r = subprocess.run(cmd + ["--steps", str(max(20, a.steps)), "--warmup", str(max(5, a.warmup))], env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True, timeout=timeout_s)
        if r.returncode != 0:
            return {"error": "tools/dropin_driver.py failed (rc %d): %s" % (r.returncode, r.stderr.strip()[-300:])}
        leg = json.loads(r.stdout.strip().splitlines()[-1])
    except (subprocess.TimeoutExpired, OSError, ValueError, IndexError) as e:
        return {"error": "dropin leg: %r" % (e,)}
    out = {"value": leg["rays_per_s"], "unit": "rays/s", "ms_per_step": leg["ms_per_step"], "median_ms_per_step": leg["median_ms_per_step"],
           "steps": leg["steps"], "warmup": leg["warmup"], "rays_per_step": leg["rays_per_step"], "final_loss": leg["final_loss"],
           "host_syncs_per_step": leg["host_syncs_per_step"], "per_kernel": leg.get("per_kernel"),
           "what": "train.py:153-394 as the reference writes it (tools/dropin_driver.py) on the drop-in modules: Graph.forward -> torch loss "
                   "lines + logger .item() reads -> loss.backward() -> optimizer_*.step() -> learning-rate updates; same workload, same "
                   "arithmetic mode as `value`"}
    rp = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    under_profiler = any(k.startswith(("ROCPROF", "ROCPROFILER", "ROCP_")) for k in os.environ)
    if rp is not None and not under_profiler and not os.environ.get("BENERF_BENCH_NO_PMC"):
        tmp = tempfile.mkdtemp(prefix="benerf_dropin_", dir="/tmp")
        try:
            r = subprocess.run([rp, "--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "t", "--"] + cmd[:-1] + ["--steps", "6", "--warmup", "3"],
                               cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
            hits = glob.glob(os.path.join(tmp, "**", "t_kernel_trace.csv"), recursive=True)
            if r.returncode == 0 and hits:
                t = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dropin_timeline.py"), hits[0], "--json"], stdout=subprocess.PIPE,
                                   text=True, timeout=60)
                tl = json.loads(t.stdout)
                out["launches_per_step"] = tl["launches_per_step"]
                out["timeline_profiled"] = {"span_ms": tl["span_ms"], "sections": tl["sections"],
                                            "note": "one iteration under rocprofv3 --kernel-trace (slower than un-profiled by ~4 us per launch); "
                                                    "`loss` = train.py's own loss lines + their autograd backward, between the render's last "
                                                    "forward and first backward launch: device idle there is host time of the reference's code"}
        except (subprocess.TimeoutExpired, OSError, ValueError, KeyError) as e:
            out["timeline_profiled"] = {"error": repr(e)[:200]}
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return out


def split_mode(a):
    return a.mlp_precision != "f32"


def build_graph(args_ns, device, seed):
    """Reference-shaped graph with the reference's own initialisation: Xavier-uniform weights, zero biases
    (run_nerf_helpers.init_nerf, applied at iteration 0 by train.py:154-157), knots rand * 0.01 (model/optimize.py:22-24)."""
    from benerf_amd.model import optimize
    from benerf_amd import run_nerf_helpers
    torch.manual_seed(seed)
    model = optimize.Model(args_ns)
    model.graph.to(device)
    g = model.build_network(args_ns)
    run_nerf_helpers.init_nerf(g.nerf)
    run_nerf_helpers.init_nerf(g.nerf_fine)
    return g


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def physical_cores():
    """Physical cores of the host (distinct (package, core) pairs of /proc/cpuinfo); os.cpu_count() if that cannot be read."""
    try:
        seen, pkg = set(), None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pkg = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                seen.add((pkg, line.split(":", 1)[1].strip()))
        if seen:
            return len(seen)
    except OSError:
        pass
    return os.cpu_count() or 1


def cpu_baseline(wl, seed, n_steps, warm_fraction=1, threads="sweep"):
    """Oracle training step (torch CPU) at the FULL size of workload `wl`: `n_steps` timed steps after one warm-up step
    (the warm-up runs on 1/warm_fraction of the pixels when a full step takes many seconds).
    threads: a number, or "sweep" - after the warm-up ONE full-size step is timed at each of {16, 64, physical cores} host threads
    (those the host has), the timed steps then run at the best count; `cores` = that count, `sweep` = what each one measured
    ("using all host cores" - the honest baseline is the best the host does, not a guess)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import benerf_oracle as O
    import golden_inputs as GI
    from benerf_amd import workloads as WL
    w = WL.WORKLOADS[wl]
    cam = WL.CAMERAS[w["cam"]]
    C, S, Ni, P = w["channels"], w["S"], w["Ni"], w["n"]
    rng = np.random.default_rng(seed)
    cfg = O.StepConfig(H=cam["H"], W=cam["W"], fx=cam["fx"], fy=cam["fy"], cx=cam["cx"], cy=cam["cy"], channels=C,
                       n_samples=S, n_importance=Ni, n_poses=P, dataset=w["dataset"], threshold=w["threshold"],
                       window=w["window"])
    pc = {k: v.requires_grad_(True) for k, v in O.xavier_params(rng, C).items()}
    pf = {k: v.requires_grad_(True) for k, v in O.xavier_params(rng, C).items()}
    knots = GI.knots_init(rng).requires_grad_(True)
    tr = torch.zeros(1, 6, requires_grad=True)
    params = list(pc.values()) + list(pf.values()) + [knots]
    state = [(torch.zeros_like(p), torch.zeros_like(p)) for p in params]
    ev = GI.synthetic_events(rng, cam, 200000)
    img = torch.from_numpy(rng.random((cam["H"] * cam["W"], C)).astype(np.float32))
    counter = [0]

    def one(frac):
        it = counter[0]
        counter[0] += 1
        Re, Rr = max(w["Re"] // frac, 8), max(w["Rr"] // frac, 1)
        t0 = time.perf_counter()
        low_t = float(rng.random() * (1 - w["window"]))
        sel, up = O.event_window(ev["ts"], low_t, w["window"])
        acc = O.accumulate_events(cam["H"], cam["W"], ev["x"][sel], ev["y"][sel], ev["pol"][sel])
        idx_e, idx_r = GI.pixel_indices(rng, cam, Re), GI.pixel_indices(rng, cam, Rr)
        d_e = GI.render_draws(rng, 2 * Re, S, Ni)
        d_r = GI.render_draws(rng, P * Rr, S, Ni)
        loss, _ = O.step_loss(cfg, pc, pf, knots, tr, torch.tensor([low_t, up], dtype=torch.float32),
                              torch.tensor([0.0, 1.0]), idx_e, idx_r, acc.reshape(-1, 1)[idx_e], img[idx_r], d_e, d_r)
        for p in params:
            p.grad = None
        loss.backward()
        with torch.no_grad():
            for p, (m, v) in zip(params, state):
                O.adam_update(p, p.grad, m, v, it + 1, 5e-4)
        return time.perf_counter() - t0

    host = os.cpu_count() or 1
    sweep = None
    if threads == "sweep":
        cands = sorted({min(c, host) for c in (16, 64, physical_cores())})
        torch.set_num_threads(cands[0])
        one(warm_fraction)                       # warm-up: allocator, thread pool
        sweep = {}
        for c in cands:
            torch.set_num_threads(c)
            sweep[str(c)] = round(one(1), 3)
        best = min(sweep, key=sweep.get)
        torch.set_num_threads(int(best))
    else:
        torch.set_num_threads(max(1, min(int(threads), host)))
        one(warm_fraction)
    times = [one(1) for _ in range(n_steps)]
    rays = 2 * w["Re"] + P * w["Rr"]
    out = {"value": round(rays / (sum(times) / len(times)), 1), "unit": "rays/s", "cores": torch.get_num_threads(),
           "host_cores": host, "host_physical_cores": physical_cores(), "cpu": cpu_model(), "kind": "port",
           "s_per_step": round(sum(times) / len(times), 3),
           "sample": "%d full-size step(s) of %s (%d rays/step, %d+%d samples), oracle torch-CPU step incl. backward + Adam"
                     % (n_steps, wl, rays, S, S + Ni)}
    if sweep is not None:
        out["sweep"] = {"s_per_step_by_threads": sweep, "note": "one warm full-size step per thread count; `cores` = the fastest, used for the timed steps"}
    return out


def torch_gpu_baseline(wl, seed, device):
    """The oracle's op-for-op torch step run EAGERLY on the same MI355X (PyTorch-ROCm, hipBLASLt fp32 GEMMs,
    autograd, unfused) at the FULL workload size: the stand-in for "reference single-GPU PyTorch"
     - the reference itself cannot be shipped to the GPU box."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import benerf_oracle as O
    import golden_inputs as GI
    from benerf_amd import workloads as WL
    w = WL.WORKLOADS[wl]
    cam = WL.CAMERAS[w["cam"]]
    Re, Rr, C, S, Ni, P = w["Re"], w["Rr"], w["channels"], w["S"], w["Ni"], w["n"]
    rng = np.random.default_rng(seed)
    prev = torch.get_default_device() if hasattr(torch, "get_default_device") else None
    torch.set_default_device(device)
    try:
        cfg = O.StepConfig(H=cam["H"], W=cam["W"], fx=cam["fx"], fy=cam["fy"], cx=cam["cx"], cy=cam["cy"], channels=C,
                           n_samples=S, n_importance=Ni, n_poses=P, dataset=w["dataset"], threshold=w["threshold"],
                           window=w["window"])
        pc = {k: v.to(device).requires_grad_(True) for k, v in O.xavier_params(rng, C).items()}
        pf = {k: v.to(device).requires_grad_(True) for k, v in O.xavier_params(rng, C).items()}
        knots = GI.knots_init(rng).to(device).requires_grad_(True)
        tr = torch.zeros(1, 6)
        params = list(pc.values()) + list(pf.values()) + [knots]
        state = [(torch.zeros_like(p), torch.zeros_like(p)) for p in params]
        HW = cam["H"] * cam["W"]
        accu = torch.from_numpy(rng.integers(-3, 4, (HW, 1)).astype(np.float64)).to(device)
        img = torch.from_numpy(rng.random((HW, C)).astype(np.float32)).to(device)
        times = []
        n_steps = 5
        for it in range(n_steps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            low_t = float(rng.random() * (1 - w["window"]))
            idx_e = torch.randperm(HW)[:Re]
            idx_r = torch.randperm(HW)[:Rr]
            d_e = {"t_rand": torch.rand(2 * Re, S), "noise0": torch.randn(2 * Re, S), "u": torch.rand(2 * Re, Ni),
                   "noise1": torch.randn(2 * Re, S + Ni)}
            d_r = {"t_rand": torch.rand(P * Rr, S), "noise0": torch.randn(P * Rr, S), "u": torch.rand(P * Rr, Ni),
                   "noise1": torch.randn(P * Rr, S + Ni)}
            loss, _ = O.step_loss(cfg, pc, pf, knots, tr, torch.tensor([low_t, low_t + w["window"]], dtype=torch.float32),
                                  torch.tensor([0.0, 1.0]), idx_e, idx_r, accu[idx_e], img[idx_r], d_e, d_r)
            for p in params:
                p.grad = None
            loss.backward()
            with torch.no_grad():
                for p, (m, v) in zip(params, state):
                    O.adam_update(p, p.grad, m, v, it + 1, 5e-4)
            torch.cuda.synchronize()
            if it > 0:
                times.append(time.perf_counter() - t0)
    finally:
        torch.set_default_device(prev if prev is not None else "cpu")
    rays = 2 * Re + P * Rr
    return {"value": round(rays / (sum(times) / len(times)), 1), "unit": "rays/s", "ms_per_step": round(1e3 * sum(times) / len(times), 2),
            "kind": "port", "sample": "%d full-size %s steps of the oracle's torch step, eager PyTorch-ROCm on the same GPU" % (n_steps, wl)}


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a))
    # The contract is ONE JSON line on stdout.  Communication libraries write to fd 1 themselves (RCCL: "Librccl path : ...",
    # gloo: "[Gloo] Rank 0 is connected to ..."): keep a private handle on the real stdout for the line and point fd 1 at
    # stderr for everybody else in this process.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world != a.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node == --gpus)" % (a.gpus, world))
    if a.oversubscribe:
        a.backend = "gloo"
        local_rank = 0
    if a.backend == "gloo" and not a.oversubscribe:
        if not a.selfcheck_only:
            sys.exit("bench.py: --backend gloo is the CPU self-check of the launch path (--selfcheck-only); the product path is nccl")
        device = torch.device("cpu")
    else:
        if torch.cuda.device_count() <= local_rank:
            sys.exit("bench.py: rank %d has no device (%d visible)" % (local_rank, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    pg = None
    comm = None
    if a.rccl_loopback:
        if world != 1:
            sys.exit("bench.py: --rccl-loopback is a one-GPU mode")
        from benerf_amd import dist as _dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=device)
        pg = torch.distributed.group.WORLD
        _dist.ALWAYS_COMMUNICATE = True
        comm = collective_selfcheck(1, device, 595586)
    if world > 1:
        if a.backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=device)   # RCCL over xGMI
        else:
            torch.distributed.init_process_group("gloo")
        pg = torch.distributed.group.WORLD
        comm_dev = device if a.backend == "nccl" else torch.device("cpu")     # gloo: host buffers
        comm = collective_selfcheck(world, comm_dev, 595586)
    if a.selfcheck_only:
        if rank == 0:
            print(json.dumps({"metric": "collective self-check", "n_gpus": world, "backend": a.backend, "scaling": a.scaling,
                              "config": {"workload": a.workload}, **(comm or {"rccl_ranks_seen": 1})}), file=json_out, flush=True)
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    from benerf_amd import engine, workloads as WL, kernels as K
    K.set_mlp_precision(a.mlp_precision)
    from benerf_amd import dist as D
    wl = dict(WL.WORKLOADS[a.workload])
    wl["bins"] = a.event_bins
    # Pixel counts.  weak: the workload's batch per rank (global = world x that); strong: the workload's batch IS the global batch
    # (C4 / C5 are 8192 rays in total) and every rank renders its contiguous share - the left-over pixels of a batch
    # the ranks cannot split evenly (C4: 215 blur pixels over 8 ranks) go one each to the low ranks (dist.shard_bounds), nothing is
    # dropped; the event pixels are dealt so that every rank renders the same number of rays +- a pixel's worth
    # (dist.balanced_shard_bounds).  --batch-fraction F (one-GPU proxy of a strong-scaled rank): the share of the rank with the
    # MOST rays of an F-way split (or of --proxy-rank).
    pe_rays, pn_rays = a.event_bins + 1, wl["n"]
    def shares(ne, nr, parts):
        return [(e1 - e0, r1 - r0) for (e0, e1), (r0, r1) in D.balanced_shard_bounds(ne, nr, pe_rays, pn_rays, parts)]
    if a.pixel_shards:
        Re_n, Rr_n = (max(D.shard_bounds(n, 0, a.batch_fraction, uneven=True)[1], 1) for n in (wl["Re"], wl["Rr"]))
    else:
        sh = shares(wl["Re"], wl["Rr"], a.batch_fraction)
        k = a.proxy_rank if a.proxy_rank >= 0 else max(range(len(sh)), key=lambda i: (pe_rays * sh[i][0] + pn_rays * sh[i][1], -i))
        Re_n, Rr_n = (max(v, 1) for v in sh[min(k, len(sh) - 1)])
    if a.scaling == "strong":
        Re_g, Rr_g = Re_n, Rr_n
        wl["Re"], wl["Rr"] = shares(Re_g, Rr_g, world)[rank]
        if min(wl["Re"], wl["Rr"]) < 1:
            sys.exit("bench.py: rank %d would render no pixels (%d event / %d blur pixels over %d ranks)" % (rank, Re_g, Rr_g, world))
    else:
        wl["Re"], wl["Rr"] = Re_n, Rr_n
        Re_g, Rr_g = Re_n * world, Rr_n * world
    rays_global = (a.event_bins + 1) * Re_g + wl["n"] * Rr_g
    cam = WL.CAMERAS[wl["cam"]]
    args_ns = WL.make_args(wl)
    g = build_graph(args_ns, device, a.seed)     # identical on every rank (same seed)
    cam_o = engine.Camera(cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    step = engine.TrainStep(g, args_ns, cam_o, cam_o, device, world_size=world, rank=rank, process_group=pg, seed=a.seed,
                            event_bins=a.event_bins, uneven_shards=a.scaling == "strong")
    if world > 1 or a.rccl_loopback:
        step.wait_events = []                    # HIP-event brackets around every wait for a gradient bucket (engine._timed_wait)

    # ---- synthetic inputs, resident in HBM ------------------------------------------------------------
    rng = np.random.default_rng(a.seed)
    HW = cam["H"] * cam["W"]
    ev_x = torch.from_numpy(rng.integers(0, cam["W"], a.n_events).astype(np.int32)).to(device)
    ev_y = torch.from_numpy(rng.integers(0, cam["H"], a.n_events).astype(np.int32)).to(device)
    ev_t = torch.from_numpy(np.sort(rng.random(a.n_events))).to(device)
    ev_p = torch.from_numpy((rng.integers(0, 2, a.n_events) * 2 - 1).astype(np.float32)).to(device)
    image = torch.from_numpy(rng.random((HW, wl["channels"])).astype(np.float32)).to(device)
    rgb_ts = torch.tensor([0.0, 1.0], device=device)
    gen = torch.Generator(device=device)
    gen.manual_seed(a.seed + 1234)               # same pixel draws on every rank, sharded by TrainStep
    accu = torch.zeros((a.event_bins, cam["H"], cam["W"]), dtype=torch.float32, device=device)

    # Inputs of a step - event-window accumulation (K7), window times, pixel draws - are a data loader's job: step k + 1's are
    # prepared inside step k, in TrainStep.step's `overlap` slot (main stream, behind the last backward launch, while the
    # weight-gradient launches of the side stream finish) - like any input pipeline; everything stays inside the timed
    # region.  Same stream as the step's own reads of the event image, so one buffer is enough.  The pixel draws are a keyed
    # bijection (np.random.choice(..., replace=False) in train.py), identical on every rank; TrainStep shards them.
    queue = []

    def prepare(k):
        low_t = float(rng.random() * (1 - wl["window"]))
        up_t = low_t + wl["window"]
        accu.zero_()
        if a.event_bins == 1:
            K.event_window_accumulate(ev_x, ev_y, ev_p, ev_t, low_t, up_t, cam["H"], cam["W"], out=accu[0])
        else:       # one K7 pass per bin over the device-resident sorted stream; bin edges = the f32 linspace the poses are evaluated at
            for b, (lo_b, up_b) in enumerate(K.event_bin_windows(low_t, up_t, a.event_bins)):      # interior bins half-open: no event counted twice
                K.event_window_accumulate(ev_x, ev_y, ev_p, ev_t, lo_b, up_b, cam["H"], cam["W"], out=accu[b])
        evt_ts = torch.full((2,), low_t, dtype=torch.float32, device=device)      # scalars by value: no host buffer to keep alive
        evt_ts[1:].fill_(up_t)
        idx_e = K.sample_pixels(HW, Re_g, a.seed + 1234, 2 * k + 2, device)
        idx_r = K.sample_pixels(HW, Rr_g, a.seed + 1234, 2 * k + 3, device)
        queue.append((k, evt_ts, idx_e, idx_r))
        return evt_ts, rgb_ts, idx_e, idx_r       # TrainStep sets up the poses / rays / depths of these in the same slack

    def one_step():
        if not queue:
            prepare(0)
        k, evt_ts, idx_e, idx_r = queue.pop(0)
        return step.step(evt_ts, rgb_ts, idx_e, idx_r, accu.view(a.event_bins, -1) if a.event_bins > 1 else accu.view(-1), image,
                         overlap=lambda: prepare(k + 1))

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # warm-up includes the HIP-event brackets: the first timing event of a process switches the HSA queue to profiling
    # mode, a one-off stall of ~40 ms that otherwise lands in the first timed step
    K.TIMERS.enabled = True
    for _ in range(a.warmup):
        one_step()
    step.check_range()                           # warm-up steps stayed inside the f16 range (synchronises; untimed)
    K.TIMERS.records.clear()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    sync()
    t0 = time.perf_counter()
    marks[0].record()
    # HIP-event brackets around the six K3 launches (the roofline's per-kernel durations, measured live in the timed region) on every
    # TIMER_EVERY-th timed step: each bracket is two timestamp packets in the stream, ~0.1 ms per fully instrumented step
    # (profiles/r06_timer_overhead.log) - instrumentation, not workload
    timer_every = max(1, int(os.environ.get("BENERF_BENCH_TIMER_EVERY", "4")))
    n_sampled = 0
    for i in range(a.steps):
        K.TIMERS.enabled = i % timer_every == 0
        n_sampled += int(K.TIMERS.enabled)
        losses = one_step()
        marks[i + 1].record()
    sync()
    dt = time.perf_counter() - t0
    K.TIMERS.enabled = False
    step.check_range()
    step_series = [marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps)]
    if os.environ.get("BENERF_BENCH_DEBUG"):
        print("step ms:", " ".join("%.2f" % t for t in step_series), file=sys.stderr)
        print("fwd launch ms:", " ".join("%.2f" % x[2].elapsed_time(x[3]) for x in K.TIMERS.records if x[0] == "mlp_fwd"), file=sys.stderr)
    step_ms = sorted(step_series)
    median_ms = step_ms[len(step_ms) // 2]
    per_rank = None
    if world > 1:
        cdev = device if a.backend == "nccl" else "cpu"
        # every rank's own wall time and median step (HIP events) next to the MAX the metric uses: a straggler shows as a spread
        mine = torch.tensor([dt / a.steps * 1e3, median_ms, float(WL.rays_per_step(wl))], dtype=torch.float64, device=cdev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(allr, mine)
        rows = [x.tolist() for x in allr]
        per_rank = {"ms_per_step_min": round(min(r[0] for r in rows), 3), "ms_per_step_max": round(max(r[0] for r in rows), 3),
                    "median_ms_min": round(min(r[1] for r in rows), 3), "median_ms_max": round(max(r[1] for r in rows), 3),
                    "ms_per_step_by_rank": [round(r[0], 3) for r in rows], "rays_per_step_by_rank": [int(r[2]) for r in rows]}
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    # how long the step's streams actually stall on each gradient bucket (HIP events around the waits, rank 0; warm-up excluded)
    bucket_wait = None
    if step.wait_events:
        ev_ = step.wait_events[-3 * a.steps:]
        bucket_wait = {}
        for nm in ("fine_net", "coarse_net", "trajectory"):
            ts_ = [e0.elapsed_time(e1) for n_, e0, e1 in ev_ if n_ == nm]
            if ts_:
                bucket_wait[nm] = {"avg_ms": round(sum(ts_) / len(ts_), 4), "max_ms": round(max(ts_), 4)}
        bucket_wait["note"] = ("HIP events around the wait for each all-reduce bucket inside the timed steps: fine_net / coarse_net on the "
                               "weight-gradient stream (issued as soon as each network's dW reduce is done), trajectory (+ range-guard "
                               "verdict) on the main stream before Adam; `allreduce_ms` is the three buckets alone, back to back")
        step.wait_events = None

    # ---- secondary: forward-only (inference) rays/s: (i) the training ray batch without activation saving, (ii) chunks of
    # args.chunk = 4096 rays of ONE pose, as Graph.render_video / render_image_test call it (model/nerf.py:353-390) - SURVEY 8d's
    # secondary metric, with its own HIP-event per-launch durations -> `roofline_inference` ------------------------------------
    summ_main = K.TIMERS.summary()         # per-launch HIP-event durations of the timed training steps (before any other leg)
    infer = None
    roof_inf = None
    if world == 1 and not a.primary_only:
        with torch.no_grad():
            idx_e = torch.randperm(HW, device=device, generator=gen)[:wl["Re"]]
            poses = K.spline_poses_fwd(step.knots, None, torch.tensor([0.2, 0.3], device=device), 2, 0)
            idx_all = torch.randperm(HW, device=device, generator=gen)[:(2 * wl["Re"] + wl["n"] * wl["Rr"]) // 2]
            d_inf = engine.Draws(seed=a.seed, offset=12345)
            for i in range(a.steps + 2):
                if i == 2:
                    torch.cuda.synchronize()
                    ti = time.perf_counter()
                engine._render_forward(cam_o, True, wl["S"], wl["Ni"], d_inf, poses, idx_all, step.net_c.packed, step.net_f.packed, False)
            torch.cuda.synchronize()
            n_inf = 2 * idx_all.shape[0]
            infer = round(n_inf / ((time.perf_counter() - ti) / a.steps), 1)
            # (ii) render_video's call shape
            chunk = int(getattr(args_ns, "chunk", 4096))
            pose1 = poses[:1].contiguous()
            idx_c = torch.randperm(HW, device=device, generator=gen)[:chunk]
            K.TIMERS.records.clear()
            n_chunks = max(a.steps, 20)
            for i in range(n_chunks + 3):
                if i == 3:
                    torch.cuda.synchronize()
                    K.TIMERS.records.clear()
                    K.TIMERS.enabled = True
                    ti = time.perf_counter()
                engine._render_forward(cam_o, True, wl["S"], wl["Ni"], d_inf, pose1, idx_c, step.net_c.packed, step.net_f.packed, False)
            torch.cuda.synchronize()
            dt_c = (time.perf_counter() - ti) / n_chunks
            K.TIMERS.enabled = False
            n_l, ms_l, pts_l = K.TIMERS.summary().get("mlp_fwd", (0, 0.0, 0))
            K.TIMERS.records.clear()
            # (iii) a FRAME as the unit of inference: Graph.render_video on the workload's camera (what test.py:112-135 and
            # train.py:404-441 call through render_image_test / render_video_test) - H x W rays in chunks of args.chunk, the four draws
            # per chunk from torch's generator like the reference, outputs concatenated into [H, W, ...] tensors
            frame = None
            try:
                Kmat = np.array([[cam["fx"], 0, cam["cx"]], [0, cam["fy"], cam["cy"]], [0, 0, 1]], dtype=np.float32)
                g.render_video(0, pose1, cam["H"], cam["W"], Kmat, args_ns, np.array([]), type="rgb")      # warm-up frame
                torch.cuda.synchronize()
                n_frames = 3
                tf0 = time.perf_counter()
                for _ in range(n_frames):
                    fr = g.render_video(0, pose1, cam["H"], cam["W"], Kmat, args_ns, np.array([]), type="rgb")
                torch.cuda.synchronize()
                dt_f = (time.perf_counter() - tf0) / n_frames
                n_ch = (HW + chunk - 1) // chunk
                frame = {"frame_ms": round(dt_f * 1e3, 2), "frames_per_s": round(1.0 / dt_f, 2), "frame": "%dx%d" % (cam["H"], cam["W"]),
                         "rays_per_frame": HW, "chunks_per_frame": n_ch, "frame_rays_per_s": round(HW / dt_f, 1),
                         "launches_per_chunk": "4 draws (torch.rand / randn, the reference's) + rays_fwd, stratified_z, mlp_fwd (coarse; split + "
                                               "the normally empty exact-f32 fallback), composite_fwd, sample_pdf_merge, mlp_fwd (fine; "
                                               "split + fallback), composite_fwd = 13; then one torch.cat per output map and frame",
                         "sustained_note": "three whole frames back to back (0.7 s of forward launches at the board's power cap): the "
                                           "SUSTAINED rate; `rays_per_s` / `avg_launch_ms` above come from a 60-ms burst of chunks behind an idle "
                                           "device, which the power controller lets run ~25 % faster (profiles/r06_idle_gap_probe.log)",
                         "shape_ok": list(fr["rgb_map"].shape) == [cam["H"], cam["W"], wl["channels"]]}
                del fr
            except Exception as e:      # noqa: BLE001 - informational
                frame = {"error": repr(e)[:200]}
            if n_l:
                fpp_i = WL.mlp_flops_per_point(wl["channels"])
                pk_i = F16_MFMA_PEAK_TFLOPS if split_mode(a) else F32_MFMA_PEAK_TFLOPS
                tf_i = pts_l * fpp_i / (ms_l * 1e-3) / 1e12
                ex_i = 3 if split_mode(a) else 1
                roof_inf = {"bound": "mfma", "kernel": "mlp_fwd (inference launch: no activation saving)", "achieved": round(tf_i, 2), "peak": pk_i,
                            "unit": "TFLOP/s", "frac": round(tf_i / pk_i, 4), "frac_executed": round(ex_i * tf_i / pk_i, 4),
                            "avg_launch_ms": round(ms_l / n_l, 4), "points_per_launch": int(pts_l / n_l), "launches": n_l,
                            "rays_per_s": round(chunk / dt_c, 1), "ms_per_chunk": round(dt_c * 1e3, 3), "chunk_rays": chunk,
                            "mlp_ms_per_chunk": round(ms_l / n_chunks, 3),
                            "note": "Graph.render_video's call shape: chunks of args.chunk rays of one pose through the whole forward "
                                    "(rays, coarse MLP, compositing, sample_pdf, fine MLP, compositing); HIP events around the two MLP "
                                    "launches of a chunk (coarse %d + fine %d samples per ray), averaged; in the split mode an inference "
                                    "launch is BENERF_MLP_AUTO: the split launch + the normally empty exact-f32 fallback launch"
                                    % (wl["S"], wl["S"] + wl["Ni"])}
                if frame is not None:
                    roof_inf.update(frame)

    # ---- secondary: the same training step with exact-f32 MFMA products (the strict arithmetic mode), >= 20 timed steps,
    # its own per-kernel HIP-event durations -> `exact_f32` + `roofline_f32` in the JSON line --------------------------------
    # ---- secondary: board power and shader clock while the step runs (amdsmi, sampled from a thread for ~1.5 s of extra steps):
    # the K3 kernels sit at the board's power cap, the clock the roofline's 2.4 GHz peak assumes is not available to them ------
    power = None
    if world == 1 and not a.primary_only:
        power = power_leg(one_step, device.index or 0, 1.5)
    exact = roof_f32 = reduced = None

    def other_mode_leg(mode, peak_tf):
        """the same training step in another arithmetic mode: >= 20 timed steps, its own per-kernel HIP-event durations"""
        K.set_mlp_precision(mode)
        K.TIMERS.records.clear()
        K.TIMERS.enabled = True
        for _ in range(3):
            one_step()
        K.TIMERS.records.clear()
        torch.cuda.synchronize()
        to = time.perf_counter()
        n_other = max(20, a.steps)
        for _ in range(n_other):
            one_step()
        torch.cuda.synchronize()
        dt_o = (time.perf_counter() - to) / n_other
        K.TIMERS.enabled = False
        step.check_range()
        leg = {"value": round(WL.rays_per_step(wl) / dt_o, 1), "unit": "rays/s", "ms_per_step": round(dt_o * 1e3, 3), "steps": n_other,
               "dtype": DTYPE_NOTE[mode]}
        fpp_ = WL.mlp_flops_per_point(wl["channels"])
        per = {}
        for name, (n_, ms_, pts_) in K.TIMERS.summary().items():
            tf_ = pts_ * fpp_ / (ms_ * 1e-3) / 1e12 if ms_ > 0 else 0.0
            per[name] = {"launches": n_, "avg_ms": round(ms_ / n_, 4), "points_per_launch": int(pts_ / n_), "tflops_algorithmic": round(tf_, 2),
                         "frac_of_mfma_peak": round(tf_ / peak_tf, 4)}
        roof_ = None
        if per:
            dom_ = max(per.items(), key=lambda kv: kv[1]["avg_ms"] * kv[1]["launches"])[0]
            pts_step_ = WL.rays_per_step(wl) * (wl["S"] + wl["S"] + wl["Ni"])
            roof_ = {"bound": "mfma", "kernel": dom_, "achieved": per[dom_]["tflops_algorithmic"], "peak": peak_tf,
                     "unit": "TFLOP/s", "frac": per[dom_]["frac_of_mfma_peak"], "avg_launch_ms": per[dom_]["avg_ms"],
                     "step_tflops_algorithmic": round(pts_step_ * fpp_ * 3 / dt_o / 1e12, 2),
                     "step_frac_of_mfma_peak": round(pts_step_ * fpp_ * 3 / dt_o / 1e12 / peak_tf, 4), "per_kernel": per}
        K.TIMERS.records.clear()
        K.set_mlp_precision(a.mlp_precision)
        return leg, roof_

    def kernels_alone(mode, peak_tf, reps=7):
        """each of the three K3 launches of the FINE network by itself (nothing else on the device), median of `reps`: inside the
        step the dW launches share the device with the other network's dX chain (second stream), so their in-step HIP-event
        durations include that time slicing - these do not"""
        K.set_mlp_precision(mode)
        try:
            net = step.net_f.packed
            N, S = WL.rays_per_step(wl), wl["S"] + wl["Ni"]
            g_ = torch.Generator(device=device)
            g_.manual_seed(7)
            ro = torch.rand((N, 3), device=device, generator=g_) * 0.2 - 0.1
            rd = torch.nn.functional.normalize(torch.rand((N, 3), device=device, generator=g_) - 0.5, dim=-1)
            z = torch.sort(torch.rand((N, S), device=device, generator=g_), dim=-1).values
            raw, acts = K.mlp_fwd(net, ro, rd, rd, z, True, status=step.guard.words)
            d_raw = (torch.rand(raw.shape, device=device, generator=g_) - 0.5).view(-1, raw.shape[-1]) * 1e-4
            gw, gb = [torch.zeros_like(w) for w in net.weights], [torch.zeros_like(b) for b in net.biases]
            dacts = [None]

            def med(fn):
                fn()
                ts = []
                for _ in range(reps):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    e0.record()
                    fn()
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1))
                return sorted(ts)[len(ts) // 2]

            def dx():
                dacts[0] = K.mlp_bwd_dx(net, d_raw, acts, N, S, status=step.guard.words)[2]
            t = {"mlp_fwd": med(lambda: K.mlp_fwd(net, ro, rd, rd, z, True, status=step.guard.words)), "mlp_bwd_dx": med(dx),
                 "mlp_bwd_dw": med(lambda: K.mlp_bwd_dw(net, d_raw, acts, dacts[0], N, S, gw, gb, False))}
            from benerf_amd import _lib as L_
            step.guard.words[:L_.ST_SKIPPED_TOTAL].zero_()      # [SKIPPED_TOTAL] stays: Adam's bias correction counts applied steps
            step.guard.words[L_.ST_MAX_CONSECUTIVE:].zero_()
            fpp_ = WL.mlp_flops_per_point(wl["channels"])
            return {k: {"ms": round(v, 4), "points": N * S, "tflops_algorithmic": round(N * S * fpp_ / (v * 1e-3) / 1e12, 2),
                        "frac_of_mfma_peak": round(N * S * fpp_ / (v * 1e-3) / 1e12 / peak_tf, 4)} for k, v in t.items()}
        finally:
            K.set_mlp_precision(a.mlp_precision)

    alone = None
    if world == 1 and not a.primary_only and a.batch_fraction == 1:
        alone = kernels_alone(a.mlp_precision, F16_MFMA_PEAK_TFLOPS if split_mode(a) else F32_MFMA_PEAK_TFLOPS)
    if world == 1 and split_mode(a) and not a.primary_only:
        exact, roof_f32 = other_mode_leg("f32", F32_MFMA_PEAK_TFLOPS)
        if roof_f32 is not None and a.batch_fraction == 1:
            roof_f32["per_kernel_alone"] = kernels_alone("f32", F32_MFMA_PEAK_TFLOPS)
            roof_f32["per_kernel_note"] = ("per_kernel: HIP-event durations inside the step (only the last, coarse dW launch runs on the second "
                                           "stream, beside small kernels); per_kernel_alone: the fine network's launches by themselves")
        if a.mlp_precision == "split":
            # NOT the headline: the opt-in mode whose backward GEMMs take f16 operands (round 3's default) - narrower arithmetic than
            # the reference's fp32, reported for scale only
            reduced, _ = other_mode_leg("split_f16bwd", F16_MFMA_PEAK_TFLOPS)
            reduced["note"] = ("reduced-precision backward (gradients 2-8e-4 of their largest entry from the fp32 result): not the "
                               "reference's precision, not the headline")
    other = exact["value"] if exact else None

    rays_step = rays_global              # what all ranks rendered per step (uneven strong-scaled shards included)
    ms_step = dt / a.steps * 1e3
    value = rays_step / (dt / a.steps)

    # ---- roofline: the fused MLP (K3) against the MFMA roof, algorithmic FLOPs / HIP-event-timed duration --
    fpp = WL.mlp_flops_per_point(wl["channels"])
    summ = summ_main
    split = a.mlp_precision != "f32"
    peak = F16_MFMA_PEAK_TFLOPS if split else F32_MFMA_PEAK_TFLOPS
    roof = None
    kern = {}
    for name, (n, ms, pts) in summ.items():
        # algorithmic flops: fwd = fpp/point; dx chain = fpp/point; dW = fpp/point (training = 3 x fwd)
        tf = pts * fpp / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        ex = EXECUTED_PER_PRODUCT[a.mlp_precision][name] if split else 1
        kern[name] = {"launches": n, "avg_ms": round(ms / n, 4), "points_per_launch": int(pts / n), "tflops_algorithmic": round(tf, 2),
                      "frac_of_mfma_peak": round(tf / peak, 4), "mfma_per_product": ex, "frac_executed": round(ex * tf / peak, 4)}
    pmc = {}
    try:    # per-launch PMC figures of the same command from the committed rocprofv3 passes (profiles/README.md): MFMA utilisation,
        # and the HBM bytes when this run cannot measure them itself
        import glob
        latest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_pmc_summary.json" % a.mlp_precision)))[-1]
        if a.workload == "C2" and world == 1 and a.batch_fraction == 1:
            pmc = json.load(open(latest))["kernels"]
            pmc_src = os.path.relpath(latest, ROOT)
    except (IndexError, KeyError, OSError, ValueError):
        pmc = {}
    # HBM traffic of the K3 launches measured by THIS run (two rocprofv3 --pmc passes over a two-step child run of this command)
    pmc_live_note = None
    # (not when this process is itself running under a profiler: nested counter collection is asking for trouble)
    under_profiler = any(k.startswith(("ROCPROF", "ROCPROFILER", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", "")
    if under_profiler:
        pmc_live_note = "this process runs under a profiler"
    if world == 1 and not a.primary_only and not a.oversubscribe and not os.environ.get("BENERF_BENCH_NO_PMC") and not under_profiler:
        torch.cuda.synchronize()
        live, info = hbm_traffic_leg(a)
        if live is not None:
            for k_, v_ in live.items():
                pmc.setdefault(k_, {}).update(v_)
            pmc_src = "this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace) over 2 steps of `bench.py " \
                      "--primary-only` in a child process, %.0f s" % info
        else:
            pmc_live_note = info

    def traffic_of(k):
        try:
            return int(pmc[k]["hbm_read_bytes_per_launch"] + pmc[k]["hbm_write_bytes_per_launch"])
        except KeyError:
            return None

    if kern:
        dom = max(summ.items(), key=lambda kv: kv[1][1])[0]          # largest share of the timed region
        n_dom, ms_dom, pts_dom = summ[dom]
        roof = {"bound": "mfma", "kernel": dom, "achieved": kern[dom]["tflops_algorithmic"], "peak": peak, "unit": "TFLOP/s",
                "frac": kern[dom]["frac_of_mfma_peak"], "frac_executed": kern[dom]["frac_executed"],
                "traffic": traffic_of(dom), "avg_launch_ms": kern[dom]["avg_ms"],
                "flops_per_point": fpp, "points_per_launch": kern[dom]["points_per_launch"],
                "peak_note": "dense %s MFMA peak at 2.4 GHz; the K3 kernels run at the board's power cap (`power` below, "
                             "profiles/r04_mfma_power_probe.log: a pure MFMA stream is clocked at 1.9 GHz by the 1400 W cap) and are clocked at "
                             "1.7-2.1 GHz: DESIGN.md 4" % ("f16" if split else "f32"),
                "per_kernel_note": "HIP-event durations on the launching stream, bracketed on every %d-th timed step (%d of %d); only the step's LAST weight-gradient launch (coarse network) "
                                   "runs on the second stream, beside the small kernels of the trajectory tail (engine.TrainStep), so the six K3 "
                                   "launches of a step add up to its duration" % (timer_every, n_sampled, a.steps),
                "per_kernel": kern}
        if pmc:
            roof["traffic_source"] = pmc_src
            if pmc_live_note:
                roof["traffic_source"] += " (in-run PMC passes unavailable: %s)" % pmc_live_note
            if "mfma_util" in pmc.get(dom, {}):
                roof["mfma_util_profiled"] = round(pmc[dom]["mfma_util"], 4)
        # whole training step against the same roof: 3 x forward FLOPs per point
        pts_step = WL.rays_per_step(wl) * (wl["S"] + wl["S"] + wl["Ni"])
        roof["step_tflops_algorithmic"] = round(pts_step * fpp * 3 / (dt / a.steps) / 1e12, 2)
        roof["step_frac_of_mfma_peak"] = round(roof["step_tflops_algorithmic"] / peak, 4)
        if split and "mlp_bwd_dw" in summ:
            # the bandwidth-bound launch (dW) and the step's HBM traffic: design bytes (saved activations) vs the bytes any
            # implementation must move
            n_dw, ms_dw, pts_dw = summ["mlp_bwd_dw"]
            gbs = pts_dw * DW_BYTES_PER_POINT[a.mlp_precision] / (ms_dw * 1e-3) / 1e9
            hbm = {"kernel": "mlp_bwd_dw", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": round(gbs / HBM_PEAK_GBS, 4), "design_bytes_per_launch": int(pts_dw / n_dw * DW_BYTES_PER_POINT[a.mlp_precision]),
                   "traffic": traffic_of("mlp_bwd_dw"),
                   "algorithmic_bytes_per_step": algorithmic_bytes_per_step(wl, wl["channels"])}
            tr = [traffic_of(k) for k in ("mlp_fwd", "mlp_bwd_dx", "mlp_bwd_dw")]
            if all(t is not None for t in tr):
                hbm["k3_traffic_per_step"] = 2 * sum(tr)                  # coarse + fine launches of each kernel
            roof["hbm"] = hbm
    mlp_ms = sum(v[1] for v in summ.values()) / max(n_sampled, 1) if summ else None

    out = {
        "metric": "training rays/s" if not a.oversubscribe else "training rays/s (ranks oversubscribed on one device: not a measurement)",
        "value": round(value, 1), "unit": "rays/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": a.scaling,
        "vs_baseline": None,
        "dtype": DTYPE_NOTE[a.mlp_precision],
        "data": "synthetic",
        "config": {"workload": "%s: %s%s" % (a.workload, wl["name"], "" if a.event_bins == 1 else
                                             " + %d dense event bins (%d event poses per pixel)" % (a.event_bins, a.event_bins + 1)),
                   "rays_global": rays_global, "event_pixels_global": Re_g, "blur_pixels_global": Rr_g, "event_bins": a.event_bins,
                   "rays_per_step_per_gpu": WL.rays_per_step(wl),
                   "samples": "%d+%d" % (wl["S"], wl["S"] + wl["Ni"]), "channels": wl["channels"],
                   "parallelism": "dp%d" % world, "batch_fraction": a.batch_fraction,
                   "median_ms_per_step": round(median_ms, 3), "median_rays_per_s": round(rays_step / world / (median_ms * 1e-3) * world, 1),
                   "mlp_ms_per_step": None if mlp_ms is None else round(mlp_ms, 3),
                   "final_loss": float(losses[0]), "inference_rays_per_s": infer, "mlp_precision": a.mlp_precision,
                   "exact_f32_mfma_rays_per_s": other},
        "roofline": roof,
    }
    if exact is not None:
        out["exact_f32"], out["roofline_f32"] = exact, roof_f32
    if reduced is not None:
        out["reduced_precision"] = reduced
    if alone and out.get("roofline"):
        out["roofline"]["per_kernel_alone"] = alone
    if power and out.get("roofline"):
        out["roofline"]["power"] = power
    if roof_inf is not None:
        out["roofline_inference"] = roof_inf
    if comm is not None:
        out.update(comm)
    if per_rank is not None:
        out["per_rank"] = per_rank
    if bucket_wait is not None:
        out["bucket_wait"] = bucket_wait
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline and not a.primary_only:
            # the workload the metric is quoted on, full size (one step is ~10-20 s of CPU work), and C1, the reference's
            # own CPU-runnable case (BASELINE.json configs[0])
            out["cpu_baseline"] = cpu_baseline(a.workload, a.seed, n_steps=2, warm_fraction=16, threads=a.cpu_threads)
            out["cpu_baseline_c1"] = cpu_baseline("C1", a.seed, n_steps=3, threads=a.cpu_threads)
            try:
                del step, g
                torch.cuda.empty_cache()
                out["torch_gpu_baseline"] = torch_gpu_baseline(a.workload, a.seed, device)
                if a.event_bins == 1:      # the baselines run the reference's one-bin step
                    out["torch_gpu_baseline"]["speedup_vs_it"] = round(out["value"] / out["torch_gpu_baseline"]["value"], 2)
            except Exception as e:   # informational leg only: never fail the bench line on it
                out["torch_gpu_baseline"] = {"error": repr(e)[:200]}
            if a.event_bins == 1 and a.batch_fraction == 1 and not a.rccl_loopback:
                # the path train.py drops onto, unchanged: timed in a child process once this one is done with the device
                torch.cuda.synchronize()
                out["dropin"] = dropin_leg(a)
                base = out.get("torch_gpu_baseline", {}).get("value")
                if base and "value" in out["dropin"]:
                    out["dropin"]["speedup_vs_torch_gpu"] = round(out["dropin"]["value"] / base, 2)
                    out["dropin"]["fraction_of_train_step"] = round(out["dropin"]["value"] / out["value"], 3)
        else:
            out["cpu_baseline"] = None
        if a.rccl_loopback:
            out["config"]["rccl_loopback"] = True
        print(json.dumps(out), file=json_out, flush=True)
    if world > 1 or a.rccl_loopback:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
