import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
REPORT = []


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The oracle's torch-CPU steps are chains of thousands of small ATen operators: on the GPU box's 128+ hardware threads the
    # default intra-op pool is 3 x SLOWER than 16 threads (bench.py's cpu_baseline sweep: 8.8 s per full C2 step at 16 threads, 13.6 s
    # at 64, 29.6 s at 128).  Round 6: the full-size oracle evaluations are 500 of the GPU suite's 815 s at the default.
    try:
        import torch
        torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    except ImportError:
        pass


@pytest.fixture(params=["split", "f32"])
def mlp_precision(request):
    """Runs a GPU test under both MFMA arithmetic modes of the fused MLP kernels (include/benerf_hip.h): 'split' (the default:
    every GEMM - forward, dX chain, dW - as 3 f16 MFMAs on hi/lo operands, 22 bits in flight, 19-bit saved operands, f32
    accumulate) and 'f32' (exact f32 MFMA).  Both modes are held to the same tolerances everywhere."""
    from benerf_amd import kernels
    kernels.set_mlp_precision(request.param)
    REPORT.append("---- mlp precision: %s (%s)" % (request.param, request.node.name))
    yield request.param
    kernels.set_mlp_precision("split")


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name + ".npz"))
        return cache[name]

    return load


def report(name, got, ref, atol=0.0, rtol=0.0):
    """Record max abs / rel error, then assert allclose.  Accepts torch or numpy."""
    import torch
    g = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    r = ref.detach().cpu().numpy() if isinstance(ref, torch.Tensor) else np.asarray(ref)
    assert g.shape == r.shape, (name, g.shape, r.shape)
    d = np.abs(g.astype(np.float64) - r.astype(np.float64))
    nan_mismatch = np.isnan(g) != np.isnan(r)
    d = np.where(np.isnan(g) & np.isnan(r), 0.0, d)
    mx = float(np.nanmax(d)) if d.size else 0.0
    scale = float(np.nanmax(np.abs(r))) if r.size and not np.isnan(r).all() else 0.0
    both_nan = np.isnan(g) & np.isnan(r)
    with np.errstate(invalid="ignore"):
        within = (d <= atol + rtol * np.abs(r)) | both_nan
    ok = (not nan_mismatch.any()) and bool(np.all(within))
    line = "%-58s max|d|=%.3e  max|ref|=%.3e  atol=%g rtol=%g  %s" % (name, mx, scale, atol, rtol, "ok" if ok else "FAIL")
    REPORT.append(line)
    print(line)
    assert ok, line
    return mx


def pytest_sessionfinish(session, exitstatus):
    if not REPORT:
        return
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_report.txt"), "a") as f:      # several pytest sessions of one GPU call share the file
            f.write("\n".join(REPORT) + "\n")
    except OSError:
        pass
