"""Per-phase wall-clock durations of the BENERF_MLP_SPLIT forward (training launch) and dX kernels from tracing builds:

  tools/experiments/build_variant.sh trfwd -DBENERF_TRACE_FWD ; tools/experiments/build_variant.sh trdx -DBENERF_TRACE_DX
  BENERF_HIP_LIB=build/lib_trfwd.so python tools/experiments/trace_phases.py fwd
  BENERF_HIP_LIB=build/lib_trdx.so  python tools/experiments/trace_phases.py dx

Thread 0 of the first 2048 workgroups stamps the 100 MHz wall clock at every phase boundary (into `raw` / `d_viewdirs`, which a tracing
build does not write).  Printed: median over the workgroups of each phase's duration in microseconds, and the tile total."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benerf_amd import kernels as K          # noqa: E402
from benerf_amd import run_nerf_helpers      # noqa: E402
from benerf_amd.model import nerf as nerf_mod  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "dx"       # fwd | fwd_inf (inference launch: nothing saved) | dx
n_rays, n_samples = 4081, 128
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = nerf_mod.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=2, skips=[4], use_viewdirs=True, channels=1).to(dev)
run_nerf_helpers.init_nerf(model)
packed = model.packed()
packed.pack()
ro = torch.randn(n_rays, 3, device=dev) * 0.1
rd = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev), dim=-1)
z = torch.sort(torch.rand(n_rays, n_samples, device=dev), dim=-1).values
for _ in range(20):     # warm clocks
    raw, acts = K.mlp_fwd(packed, ro, rd, rd, z, True)
torch.cuda.synchronize()
if which == "fwd_inf":
    for _ in range(20):
        raw, _ = K.mlp_fwd(packed, ro, rd, rd, z, False)
    torch.cuda.synchronize()
if which.startswith("fwd"):
    t = raw.view(-1).view(torch.int64)[:2048 * 32].cpu().numpy().reshape(2048, 32)
    names = ["PE prologue", "PE save + L0 GEMM + epilogue"] + [x for l in range(1, 8) for x in ("L%d K-loop" % l, "L%d epilogue" % l)] + \
            ["alpha head + PE(dir)", "VIEWS K-loop", "VIEWS epilogue", "hv save + rgb head + raw"]
    idx = list(range(0, 21))
    hw = t[:, 31]
else:
    d_raw = torch.randn_like(raw) * 1e-4
    for _ in range(20):
        d_pts, d_vd, _ = K.mlp_bwd_dx(packed, d_raw.view(-1, 2), acts, n_rays, n_samples)
    torch.cuda.synchronize()
    t = d_vd.view(-1).view(torch.int64)[:2048 * 64].cpu().numpy().reshape(2048, 64)
    names = ["P0 d_raw scale", "P1 rgb backward -> dYv", "P2 dPE(dir) + VIEWSC GEMM", "P2 d_viewdirs + epilogue (dY7)"] + \
            [x for l in range(7, 0, -1) for x in ("L%d K-loop%s" % (l, " (+ skip dPE block)" if l == 5 else ""), "L%d epilogue (dY%d)" % (l, l - 1))] + \
            ["guard + dPE reload", "P5 L0 row GEMM + scratch fill", "P6 d_pts"]
    idx = list(range(0, 22))
    hw = t[:, 63]
d = np.stack([t[:, idx[i + 1]] - t[:, idx[i]] for i in range(len(idx) - 1)], axis=1) / 100.0
tot = (t[:, idx[-1]] - t[:, idx[0]]) / 100.0
print("%s kernel, %d points: tile total median %.1f us (min %.1f, max %.1f) over %d workgroups" % (which, n_rays * n_samples, float(np.median(tot)), tot.min(), tot.max(), len(tot)))
for i, n in enumerate(names):
    print("   %-36s median %6.2f us   (10%% %6.2f, 90%% %6.2f)   %4.1f %% of the tile" % (n, float(np.median(d[:, i])), float(np.percentile(d[:, i], 10)), float(np.percentile(d[:, i], 90)), 100 * float(np.median(d[:, i])) / float(np.median(tot))))
# gap between two workgroups of one CU: the kernel holds a CU's whole LDS, so the next workgroup starts when the previous one has
# drained its stores and released its resources.  HW_ID (gfx9): CU_ID bits [11:8], SH_ID bit 12, SE_ID bits [15:13]; XCC_ID bits [3:0]
cu = ((hw >> 32) & 0xf) * 4096 + ((hw >> 8) & 0xff)
gaps, first_last = [], []
for c in np.unique(cu):
    m = np.where(cu == c)[0]
    o = m[np.argsort(t[m, idx[0]])]
    if len(o) > 1:
        gaps.extend(((t[o[1:], idx[0]] - t[o[:-1], idx[-1]]) / 100.0).tolist())
gaps = np.array(gaps)
print("   %d CUs seen, %d successive workgroup pairs: gap (last stamp of one workgroup -> first stamp of the next on the same CU) median %.2f us (10%% %.2f, 90%% %.2f)"
      % (len(np.unique(cu)), len(gaps), float(np.median(gaps)), float(np.percentile(gaps, 10)), float(np.percentile(gaps, 90))))
