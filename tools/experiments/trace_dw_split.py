"""Per-workgroup start / finish times of the BENERF_MLP_SPLIT dW kernels (mlp_dw_s.hip, round-5 tables) from a tracing build:
which workgroups does each launch wait for, and how long do the thin kernel's three kinds run?

  tools/experiments/build_variant.sh tracedw -DBENERF_TRACE_DW
  BENERF_HIP_LIB=build/lib_tracedw.so python tools/experiments/trace_dw_split.py [n_rays n_samples]

Workspace layout (mlp_common.h: dws_inst_offset / dws_splits): 7 plain instances x DWS_LS splits, FEAT slot and G (VIEWSF) x DWS_GS,
L0 / L5P x DWH_T0, VIEWSP x DWH_T1, RGB x DWH_T2; the stamps sit behind them (u64 [kernel 0 small / 1 big][512][2], 100 MHz)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benerf_amd import kernels as K          # noqa: E402
from benerf_amd import run_nerf_helpers      # noqa: E402
from benerf_amd.model import nerf as nerf_mod  # noqa: E402

LS, GS, T0, T1, T2 = (int(os.environ.get(k, d)) for k, d in (("DWS_LS", 31), ("DWS_GS", 39), ("DWH_T0", 128), ("DWH_T1", 128), ("DWH_T2", 128)))
n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 4081
n_samples = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = nerf_mod.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=2, skips=[4], use_viewdirs=True, channels=1).to(dev)
run_nerf_helpers.init_nerf(model)
packed = model.packed()
packed.pack()
ro = torch.randn(n_rays, 3, device=dev) * 0.1
rd = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev), dim=-1)
z = torch.sort(torch.rand(n_rays, n_samples, device=dev), dim=-1).values
raw, acts = K.mlp_fwd(packed, ro, rd, rd, z, True)
d_raw = torch.randn_like(raw) * 1e-4
d_pts, d_vd, dacts = K.mlp_bwd_dx(packed, d_raw.view(-1, 2), acts, n_rays, n_samples)
gw = [torch.zeros_like(w) for w in packed.weights]
gb = [torch.zeros_like(b) for b in packed.biases]
for _ in range(3):
    K.mlp_bwd_dw(packed, d_raw.view(-1, 2), acts, dacts, n_rays, n_samples, gw, gb, False)
torch.cuda.synchronize()
full, feat, views, thin0, thin1, rgb = 256 * 256 + 256, 256 * 256 + 256 + 257, 128 * 256 + 128, 256 * 64 + 256, 128 * 32 + 128, 4 * 128 + 4
used = 7 * LS * full + GS * feat + GS * views + 2 * T0 * thin0 + T1 * thin1 + T2 * rgb
off = (used + 63) & ~63
ws = K.scratch("dw_ws", 0, dev)
t = ws[off:off + 2 * 512 * 2 * 2].view(torch.int64).cpu().numpy().reshape(2, 512, 2)
small = [("L0+L5P", range(0, T0)), ("VIEWSP", range(T0, T0 + T1 // 2)), ("RGB", range(T0 + T1 // 2, T0 + T1 // 2 + T2))]
big = [("L%d" % (i + 1) if i != 4 else "L5H", range(i * LS, (i + 1) * LS)) for i in range(7)] + [("G+alpha", range(7 * LS, 7 * LS + GS))]
for kern, label, groups in ((0, "thin kernel", small), (1, "big kernel", big)):
    n = groups[-1][1][-1] + 1
    tk = t[kern][:n]
    t0 = tk[:, 0].min()
    print("%s: %d workgroups, span %.1f us (first start -> last finish); starts within %.1f us" %
          (label, n, (tk[:, 1].max() - t0) / 100.0, (tk[:, 0].max() - t0) / 100.0))
    for name, idx in groups:
        idx = list(idx)
        e = (tk[idx, 1] - t0) / 100.0
        s = (tk[idx, 0] - t0) / 100.0
        d = (tk[idx, 1] - tk[idx, 0]) / 100.0
        print("   %-8s x%-3d start median %.1f max %.1f   finish min %.1f median %.1f max %.1f us   duration median %.1f max %.1f us"
              % (name, len(idx), float(np.median(s)), s.max(), e.min(), float(np.median(e)), e.max(), float(np.median(d)), d.max()))
