"""Per-workgroup start / finish times of the split-mode dW kernels from a tracing build: which instance's workgroups does the
launch wait for?

  cd benerf_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DBENERF_TRACE_DW [-DDWH_LS=.. -DDWH_VS=..] -c mlp_dw_h.hip -o /tmp/dw_tr.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/experiments/libbenerf_trace_dw.so $(ls *.o | grep -v mlp_dw_h.o) /tmp/dw_tr.o
  BENERF_HIP_LIB=$PWD/tools/experiments/libbenerf_trace_dw.so python tools/experiments/trace_dw.py [LS FS VS] [n_rays n_samples]

(the other objects must have been compiled with the same -DDWH_* values: the reduce kernel reads the same tables)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benerf_amd import kernels as K          # noqa: E402
from benerf_amd import run_nerf_helpers      # noqa: E402
from benerf_amd.model import nerf as nerf_mod  # noqa: E402

LS = int(sys.argv[1]) if len(sys.argv) > 1 else 28      # point-splits: seven plain 256x256 instances, FEAT, views block
FS = int(sys.argv[2]) if len(sys.argv) > 2 else 38
VS = int(sys.argv[3]) if len(sys.argv) > 3 else 22
n_rays = int(sys.argv[4]) if len(sys.argv) > 4 else 4081
n_samples = int(sys.argv[5]) if len(sys.argv) > 5 else 128
THIN = 128
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
used = 7 * LS * full + FS * feat + VS * views + 2 * THIN * thin0 + THIN * thin1 + THIN * rgb
off = (used + 63) & ~63
if os.environ.get("BENERF_MLP_PRECISION", "split") == "f32":
    # exact-f32 kernel (mlp_dw.hip): one launch of 512 workgroups; split counts of mlp_common.h dw_splits(), overridable here
    # as "name=count,..." in TRACE_DW_SPLITS for variant builds
    splits = dict(L=52, VIEWSF=28, L0=14, L5P=14, VIEWSP=16, RGB=24)
    for kv in filter(None, os.environ.get("TRACE_DW_SPLITS", "").split(",")):
        k_, v_ = kv.split("=")
        splits[k_] = int(v_)
    order = [("L%d" % (i + 1) if i != 4 else "L5H", splits["L"]) for i in range(7)] + [("FEAT", splits.get("FEAT", splits["L"])), ("VIEWSF", splits["VIEWSF"]),
                                                                                     ("L0", splits["L0"]), ("L5P", splits["L5P"]), ("VIEWSP", splits["VIEWSP"]), ("RGB", splits["RGB"])]
    n_wg = sum(c for _, c in order)
    ws = K.scratch("dw_ws", 0, dev)
    lib = __import__("benerf_amd._lib", fromlist=["load"]).load()
    base = lib.benerf_mlp_dw_workspace_floats(n_rays * n_samples) - 4096
    t = ws[base:base + n_wg * 4].view(torch.int64).cpu().numpy().reshape(n_wg, 2)
    t0 = t[:, 0].min()
    print("f32 dW kernel: %d workgroups, span %.1f us; starts within %.1f us" % (n_wg, (t[:, 1].max() - t0) / 100.0, (t[:, 0].max() - t0) / 100.0))
    b0 = 0
    for name, cnt in order:
        e = (t[b0:b0 + cnt, 1] - t0) / 100.0
        s_ = (t[b0:b0 + cnt, 0] - t0) / 100.0
        print("   %-7s x%-3d start median %.1f  finish min %.1f  median %.1f  max %.1f us" % (name, cnt, float(np.median(s_)), e.min(), float(np.median(e)), e.max()))
        b0 += cnt
    sys.exit(0)
ws = K.scratch("dw_ws", 0, dev)
t = ws[off:off + 2 * 512 * 2 * 2].view(torch.int64).cpu().numpy().reshape(2, 512, 2)
names_big = ["L1", "L2", "L3", "L4", "L5H", "L6", "L7", "FEAT"]
for kern, label in ((0, "thin kernel"), (1, "big kernel")):
    tk = t[kern]
    n = 512 if kern == 0 else 256
    tk = tk[:n]
    t0 = tk[:, 0].min()
    print("%s: %d workgroups, span %.1f us (first start -> last finish); starts within %.1f us" %
          (label, n, (tk[:, 1].max() - t0) / 100.0, (tk[:, 0].max() - t0) / 100.0))
    groups = [(names_big[i], range(i * LS, (i + 1) * LS)) for i in range(7)] + [("FEAT", range(7 * LS, 7 * LS + FS)),
                                                                                ("VIEWS", range(7 * LS + FS, 7 * LS + FS + VS))] if kern == 1 else \
        [("L0", range(0, THIN)), ("L5P", range(THIN, 2 * THIN)), ("VIEWSP", range(2 * THIN, 3 * THIN)), ("RGB", range(3 * THIN, 4 * THIN))]
    for name, idx in groups:
        e = (tk[list(idx), 1] - t0) / 100.0
        d = (tk[list(idx), 1] - tk[list(idx), 0]) / 100.0
        print("   %-7s finish min %.1f  median %.1f  max %.1f us   duration median %.1f us" % (name, e.min(), float(np.median(e)), e.max(), float(np.median(d))))
