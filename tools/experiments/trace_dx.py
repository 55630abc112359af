"""Decodes the phase timestamps a tracing build of the dX kernel leaves in its d_viewdirs output.

  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DBENERF_TRACE_DX -c benerf_amd/csrc/mlp_bwd_h.hip -o /tmp/bwd_tr.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libtr.so $(ls benerf_amd/csrc/*.o | grep -v mlp_bwd_h.o) /tmp/bwd_tr.o
  BENERF_HIP_LIB=/tmp/libtr.so python tools/experiments/trace_dx.py        # on an MI355X

Columns: P0a, P0b, P1 (rgb head), P2 GEMM, P2 epilogue, P3 GEMM, P3 epilogue, then K-loop / epilogue of layers 7..1
(the layer-5 entry includes the skip connection's row GEMM), P5 (L0 row GEMM, scratch fill), P6; units of 10 ns."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benerf_amd import kernels as K
from benerf_amd import run_nerf_helpers
from benerf_amd.model import nerf as nerf_mod
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = nerf_mod.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=2, skips=[4], use_viewdirs=True, channels=1).to(dev)
run_nerf_helpers.init_nerf(model)
packed = model.packed(); packed.pack()
n_rays, n_samples = 4081, 128
ro = torch.randn(n_rays, 3, device=dev) * 0.1
rd = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev), dim=-1)
z = torch.sort(torch.rand(n_rays, n_samples, device=dev), dim=-1).values
raw, acts = K.mlp_fwd(packed, ro, rd, rd, z, True)
d_raw = torch.randn_like(raw) * 1e-4
for _ in range(3):
    d_pts, d_vd, _ = K.mlp_bwd_dx(packed, d_raw.view(-1, 2), acts, n_rays, n_samples)
torch.cuda.synchronize()
t = d_vd.view(-1).view(torch.int64)[:2048 * 64].cpu().numpy().reshape(2048, 64)
np.save("gpurun_out/trace_dx.npy", t)
hw = t[:, 63]
t0 = t[:, 0].min()
print("blk  xcc hw_id(hex)  start(10ns)  phase durations (10 ns units)")
for b in list(range(0, 8)) + list(range(256, 264)) + list(range(512, 520)):
    r = t[b]
    idx = [0, 1, 2, 3, 4, 5, 6, 7] + list(range(10, 24)) + [40, 41, 42]
    ts = [int(r[i] - t0) for i in idx]
    print(b, int(hw[b] >> 32) & 0xf, hex(int(hw[b] & 0xffffffff)), ts[0], [ts[i + 1] - ts[i] for i in range(len(ts) - 1)])
# co-residency: same (xcc, cu/se bits) -> group
key = (hw >> 32 << 32) | (hw & 0xfff00)      # drop wave/simd id bits
import collections
g = collections.defaultdict(list)
for b in range(512):
    g[int(key[b])].append(b)
print("distinct CU keys among first 512 blocks:", len(g), "sizes", collections.Counter(len(v) for v in g.values()))
print("examples", list(g.values())[:6])

# gaps between consecutive workgroups in one CU slot: a CU holds two dX workgroups at a time; when one finishes, how long until
# the next one of the launch starts there?  (finish = last stamp, TR(42); start = TR(0))
slots = collections.defaultdict(list)
for b in range(2048):
    slots[int(key[b])].append((int(t[b, 0]), int(t[b, 42]), b))
gaps, durs = [], []
for k_, lst in slots.items():
    lst.sort()
    ends = []
    for s_, e_, b_ in lst:
        durs.append(e_ - s_)
        free = [x for x in ends if x <= s_]
        if free:            # the slot this workgroup took over: the latest finish before its start
            prev = max(free)
            gaps.append(s_ - prev)
            ends.remove(prev)
        ends.append(e_)
gaps, durs = np.array(gaps), np.array(durs)
print("workgroup duration (10 ns): median %d, p10 %d, p90 %d" % (np.median(durs), np.percentile(durs, 10), np.percentile(durs, 90)))
print("gap between a workgroup's finish and its successor's start on the same CU (10 ns): median %d, p10 %d, p90 %d, n = %d"
      % (np.median(gaps), np.percentile(gaps, 10), np.percentile(gaps, 90), len(gaps)))
