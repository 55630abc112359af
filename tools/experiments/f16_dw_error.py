"""GPU experiment: error of dW = dY^T X at BASELINE size when the operands are rounded to f16 (f32 accumulation), with
the SAME ReLU masks - isolates operand rounding from the mask flips that separate two forward arithmetics.
Takes the f32-mode kernels' saved activations / gradient rows, recomputes layer 6's dW with torch in f64, in f32 and
with f16-rounded operands."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
from benerf_amd import kernels as K  # noqa: E402
import benerf_oracle as O  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(80)
C, N, S = 1, 4081, 128
p = O.xavier_params(rng, C)
p["alpha_linear.bias"] += 2.0
ws = [p[n + ".weight"].to(dev) for n in K.LAYER_NAMES]
bs = [p[n + ".bias"].to(dev) for n in K.LAYER_NAMES]
net = K.PackedMlp(ws, bs, C)
net.pack()
f32 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32)).to(dev)  # noqa: E731
ro, rd = f32(rng.uniform(-0.5, 0.5, (N, 3))), f32(rng.uniform(-1, 1, (N, 3)))
vd = f32(rng.standard_normal((N, 3)))
vd = vd / vd.norm(dim=-1, keepdim=True)
z = f32(np.sort(rng.random((N, S)), -1))
G = f32(rng.standard_normal((N * S, C + 1)) * np.exp(rng.uniform(-4, 0, (N * S, 1))) / N)
M = N * S
K.set_mlp_precision("f32")
raw, acts = K.mlp_fwd(net, ro, rd, vd, z, True)
gw = [torch.zeros_like(w) for w in ws]
gb = [torch.zeros_like(b) for b in bs]
K.mlp_bwd(net, G, acts, N, S, gw, gb, False)
dacts = K._scratch[("dacts", str(dev), torch.float32)]
for l in (1, 3, 6, 7):
    X = acts[M * 64 + (l - 1) * M * 256: M * 64 + l * M * 256].view(M, 256)
    dY = dacts[l * M * 256:(l + 1) * M * 256].view(M, 256)
    ref = (dY.double().t() @ X.double())
    mx = float(ref.abs().max())
    k32 = gw[l].double()
    t32 = (dY.t() @ X).double()
    s = 2.0 ** (6 - np.floor(np.log2(float(G.abs().max()))))
    t16 = ((dY * s).half().t().float() @ X.half().float()).double() / s
    t16b = ((dY * s).half().t() @ X.half()).double() / s          # f16 GEMM (f32 accumulate inside, f16 output rounding)
    print("layer %d: max|dW| %.3e | vs f64: kernel-f32 %.2e  torch-f32 %.2e  f16-operands %.2e (of max entry); rel L2 f16-operands %.2e"
          % (l, mx, float((k32 - ref).abs().max()) / mx, float((t32 - ref).abs().max()) / mx, float((t16 - ref).abs().max()) / mx,
             float((t16 - ref).norm() / ref.norm())))
K.set_mlp_precision("split")
raw2, acts2 = K.mlp_fwd(net, ro, rd, vd, z, True)
gw2 = [torch.zeros_like(w) for w in ws]
gb2 = [torch.zeros_like(b) for b in bs]
K.mlp_bwd(net, G, acts2, N, S, gw2, gb2, False)
for l in (1, 3, 6, 7):
    mx = float(gw[l].abs().max())
    print("layer %d: split-mode kernels vs f32-mode kernels: max %.2e of max entry, rel L2 %.2e" % (l, float((gw2[l] - gw[l]).abs().max()) / mx,
          float((gw2[l] - gw[l]).norm() / gw[l].norm())))
