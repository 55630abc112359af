"""GPU experiment: the split forward kernel (training mode) is sometimes ~25 % slower in the FIRST process on a fresh box.
Times the same launch with its saved-activation buffer placed at different offsets of freshly allocated memory."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import benerf_oracle as O  # noqa: E402
from benerf_amd import _lib, kernels as K  # noqa: E402

dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
C = 1
p = O.xavier_params(rng, C)
net = K.PackedMlp([p[n + ".weight"].to(dev) for n in K.LAYER_NAMES], [p[n + ".bias"].to(dev) for n in K.LAYER_NAMES], C)
net.pack()
N, S = 4081, 128
ro = torch.rand(N, 3, device=dev) - 0.5
rd = torch.rand(N, 3, device=dev) - 0.5
vd = torch.nn.functional.normalize(torch.randn(N, 3, device=dev), dim=-1)
z = torch.sort(torch.rand(N, S, device=dev), -1)[0]
lib = _lib.load()
n_act = lib.benerf_mlp_act_floats(N * S)
raw = torch.empty(N, S, C + 1, device=dev)
s = net.struct()
st = K.mlp_status(dev)


def run(acts, iters=6):
    ts = []
    for i in range(iters + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _lib.check(lib.benerf_mlp_fwd(ctypes.byref(s), net.packed.data_ptr(), C, N, S, ro.data_ptr(), rd.data_ptr(), vd.data_ptr(),
                                      z.data_ptr(), raw.data_ptr(), acts.data_ptr(), 1, st.data_ptr(), torch.cuda.current_stream().cuda_stream), "fwd")
        b.record()
        torch.cuda.synchronize()
        if i:
            ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


print("acts buffer: %.2f GB" % (n_act * 4 / 1e9))
bufs = []
for k in range(8):
    buf = torch.empty(n_act, dtype=torch.float32, device=dev)
    bufs.append(buf)
    print("buffer %d at 0x%x: fwd_train %.3f ms" % (k, buf.data_ptr(), run(buf)), flush=True)
print("again, reverse order")
for k in reversed(range(8)):
    print("buffer %d: fwd_train %.3f ms" % (k, run(bufs[k])), flush=True)
