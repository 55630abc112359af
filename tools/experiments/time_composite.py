"""GPU experiment: compositing forward / backward launch times at C2 size (4081 rays, 64 and 192 samples), with and without
the max |d_raw| output.  usage: python tools/experiments/time_composite.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benerf_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for S in (64, 192):
    N, C = 4081, 1
    raw = torch.randn(N, S, C + 1, device=dev)
    z = torch.sort(torch.rand(N, S, device=dev), dim=-1).values
    rd = torch.randn(N, 3, device=dev)
    g = torch.randn(N, C, device=dev) * 1e-3
    amax = torch.zeros(1, device=dev)
    dd = torch.zeros(N, 3, device=dev)
    t_f = timeit(lambda: K.composite_fwd(raw, z, rd, None, 1.0, 7, 3, want=("rgb_map", "weights")))
    t_b0 = timeit(lambda: K.composite_bwd(raw, z, rd, None, 1.0, 7, 3, g, d_rays_d=dd))
    def with_amax():      # the word is zero at the start of every training step: the running maximum has to be rebuilt each time
        amax.zero_()
        K.composite_bwd(raw, z, rd, None, 1.0, 7, 3, g, d_rays_d=dd, absmax_out=amax)
    t_z = timeit(lambda: amax.zero_())
    t_b1 = timeit(with_amax) - t_z
    t_b2 = timeit(lambda: K.composite_bwd(raw, z, rd, None, 0.0, 7, 3, g, d_rays_d=dd))
    print("S=%3d  fwd %.1f us   bwd %.1f us   bwd + absmax %.1f us   bwd without noise %.1f us  (back-to-back launches, includes launch overhead)"
          % (S, t_f, t_b0, t_b1, t_b2))
