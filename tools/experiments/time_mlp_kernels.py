"""Times the three MLP launches in isolation (HIP events, median of N launches) at one network's full-size point count.
Used with BENERF_HIP_LIB=<variant .so> to attribute kernel time to its phases by building variants with parts
compiled out.  Usage: python tools/experiments/time_mlp_kernels.py [n_rays n_samples reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benerf_amd import kernels as K          # noqa: E402
from benerf_amd import run_nerf_helpers      # noqa: E402


def main():
    n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 4081
    n_samples = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    from benerf_amd.model import nerf as nerf_mod
    model = nerf_mod.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=2, skips=[4], use_viewdirs=True,
                          channels=1).to(dev)
    run_nerf_helpers.init_nerf(model)
    packed = model.packed()
    packed.pack()
    ro = torch.randn(n_rays, 3, device=dev) * 0.1
    rd = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev), dim=-1)
    z = torch.sort(torch.rand(n_rays, n_samples, device=dev), dim=-1).values
    raw, acts = K.mlp_fwd(packed, ro, rd, rd, z, True)
    d_raw = torch.randn_like(raw) * 1e-4
    M = n_rays * n_samples

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        ts.sort()
        return ts[len(ts) // 2]

    t_f = timed(lambda: K.mlp_fwd(packed, ro, rd, rd, z, True))
    t_i = timed(lambda: K.mlp_fwd(packed, ro, rd, rd, z, False))
    dacts = [None]

    def dx():
        dacts[0] = K.mlp_bwd_dx(packed, d_raw.view(-1, raw.shape[-1]), acts, n_rays, n_samples)[2]
    t_x = timed(dx)
    gw = [torch.zeros_like(w) for w in packed.weights]
    gb = [torch.zeros_like(b) for b in packed.biases]
    t_w = timed(lambda: K.mlp_bwd_dw(packed, d_raw.view(-1, raw.shape[-1]), acts, dacts[0], n_rays, n_samples, gw, gb, False))
    print("lib=%s mode=%s M=%d fwd %.3f ms (inference launch %.3f)  dx %.3f ms  dw %.3f ms"
          % (os.environ.get("BENERF_HIP_LIB", "default"), K.get_mlp_precision(), M, t_f, t_i, t_x, t_w))


if __name__ == "__main__":
    main()
