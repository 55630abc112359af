"""Does a device-idle gap between the forward and the backward launches slow the backward kernels down?  (Round 6: inside the
reference-shaped loop - tools/dropin_driver.py - the K3 launches take 3-8 % longer than inside engine.TrainStep on the same box; the
loop's loss lines leave the device idle for ~1.7 ms between them.)  One network, C2's fine-pass size: forward, [gap], dX, dW, HIP-event
durations, with the host sleeping `gap` ms behind a synchronize between forward and backward.
    python tools/experiments/idle_gap_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benerf_amd import kernels as K          # noqa: E402
from benerf_amd import run_nerf_helpers      # noqa: E402
from benerf_amd.model import nerf as nerf_mod   # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    n_rays, n_samples = 4081, 128
    model = nerf_mod.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=2, skips=[4], use_viewdirs=True, channels=1).to(dev)
    run_nerf_helpers.init_nerf(model)
    packed = model.packed()
    packed.pack()
    ro = torch.randn(n_rays, 3, device=dev) * 0.1
    rd = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev), dim=-1)
    z = torch.sort(torch.rand(n_rays, n_samples, device=dev), dim=-1).values
    gw = [torch.zeros_like(w) for w in packed.weights]
    gb = [torch.zeros_like(b) for b in packed.biases]
    d_raw = torch.randn(n_rays * n_samples, 2, device=dev) * 1e-4

    def once(gap_ms, spin):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        raw, acts = K.mlp_fwd(packed, ro, rd, rd, z, True)
        ev[1].record()
        if gap_ms > 0:
            torch.cuda.synchronize()
            if spin:      # a device that is kept busy with nothing: 1-workgroup fills while the host waits
                t0 = time.perf_counter()
                while time.perf_counter() - t0 < gap_ms * 1e-3:
                    d_raw[:64].mul_(1.0)
            else:
                time.sleep(gap_ms * 1e-3)
        e2 = torch.cuda.Event(enable_timing=True)
        e2.record()
        dx = K.mlp_bwd_dx(packed, d_raw, acts, n_rays, n_samples)
        ev[2].record()
        K.mlp_bwd_dw(packed, d_raw, acts, dx[2], n_rays, n_samples, gw, gb, False)
        ev[3].record()
        torch.cuda.synchronize()
        return ev[0].elapsed_time(ev[1]), e2.elapsed_time(ev[2]), ev[2].elapsed_time(ev[3])

    for gap, spin in ((0, False), (2.0, False), (2.0, True), (0, False), (0.5, False), (5.0, False)):
        for _ in range(3):
            once(gap, spin)
        rows = [once(gap, spin) for _ in range(15)]
        med = [sorted(r[i] for r in rows)[len(rows) // 2] for i in range(3)]
        print("gap %.1f ms %-22s fwd %.3f  dX %.3f  dW %.3f ms" % (gap, "(tiny launches)" if spin else "(device idle)" if gap else "", *med))


main()
