"""What does running two K3 launches on two streams buy?  Times the six K3 launches of a C2 step (forward / dX / dW of the coarse
and the fine network) alone and in pairs on two HIP streams (HIP events on a third stream that waits for both), so that the step's
launch order can be chosen from measured pair times instead of assumed ones ("time slicing, not overlap" was
measured on round 3's kernels).  Usage: python tools/experiments/overlap_probe.py [n_rays reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import benerf_amd                             # noqa: E402

benerf_amd.configure_runtime()
from benerf_amd import kernels as K          # noqa: E402
from benerf_amd import run_nerf_helpers      # noqa: E402


def main():
    n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 4081
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 9
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    from benerf_amd.model import nerf as nerf_mod
    nets = {}
    for name, ns in (("c", 64), ("f", 192)):
        model = nerf_mod.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=2, skips=[4], use_viewdirs=True,
                              channels=1).to(dev)
        run_nerf_helpers.init_nerf(model)
        packed = model.packed()
        packed.pack()
        ro = torch.randn(n_rays, 3, device=dev) * 0.1
        rd = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev), dim=-1)
        z = torch.sort(torch.rand(n_rays, ns, device=dev), dim=-1).values
        raw, acts = K.mlp_fwd(packed, ro, rd, rd, z, True)
        d_raw = (torch.randn_like(raw) * 1e-4).view(-1, raw.shape[-1])
        dacts = K.mlp_bwd_dx(packed, d_raw, acts, n_rays, ns, slot="_" + name)[2]
        gw = [torch.zeros_like(w) for w in packed.weights]
        gb = [torch.zeros_like(b) for b in packed.biases]
        nets[name] = dict(p=packed, ro=ro, rd=rd, z=z, acts=acts, d_raw=d_raw, dacts=dacts, gw=gw, gb=gb, ns=ns)
    torch.cuda.synchronize()

    def op(kind, name):
        n = nets[name]
        if kind == "fwd":
            return lambda: K.mlp_fwd(n["p"], n["ro"], n["rd"], n["rd"], n["z"], True)
        if kind == "dx":
            return lambda: K.mlp_bwd_dx(n["p"], n["d_raw"], n["acts"], n_rays, n["ns"], slot="_" + name)
        return lambda: K.mlp_bwd_dw(n["p"], n["d_raw"], n["acts"], n["dacts"], n_rays, n["ns"], n["gw"], n["gb"], False)

    main_s = torch.cuda.current_stream(dev)
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)

    def timed(fa, fb=None):
        ts = []
        for it in range(reps + 2):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(main_s)
            s1.wait_stream(main_s)
            s2.wait_stream(main_s)
            with torch.cuda.stream(s1):
                fa()
            if fb is not None:
                with torch.cuda.stream(s2):
                    fb()
            main_s.wait_stream(s1)
            main_s.wait_stream(s2)
            b.record(main_s)
            torch.cuda.synchronize()
            if it >= 2:
                ts.append(a.elapsed_time(b))
        ts.sort()
        return ts[len(ts) // 2]

    kinds = [("fwd", "c"), ("fwd", "f"), ("dx", "c"), ("dx", "f"), ("dw", "c"), ("dw", "f")]
    alone = {}
    for k in kinds:
        alone[k] = timed(op(*k))
        print("alone %-3s %s  %.3f ms" % (k[0], k[1], alone[k]), flush=True)
    pairs = [(("dx", "c"), ("dw", "f")), (("dw", "f"), ("dx", "c")), (("dx", "f"), ("dw", "c")), (("dw", "c"), ("dx", "f")),
             (("fwd", "c"), ("dw", "f")), (("dw", "f"), ("fwd", "c")), (("fwd", "f"), ("dw", "c")), (("dw", "c"), ("fwd", "f")),
             (("fwd", "c"), ("dw", "c")), (("dx", "c"), ("dw", "c")), (("dx", "f"), ("dw", "f")), (("fwd", "f"), ("dw", "f")),
             (("fwd", "c"), ("dx", "f")), (("dx", "c"), ("dx", "f"))]
    for ka, kb in pairs:
        t = timed(op(*ka), op(*kb))
        s = alone[ka] + alone[kb]
        print("pair  %-3s %s (first) || %-3s %s   %.3f ms   sum alone %.3f   ratio %.3f" % (ka[0], ka[1], kb[0], kb[1], t, s, t / s), flush=True)


if __name__ == "__main__":
    main()
