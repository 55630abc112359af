"""GPU experiment: the register-chain forward schedule (mlp_fwd_r.hip, BENERF_FWD_R=1, inference launches) against the shipped
split forward: outputs (same arithmetic; the two heads sum in a different order) and launch time at 522 k points.
usage: BENERF_FWD_R=1 python tools/experiments/fwd_r_check.py   (and without the variable for the reference timing)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benerf_amd import kernels as K, run_nerf_helpers  # noqa: E402
from benerf_amd.model import nerf as nerf_mod  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
for C in (1, 3):
    model = nerf_mod.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=C + 1, skips=[4], use_viewdirs=True, channels=C).to(dev)
    run_nerf_helpers.init_nerf(model)
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() == 1:
                p.uniform_(-0.1, 0.1)          # non-zero biases: the bias path is part of the check
    packed = model.packed()
    packed.pack()
    for n_rays, n_samples in ((37, 48), (4081, 128)):
        ro = torch.randn(n_rays, 3, device=dev) * 0.1
        rd = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev), dim=-1)
        z = torch.sort(torch.rand(n_rays, n_samples, device=dev), dim=-1).values
        raw_i, _ = K.mlp_fwd(packed, ro, rd, rd, z, False, precision="split")       # R schedule when BENERF_FWD_R=1
        raw_t, _ = K.mlp_fwd(packed, ro, rd, rd, z, True, precision="split")        # training launch: always the shipped kernel
        err = float((raw_i - raw_t).abs().max() / raw_t.abs().max())
        print("C=%d %d x %d: inference vs training launch, max |d raw| / max |raw| = %.2e%s" % (C, n_rays, n_samples, err, "  (identical)" if torch.equal(raw_i, raw_t) else ""))
        assert err < 2e-6, err

        def timed(fn, reps=20):
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
        if n_rays > 1000:
            print("   FWD_R=%s  inference launch %.3f ms (incl. the empty AUTO launch)   training launch %.3f ms" % (
                os.environ.get("BENERF_FWD_R", "0"), timed(lambda: K.mlp_fwd(packed, ro, rd, rd, z, False, precision="split")),
                timed(lambda: K.mlp_fwd(packed, ro, rd, rd, z, True, precision="split"))))
K.check_mlp_status(dev)
