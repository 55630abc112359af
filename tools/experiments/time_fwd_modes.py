import os, sys, torch
sys.path.insert(0, "/root/repo")
from benerf_amd import kernels as K, run_nerf_helpers
from benerf_amd.model import nerf as nerf_mod
dev = torch.device("cuda:0"); torch.manual_seed(0)
model = nerf_mod.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=2, skips=[4], use_viewdirs=True, channels=1).to(dev)
run_nerf_helpers.init_nerf(model)
packed = model.packed(); packed.pack()
n_rays, n_samples = 4081, 128
ro = torch.randn(n_rays, 3, device=dev) * 0.1
rd = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev), dim=-1)
z = torch.sort(torch.rand(n_rays, n_samples, device=dev), dim=-1).values
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort(); return ts[len(ts) // 2]
print("fwd save %.3f ms   fwd inference(split only) %.3f ms" % (timed(lambda: K.mlp_fwd(packed, ro, rd, rd, z, True)), timed(lambda: K.mlp_fwd(packed, ro, rd, rd, z, False, precision="split"))))
