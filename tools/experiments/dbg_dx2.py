import os, sys, torch
sys.path.insert(0, "/root/repo")
from benerf_amd import kernels as K, run_nerf_helpers
from benerf_amd.model import nerf as nerf_mod
dev = torch.device("cuda:0"); torch.manual_seed(0)
model = nerf_mod.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=2, skips=[4], use_viewdirs=True, channels=1).to(dev)
run_nerf_helpers.init_nerf(model)
packed = model.packed(); packed.pack()
n_rays, n_samples = 24, 16
ro = torch.randn(n_rays, 3, device=dev) * 0.1
rd = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev), dim=-1)
z = torch.sort(torch.rand(n_rays, n_samples, device=dev), dim=-1).values
raw, acts = K.mlp_fwd(packed, ro, rd, rd, z, True)
d_raw = torch.randn_like(raw) * 1e-3
K.scratch("dacts", 10_000_000, dev).fill_(0)
d_pts, d_vd, dacts = K.mlp_bwd_dx(packed, d_raw.view(-1, 2), acts, n_rays, n_samples)
torch.cuda.synchronize()
M = n_rays * n_samples; Mp = (M + 127) // 128 * 128
h = dacts.view(torch.int16)[: 10 * Mp * 256].cpu()
tag = sys.argv[1]
torch.save(h, "/tmp/dy_%s.pt" % tag)
if tag == "b":
    a = torch.load("/tmp/dy_a.pt")
    diff = (a != h).nonzero().view(-1)
    print("differing halfs:", diff.numel())
    for i in diff[:24].tolist():
        l = i // (Mp * 256); r = i % (Mp * 256); blk = r // 2048; n = (r % 2048) // 8; pnt = r % 8
        print("layer", l, "block", blk, "feature", n, "point", pnt, "ptr-variant %04x  buffer-variant %04x" % (int(a[i]) & 0xffff, int(h[i]) & 0xffff))
