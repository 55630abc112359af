import os, sys, torch
sys.path.insert(0, "/root/repo")
from benerf_amd import kernels as K, run_nerf_helpers
from benerf_amd.model import nerf as nerf_mod
dev = torch.device("cuda:0"); torch.manual_seed(0)
model = nerf_mod.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=2, skips=[4], use_viewdirs=True, channels=1).to(dev)
run_nerf_helpers.init_nerf(model)
packed = model.packed(); packed.pack()
for n_rays, n_samples in ((24, 16), (128, 64)):
    ro = torch.randn(n_rays, 3, device=dev) * 0.1
    rd = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev), dim=-1)
    z = torch.sort(torch.rand(n_rays, n_samples, device=dev), dim=-1).values
    raw, acts = K.mlp_fwd(packed, ro, rd, rd, z, True)
    d_raw = torch.randn_like(raw) * 1e-3
    d_pts, d_vd, dacts = K.mlp_bwd_dx(packed, d_raw.view(-1, 2), acts, n_rays, n_samples)
    torch.cuda.synchronize()
    M = n_rays * n_samples; Mp = (M + 127) // 128 * 128
    h = dacts.view(torch.float16)
    for l in range(10):
        w = 256 if l < 9 else 128
        off = l * Mp * 256
        seg = h[off: off + Mp * w].float()
        print(n_rays, n_samples, "layer", l, "nan", int(torch.isnan(seg).sum()), "inf", int(torch.isinf(seg).sum()), "absmax %.3g" % float(seg[torch.isfinite(seg)].abs().max()))
    seg = h[7 * Mp * 256: 8 * Mp * 256].float().view(Mp // 8, 256, 8)
    bad = (~torch.isfinite(seg)) | (seg.abs() > 1e3)
    idx = bad.nonzero()
    print("bad count", idx.shape[0]); print(idx[:40].tolist())
    print("bad values", seg[bad][:24].tolist())
    b0 = idx[0].tolist()
    print("neighbours of first bad unit", seg[b0[0], b0[1]].tolist(), "prev feature", seg[b0[0], b0[1] - 1].tolist())
    raw_bits = h[7 * Mp * 256: 8 * Mp * 256].view(torch.int16).view(Mp // 8, 256, 8)
    print("bits", [hex(int(x) & 0xffff) for x in raw_bits[b0[0], b0[1]].tolist()])
    import collections
    cnt = collections.Counter()
    for b, n, pnt in idx.tolist():
        row = (b * 8 + pnt) % 128
        cnt[(row >> 5, (row & 31) >> 3, (row >> 2) & 1, row & 3, n >> 5 & 1, n & 31)] += 1
    print("distinct (rt, e>>2, half, e&3, c, lane&31):", len(cnt))
    print(sorted(cnt.items())[:60])
