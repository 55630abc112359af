"""Renders a few 480 x 768 frames through Graph.render_video (the reference's call shape) - for a kernel trace of one frame:
    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -o t -- python tools/experiments/frame_trace.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benerf_amd import workloads as WL, run_nerf_helpers      # noqa: E402
from benerf_amd.model import optimize      # noqa: E402

args = WL.make_args("C2")
cam = WL.CAMERAS["unreal"]
torch.manual_seed(0)
model = optimize.Model(args)
model.graph.to("cuda:0")
g = model.build_network(args)
run_nerf_helpers.init_nerf(g.nerf)
run_nerf_helpers.init_nerf(g.nerf_fine)
K = np.array([[cam["fx"], 0, cam["cx"]], [0, cam["fy"], cam["cy"]], [0, 0, 1]], dtype=np.float32)
pose = g.get_pose_rgb(args, [0, 1], seg_num=3).detach()[1:2]
for i in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = g.render_video(0, pose, cam["H"], cam["W"], K, args, np.array([]), type="rgb")
    torch.cuda.synchronize()
    print("frame %d: %.1f ms" % (i, (time.perf_counter() - t0) * 1e3))
