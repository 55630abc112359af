"""GPU experiment: is the training step power-limited?  Samples socket power, power cap, shader clock and temperature through
amdsmi every ~10 ms from a thread while the C2 step runs in a loop (bench-shaped: same inputs, same engine.TrainStep), then
while single K3 kernels run back to back, and prints the averages per phase.
usage: python tools/experiments/power_trace.py [seconds per phase, default 4]"""
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import amdsmi  # noqa: E402


class Sampler(threading.Thread):
    def __init__(self, handle):
        super().__init__(daemon=True)
        self.h, self.rows, self.stop_flag, self.tag = handle, [], False, "idle"

    def run(self):
        while not self.stop_flag:
            try:
                p = amdsmi.amdsmi_get_power_info(self.h)
                c = amdsmi.amdsmi_get_clock_info(self.h, amdsmi.AmdSmiClkType.GFX)
                self.rows.append((self.tag, time.perf_counter(), p, c))
            except Exception as e:      # noqa: BLE001
                self.rows.append((self.tag, time.perf_counter(), {"error": repr(e)}, {}))
            time.sleep(0.01)


def num(d, *keys):
    for k in keys:
        v = d.get(k)
        if isinstance(v, (int, float)):
            return float(v)
    return float("nan")


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    amdsmi.amdsmi_init()
    h = amdsmi.amdsmi_get_processor_handles()[0]
    try:
        print("power cap info:", amdsmi.amdsmi_get_power_cap_info(h))
    except Exception as e:      # noqa: BLE001
        print("power cap info unavailable:", repr(e))
    s = Sampler(h)
    s.start()
    time.sleep(1.0)

    import bench
    from benerf_amd import engine, kernels as K, workloads as WL
    dev = torch.device("cuda:0")
    wl = dict(WL.WORKLOADS["C2"])
    cam = WL.CAMERAS[wl["cam"]]
    args = WL.make_args(wl)
    g = bench.build_graph(args, dev, 0)
    cam_o = engine.Camera(cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    step = engine.TrainStep(g, args, cam_o, cam_o, dev)
    rng = np.random.default_rng(0)
    HW = cam["H"] * cam["W"]
    accu = torch.from_numpy(rng.integers(-3, 4, HW).astype(np.float32)).to(dev)
    img = torch.from_numpy(rng.random((HW, 1)).astype(np.float32)).to(dev)
    idx_e = torch.from_numpy(rng.permutation(HW)[:wl["Re"]]).to(dev)
    idx_r = torch.from_numpy(rng.permutation(HW)[:wl["Rr"]]).to(dev)
    ets, rts = torch.tensor([0.3, 0.4], device=dev), torch.tensor([0.0, 1.0], device=dev)

    def phase(tag, fn, sync_every=20):
        torch.cuda.synchronize()
        s.tag = tag
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < secs:
            for _ in range(sync_every):
                fn()
            torch.cuda.synchronize()
            n += sync_every
        dt = time.perf_counter() - t0
        s.tag = "idle"
        time.sleep(0.5)
        return dt / n * 1e3

    res = {}
    res["step"] = phase("step", lambda: step.step(ets, rts, idx_e, idx_r, accu, img))
    # single kernels back to back, full fine-network size
    N, S = 4081, 128
    net = step.net_f.packed
    ro = torch.randn(N, 3, device=dev) * 0.1
    rd = torch.nn.functional.normalize(torch.randn(N, 3, device=dev), dim=-1)
    z = torch.sort(torch.rand(N, S, device=dev), dim=-1).values
    raw, acts = K.mlp_fwd(net, ro, rd, rd, z, True)
    d_raw = torch.randn_like(raw) * 1e-4
    res["fwd"] = phase("fwd", lambda: K.mlp_fwd(net, ro, rd, rd, z, True))
    res["fwd_inference"] = phase("fwd_inference", lambda: K.mlp_fwd(net, ro, rd, rd, z, False))
    out = K.mlp_bwd_dx(net, d_raw.view(-1, 2), acts, N, S)
    res["dx"] = phase("dx", lambda: K.mlp_bwd_dx(net, d_raw.view(-1, 2), acts, N, S))
    res["dw"] = phase("dw", lambda: K.mlp_bwd_dw(net, d_raw.view(-1, 2), acts, out[2], N, S, step.net_f.gviews_w, step.net_f.gviews_b, False))
    big = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    res["copy"] = phase("copy", lambda: big[: 1 << 27].copy_(big[1 << 27:]))
    s.stop_flag = True
    s.join()
    print("%-14s %9s %9s %9s %9s %8s" % ("phase", "ms/iter", "power W", "max W", "gfx MHz", "samples"))
    for tag in ["idle"] + list(res):
        rows = [r for r in s.rows if r[0] == tag]
        pw = np.array([num(r[2], "current_socket_power", "average_socket_power", "socket_power") for r in rows])
        ck = np.array([num(r[3], "clk", "cur_clk", "current_clk") for r in rows])
        print("%-14s %9.3f %9.1f %9.1f %9.1f %8d" % (tag, res.get(tag, float("nan")), np.nanmean(pw) if len(pw) else float("nan"),
                                                    np.nanmax(pw) if len(pw) else float("nan"), np.nanmean(ck) if len(ck) else float("nan"), len(rows)))
    print("raw sample:", s.rows[len(s.rows) // 2][2], s.rows[len(s.rows) // 2][3])


if __name__ == "__main__":
    main()
