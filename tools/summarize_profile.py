#!/usr/bin/env python
"""Condenses a rocprofv3 output directory (kernel-trace stats + separate PMC passes, as produced
by the commands in profiles/README.md) into the small files committed under profiles/.

    python tools/summarize_profile.py gpurun_out/prof_r1_split profiles r01_split
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

# kernel-name fragment -> (bench.py timer group, part).  A group's per-launch figure is the SUM over its parts
# (the split-mode dW launch is three kernels: thin instances, big instances, reduce).
PARTS = [
    ("mlp_fwd_split_kernel", "mlp_fwd", "fwd"),
    ("mlp_fwd_kernel", "mlp_fwd", "fwd"),
    ("mlp_bwd_split_kernel", "mlp_bwd_dx", "dx"),
    ("grad_absmax22_kernel", "mlp_bwd_dx", "absmax"),
    ("mlp_dw_split_big_kernel", "mlp_bwd_dw", "big"),
    ("mlp_dw_split_small_kernel", "mlp_bwd_dw", "small"),
    ("mlp_bwd_f16_kernel", "mlp_bwd_dx", "dx"),
    ("grad_absmax_kernel", "mlp_bwd_dx", "absmax"),
    ("mlp_bwd_kernel", "mlp_bwd_dx", "dx"),
    ("mlp_dw_f16_big_kernel", "mlp_bwd_dw", "big"),
    ("mlp_dw_f16_small_kernel", "mlp_bwd_dw", "small"),
    ("mlp_dw_kernel", "mlp_bwd_dw", "dw"),
    ("dw_reduce_kernel", "mlp_bwd_dw", "reduce"),
    ("dw_compose_kernel", "mlp_bwd_dw", "compose"),      # round 5: feature / views weight gradients composed from G = dhv^T h7
]


def classify(name):
    for frag, group, part in PARTS:
        if frag in name:
            return group, part
    return None


# HBM-bound kernels outside K3: name fragment -> (label, algorithmic bytes per launch at C2 as a function of
# the launch's grid: R = rays of the step (4081), S/F = 64 / 128 samples, C = 1).  Bytes: what the launch must read + write.
R_, S_, F_, C_ = 4081, 64, 192, 1
SMALL = [
    ("spline_fwd_kernel", "K1 spline poses fwd", lambda n: 30 * 4 + 19 * 48),
    ("spline_bwd_pair_kernel", "K1 spline poses bwd (both trajectories)", lambda n: 30 * 4 + 21 * 48 + 54 * 4),
    ("rays_fwd_kernel", "K2 rays fwd", lambda n: R_ * (8 + 36) // 2),
    ("rays_bwd_kernel", "K2 rays bwd", lambda n: R_ * (8 + 36) // 2),
    ("stratified_z_kernel", "K2 stratified depths", lambda n: R_ * S_ * 4),
    ("composite_fwd_kernel", "K4 compositing fwd (avg coarse / fine)", lambda n: R_ * ((S_ + F_) // 2) * (8 + 4 + 4) ),
    ("composite_bwd_kernel", "K4 compositing bwd (avg coarse / fine)", lambda n: R_ * ((S_ + F_) // 2) * (8 + 4 + 8)),
    ("sample_pdf_merge_kernel", "K5 sample_pdf + merge", lambda n: R_ * (S_ * 8 + F_ * 4)),
    ("loss_stats_kernel", "K6 loss statistics", lambda n: R_ * 2 * 4),
    ("loss_grads_kernel", "K6 loss gradients", lambda n: R_ * 4 * 4),
    ("event_window_accumulate_kernel", "K7 event window accumulate (10% of 2M events)", lambda n: 200000 * 12 + 480 * 768 * 4),
    ("adam_kernel", "K8 Adam (avg nets / pose)", lambda n: (1191172 + 30) // 2 * 28),
    ("pack_kernel", "K8 weight re-pack (one network)", lambda n: 595586 * 4 + 2 * 1191936 * 4),
    ("ray_grad_reduce_kernel", "K2 per-point -> per-ray gradient reduce (avg)", lambda n: R_ * ((S_ + F_) // 2) * 28),
    ("dw_reduce_kernel", "K3 dW partial-sum reduce", lambda n: 256 * 66000 * 4),
]


def small_kernel_table(trace_csv):
    """avg duration and achieved GB/s against the algorithmic bytes for the kernels outside the fused MLP (C2 shapes)."""
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(trace_csv)):
        for frag, label, _ in SMALL:
            if frag in r["Kernel_Name"]:
                dur[frag].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
                break
    out = {}
    for frag, label, fn in SMALL:
        d = dur.get(frag)
        if not d:
            continue
        us = sum(d) / len(d)
        b = fn(0)
        out[label] = {"dispatches": len(d), "avg_us": round(us, 2), "algorithmic_bytes": b, "achieved_GBps": round(b / us / 1e3, 2)}
    return out


def find(src, sub, leaf):
    hits = glob.glob(os.path.join(src, sub, "**", leaf), recursive=True)
    return hits[0] if hits else None


def main():
    src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    os.makedirs(dst, exist_ok=True)
    shutil.copy(find(src, "trace", "t_kernel_stats.csv"), os.path.join(dst, "%s_kernel_stats.csv" % tag))
    dur = collections.defaultdict(list)                       # (group, part) -> ms per dispatch
    for r in csv.DictReader(open(find(src, "trace", "t_kernel_trace.csv"))):
        c = classify(r["Kernel_Name"])
        if c:
            dur[c].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    out = {"note": "FETCH_SIZE / WRITE_SIZE are rocprofv3 KB units; on gfx950 FETCH_SIZE counts 64 B per 128-B request for "
                   "wide coalesced streaming reads (MI355X_MICROARCH.md, HBM) => fetch_bytes_corrected = 2 x FETCH_SIZE x 1024. "
                   "Launches alternate coarse (M = 261 184 points) / fine (M = 522 368 points); per-launch averages over both. "
                   "A group's figures are sums over its kernels ('parts').",
           "kernels": {}}
    pm = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub in ("pmc_fetch", "pmc_write", "pmc_mfma"):
        p = find(src, sub, "p_counter_collection.csv")
        if not p:
            continue
        for r in csv.DictReader(open(p)):
            c = classify(r["Kernel_Name"])
            if c:
                pm[c][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for g in ("mlp_fwd", "mlp_bwd_dx", "mlp_bwd_dw"):
        parts = sorted({c[1] for c in list(dur) + list(pm) if c[0] == g})
        k = {"parts": {}, "avg_ms": 0.0, "launches_in_trace": 0}
        tot = collections.defaultdict(float)
        for part in parts:
            d = dur.get((g, part), [])
            c = {name: sum(v) / len(v) for name, v in pm.get((g, part), {}).items()}
            k["parts"][part] = {"dispatches": len(d), "avg_ms": sum(d) / max(len(d), 1), "counters_avg_per_dispatch": c}
            k["avg_ms"] += sum(d) / max(len(d), 1)
            k["launches_in_trace"] = max(k["launches_in_trace"], len(d))
            for name, v in c.items():
                tot[name] += v
        k["counters_avg_per_launch"] = dict(tot)
        if "FETCH_SIZE" in tot:
            k["hbm_read_bytes_per_launch"] = 2 * tot["FETCH_SIZE"] * 1024
        if "WRITE_SIZE" in tot:
            k["hbm_write_bytes_per_launch"] = tot["WRITE_SIZE"] * 1024
        if "SQ_VALU_MFMA_BUSY_CYCLES" in tot and tot.get("GRBM_GUI_ACTIVE"):
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles over the 1024 SIMDs
            k["mfma_util"] = tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * tot["GRBM_GUI_ACTIVE"] / 8)
        out["kernels"][g] = k
    out["small_kernels"] = small_kernel_table(find(src, "trace", "t_kernel_trace.csv"))
    json.dump(out, open(os.path.join(dst, "%s_pmc_summary.json" % tag), "w"), indent=1)
    print(json.dumps(out, indent=1)[:3000])


if __name__ == "__main__":
    main()
