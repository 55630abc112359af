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
    ("mlp_bwd_f16_kernel", "mlp_bwd_dx", "dx"),
    ("grad_absmax_kernel", "mlp_bwd_dx", "absmax"),
    ("mlp_bwd_kernel", "mlp_bwd_dx", "dx"),
    ("mlp_dw_f16_big_kernel", "mlp_bwd_dw", "big"),
    ("mlp_dw_f16_small_kernel", "mlp_bwd_dw", "small"),
    ("mlp_dw_kernel", "mlp_bwd_dw", "dw"),
    ("dw_reduce_kernel", "mlp_bwd_dw", "reduce"),
]


def classify(name):
    for frag, group, part in PARTS:
        if frag in name:
            return group, part
    return None


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
                   "Launches alternate coarse (M = 261 184 points) / fine (M = 783 552 points); per-launch averages over both. "
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
    json.dump(out, open(os.path.join(dst, "%s_pmc_summary.json" % tag), "w"), indent=1)
    print(json.dumps(out, indent=1)[:3000])


if __name__ == "__main__":
    main()
