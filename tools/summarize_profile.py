#!/usr/bin/env python
"""Condenses a rocprofv3 output directory (kernel-trace stats + separate PMC passes, as produced
by the commands in profiles/README.md) into the small files committed under profiles/.

    python tools/summarize_profile.py gpurun_out/prof_r1 profiles r01
"""
import collections
import csv
import json
import os
import shutil
import sys


def short(name):
    if "mlp_fwd_kernel" in name:
        return "mlp_fwd"
    if "mlp_bwd_kernel" in name:
        return "mlp_bwd_dx"
    if "mlp_dw_kernel" in name:
        return "mlp_bwd_dw"
    return None


def main():
    src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, "trace", "t_kernel_stats.csv"), os.path.join(dst, "%s_kernel_stats.csv" % tag))
    # per-dispatch durations of the MLP kernels (kernel-trace pass)
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(src, "trace", "t_kernel_trace.csv"))):
        s = short(r["Kernel_Name"])
        if s:
            dur[s].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    out = {"note": "FETCH_SIZE / WRITE_SIZE are rocprofv3 KB units; on gfx950 FETCH_SIZE counts 64 B per 128-B request for "
                   "wide coalesced streaming reads (MI355X_MICROARCH.md, HBM) => fetch_bytes_corrected = 2 x FETCH_SIZE x 1024. "
                   "Launches alternate coarse (M = 261 184 points) / fine (M = 522 368 points); per-launch averages over both.",
           "kernels": {}}
    pm = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub in ("pmc_fetch", "pmc_write", "pmc_mfma"):
        p = os.path.join(src, sub, "p_counter_collection.csv")
        if not os.path.exists(p):
            continue
        for r in csv.DictReader(open(p)):
            s = short(r["Kernel_Name"])
            if s:
                pm[s][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for s in ("mlp_fwd", "mlp_bwd_dx", "mlp_bwd_dw"):
        k = {"launches_in_trace": len(dur[s]), "avg_ms": sum(dur[s]) / max(len(dur[s]), 1)}
        c = {name: sum(v) / len(v) for name, v in pm[s].items()}
        k["counters_avg_per_launch"] = c
        if "FETCH_SIZE" in c:
            k["hbm_read_bytes_per_launch"] = 2 * c["FETCH_SIZE"] * 1024
        if "WRITE_SIZE" in c:
            k["hbm_write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles over the 1024 SIMDs
            k["mfma_util"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * c["GRBM_GUI_ACTIVE"] / 8)
            k["avg_clock_ghz_profiled"] = None
        out["kernels"][s] = k
    json.dump(out, open(os.path.join(dst, "%s_pmc_summary.json" % tag), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
