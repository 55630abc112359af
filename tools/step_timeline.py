#!/usr/bin/env python
"""Timeline of ONE training step out of a rocprofv3 kernel trace (csv: `--kernel-trace --output-format csv`, or the rocpd .db the
default output writes): every launch with its start relative to the step's first kernel, duration, stream and workgroup
count; then span, busy time (union of the launch intervals), idle time and the per-kernel sums.  A step = the launches
between two step_gate_kernel launches (the device-side range gate closes every TrainStep.step).

    python tools/step_timeline.py gpurun_out/prof/t_kernel_trace.csv [step-from-the-end, default 2] [--quiet]
"""
import collections
import csv
import re
import sqlite3
import sys


def load(path):
    """[(name, start_ns, end_ns, stream, workgroups)] sorted by start."""
    rows = []
    if path.endswith(".db"):
        c = sqlite3.connect(path)
        for n, s, e, st, g, w in c.execute("select name, start, end, stream_id, grid_x, workgroup_x from kernels"):
            rows.append((n, s, e, st, g // max(w, 1)))
    else:
        for r in csv.DictReader(open(path)):
            rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r.get("Stream_Id", 0) or 0),
                         int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)))
    rows.sort(key=lambda r: r[1])
    return rows


def short(name):
    m = re.search(r"(\w+)(<.*>)?\(", name)
    return (m.group(1) if m else name)[:30]


def union(iv):
    iv = sorted(iv)
    tot, (cs, ce) = 0, iv[0]
    for s, e in iv[1:]:
        if s > ce:
            tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return tot + ce - cs


def main():
    path = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else 2
    quiet = "--quiet" in sys.argv
    rows = load(path)
    gates = [i for i, r in enumerate(rows) if "step_gate" in r[0]]
    # single-rank steps gate once; keep the LAST gate of a run of adjacent ones (data-parallel steps gate twice)
    gates = [g for k, g in enumerate(gates) if k + 1 == len(gates) or gates[k + 1] - g > 4]
    spans = []
    for k in range(len(gates) - 1):
        seg = rows[gates[k]:gates[k + 1]]
        t0, t1 = seg[0][1], max(r[2] for r in seg)
        busy = union([(r[1], r[2]) for r in seg])
        spans.append(((t1 - t0) / 1e6, busy / 1e6, sum(r[2] - r[1] for r in seg) / 1e6, len(seg)))
    print("steps in trace: %d   (span ms, busy ms, sum of kernel ms, launches), last 5:" % len(spans))
    for s in spans[-5:]:
        print("   span %.3f  busy %.3f  idle %.3f  sum %.3f  launches %d" % (s[0], s[1], s[0] - s[1], s[2], s[3]))
    seg = rows[gates[-back - 1]:gates[-back]]
    t0 = seg[0][1]
    if not quiet:
        print("%10s %9s  %-3s %-30s %s" % ("start us", "dur us", "str", "kernel", "workgroups"))
        for r in seg:
            print("%10.1f %9.1f  s%-2d %-30s %d" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], short(r[0]), r[4]))
    agg = collections.OrderedDict()
    for r in seg:
        a = agg.setdefault(short(r[0]), [0, 0])
        a[0] += 1
        a[1] += r[2] - r[1]
    print("per kernel (launches, total us):")
    for n, (k, t) in sorted(agg.items(), key=lambda x: -x[1][1]):
        print("   %-30s %3d %9.1f" % (n, k, t / 1e3))


if __name__ == "__main__":
    main()
