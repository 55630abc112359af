#!/usr/bin/env python
"""Per-launch timeline of ONE iteration of the reference-shaped training loop (tools/dropin_driver.py) out of a rocprofv3 kernel
trace, cut into the sections a reader of train.py knows:

    render     first MLP forward launch .. last compositing launch of graph.forward
    loss       train.py:163-337 - the loss lines on torch tensors, their `.item()` reads, and the autograd backward of those lines
               (everything between the render's last forward launch and its first backward launch)
    backward   first compositing-backward launch .. the last launch of loss.backward()
    optimise+front   the optimiser steps, then the NEXT iteration's graph.forward up to its first MLP launch (event window, pixel
               draws, sampling draws - on a second stream, started while the backward launches above still run -, trajectory, rays)

    rocprofv3 --kernel-trace --output-format csv -d OUT -o t -- python tools/dropin_driver.py --steps 6 --warmup 3
    python tools/dropin_timeline.py OUT/**/t_kernel_trace.csv [--json] [--step-from-end 2]

A step is cut at its first MLP forward launch.  `busy` = union of the launch
intervals, `idle` = span - busy: device time with nothing to run, i.e. the host (python / autograd) deciding what to launch next.
"""
import collections
import csv
import json
import re
import sys


def load(path):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r.get("Stream_Id", 0) or 0)))
    rows.sort(key=lambda r: r[1])
    return rows


def short(name):
    m = re.search(r"(\w+)(<.*>)?\(", name)
    return (m.group(1) if m else name)[:44]


def union(iv):
    iv = sorted(iv)
    if not iv:
        return 0
    tot, (cs, ce) = 0, iv[0]
    for s, e in iv[1:]:
        if s > ce:
            tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return tot + ce - cs


K3 = ("mlp_fwd", "mlp_bwd", "mlp_dw", "dw_reduce", "dw_compose")


def first_fwd_of_steps(rows):
    """index of the first MLP forward launch of every iteration (the coarse one: the forward launch that follows a launch of the
    backward side or the start of the trace)"""
    out, seen_bwd = [], True
    for i, r in enumerate(rows):
        n = short(r[0])
        if n.startswith("mlp_fwd") and seen_bwd:
            out.append(i)
            seen_bwd = False
        elif n.startswith(("mlp_bwd", "mlp_dw")):
            seen_bwd = True
    return out


def main():
    path = sys.argv[1]
    want_json = "--json" in sys.argv
    back = 2
    if "--step-from-end" in sys.argv:
        back = int(sys.argv[sys.argv.index("--step-from-end") + 1])
    rows = load(path)
    starts = first_fwd_of_steps(rows)
    if len(starts) < back + 1:
        sys.exit("not enough steps in the trace")
    per_step = [starts[k + 1] - starts[k] for k in range(len(starts) - 1)]
    seg = rows[starts[-back - 1]:starts[-back]]
    t0 = seg[0][1]
    t_end = rows[starts[-back]][1]
    names = [short(r[0]) for r in seg]
    i_cf = max(i for i, n in enumerate(names) if n.startswith("composite_fwd"))
    i_cb = next(i for i, n in enumerate(names) if n.startswith("composite_bwd"))
    i_bw = max(i for i, n in enumerate(names) if n.startswith(("spline_bwd", "rays_bwd", "dw_compose", "dw_reduce")))
    cuts = [("render", 0, i_cf + 1), ("loss", i_cf + 1, i_cb), ("backward", i_cb, i_bw + 1), ("optimise+front", i_bw + 1, len(seg))]
    summ = {"launches_per_step": per_step[-back], "launches_per_step_all": per_step, "span_ms": round((t_end - t0) / 1e6, 3), "sections": {}}
    bounds = {}
    for name, a, b in cuts:
        part = seg[a:b]
        if not part:
            summ["sections"][name] = {"launches": 0, "span_ms": 0.0, "busy_ms": 0.0, "idle_ms": 0.0}
            continue
        s0 = seg[a][1] if a else t0
        s1 = seg[b][1] if b < len(seg) else t_end
        # busy: every launch that overlaps the section's window counts (launches of the second stream started earlier included)
        busy = union([(max(r[1], s0), min(r[2], s1)) for r in seg if min(r[2], s1) > max(r[1], s0)])
        k3 = sum(r[2] - r[1] for r in part if short(r[0]).startswith(K3))
        summ["sections"][name] = {"launches": len(part), "span_ms": round((s1 - s0) / 1e6, 3), "busy_ms": round(busy / 1e6, 3),
                                  "idle_ms": round((s1 - s0 - busy) / 1e6, 3), "k3_ms": round(k3 / 1e6, 3)}
        bounds[name] = (a, b)
    if want_json:
        print(json.dumps(summ))
        return
    print("# one iteration of tools/dropin_driver.py (train.py:153-394 on the drop-in modules), rocprofv3 kernel trace")
    print("# %d launches, %.3f ms from this iteration's first launch to the next one's (profiled: ~4 us per launch slower than un-profiled)"
          % (len(seg), summ["span_ms"]))
    for name, _, _ in cuts:
        v = summ["sections"][name]
        print("#   %-9s %4d launches  span %7.3f ms  device busy %7.3f ms (K3 %6.3f)  idle %7.3f ms" % (name, v["launches"], v["span_ms"], v["busy_ms"], v.get("k3_ms", 0.0), v["idle_ms"]))
    print("#\n#  start us   dur us   gap us  stream  kernel       (runs of launches shorter than 40 us with gaps below 40 us are folded)")
    prev_end = t0
    cnt, small, gapsum, names = 0, 0.0, 0.0, collections.Counter()
    cur = None

    def flush():
        nonlocal cnt, small, gapsum
        if cnt:
            print("           ... %d small launches, %.0f us busy, %.0f us of gaps: %s"
                  % (cnt, small, gapsum, ", ".join("%s x%d" % (k[:30], v) for k, v in names.most_common(7))))
        cnt, small, gapsum = 0, 0.0, 0.0
        names.clear()

    sec_at = {a: name for name, (a, b) in bounds.items()}
    for i, (n, s, e, st) in enumerate(seg):
        if i in sec_at:
            flush()
            print("# ---- %s" % sec_at[i])
        gap, dur = (s - prev_end) / 1e3, (e - s) / 1e3
        if dur > 40 or gap > 40:
            flush()
            print("%10.1f %8.1f %8.1f  s%-5d %s" % ((s - t0) / 1e3, dur, gap, st, short(n)))
        else:
            cnt += 1
            small += dur
            gapsum += max(gap, 0.0)
            names[short(n)] += 1
        prev_end = max(prev_end, e)
    flush()


if __name__ == "__main__":
    main()
