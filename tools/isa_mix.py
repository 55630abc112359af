#!/usr/bin/env python
"""Static instruction mix of the kernels in a gfx950 assembly file (hipcc -S --cuda-device-only): per kernel, and per
loop-free region between two s_barrier instructions when --regions is given.  Used to see what an epilogue costs in issue slots
next to the MFMAs of its K-loop ('Instruction mix')."""
import collections
import re
import sys


def classify(op):
    if "mfma" in op:
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else None
    top = "--top" in sys.argv
    lines = open(path).read().split("\n")
    starts = [(i, l) for i, l in enumerate(lines) if re.match(r"^_Z[^ ]*:", l)]
    for (i, name), (j, _) in zip(starts, starts[1:] + [(len(lines), "")]):
        if want and want not in name:
            continue
        c, ops = collections.Counter(), collections.Counter()
        for x in lines[i:j]:
            m = re.match(r"^\s+([a-z_0-9]+)(\s|$)", x)
            if m and not m.group(1).startswith("."):
                c[classify(m.group(1))] += 1
                ops[m.group(1)] += 1
        if not c["mfma"]:
            continue
        print("%s\n   mfma %d  valu %d (%.2f / mfma)  lds %d (%.2f)  vmem %d (%.2f)  salu %d (%.2f)" % (
            name[:100], c["mfma"], c["valu"], c["valu"] / c["mfma"], c["lds"], c["lds"] / c["mfma"], c["vmem"], c["vmem"] / c["mfma"],
            c["salu"], c["salu"] / c["mfma"]))
        if top:
            print("   " + "  ".join("%s %d" % kv for kv in ops.most_common(40)))


if __name__ == "__main__":
    main()
