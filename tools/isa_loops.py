#!/usr/bin/env python
"""Instruction mix of every loop body (label .. backward branch) of one kernel in a gfx950 assembly file:
    python tools/isa_loops.py file.s <kernel-name-substring>
The layer loops of the K3 kernels are `#pragma unroll 1` loops whose bodies hold a whole stage (K-loop + epilogue), so their static
mix IS the dynamic mix of a layer."""
import collections
import re
import sys


def main():
    path, want = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = [i for i, l in enumerate(lines) if re.match(r"^_Z[^ ]*:", l) and want in l][0]
    end = [i for i, l in enumerate(lines) if i > start and "s_endpgm" in l][0]
    labels = {}
    for i in range(start, end):
        m = re.match(r"^(\.LBB\d+_\d+):", lines[i])
        if m:
            labels[m.group(1)] = i
    seen = set()
    for i in range(start, end):
        m = re.match(r"^\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", lines[i])
        if m and m.group(1) in labels and labels[m.group(1)] < i and m.group(1) not in seen:
            seen.add(m.group(1))
            a = labels[m.group(1)]
            c = collections.Counter()
            for x in lines[a:i]:
                mm = re.match(r"^\s+([a-z_0-9]+)(\s|$)", x)
                if mm:
                    c[mm.group(1)] += 1
            mf = sum(v for k, v in c.items() if "mfma" in k)
            valu = sum(v for k, v in c.items() if k.startswith("v_") and "mfma" not in k)
            lds = sum(v for k, v in c.items() if k.startswith("ds_"))
            vm = sum(v for k, v in c.items() if k.startswith(("buffer_", "global_", "scratch_")))
            print("loop %s lines %d-%d: mfma %d valu %d (%.2f/mfma) lds %d vmem %d salu %d" % (
                m.group(1), a, i, mf, valu, valu / max(mf, 1), lds, vm, sum(v for k, v in c.items() if k.startswith("s_"))))
            if mf >= 32 or "--all" in sys.argv:
                print("   " + "  ".join("%s %d" % (k, v) for v, k in sorted([(v, k) for k, v in c.items() if not k.startswith("s_")], reverse=True)[:36]))


if __name__ == "__main__":
    main()
