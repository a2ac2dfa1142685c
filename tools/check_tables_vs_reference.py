#!/usr/bin/env python3
"""Container-only sanity tool: parse the numeric arrays of the reference's BC7 table
section (ConvectionKernels_BC67.cpp:173-641) and compare them with what
tools/gen_tables.py derives from the format-spec data.  Reads /root/reference; never run
on the GPU box and not part of the test-suite."""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_tables as G

REF = "/root/reference/ConvectionKernels_BC67.cpp"


def grab(src, name):
    m = re.search(r"\b%s\b[^=]*=\s*\{(.*?)\};" % re.escape(name), src, re.S)
    assert m, name
    body = re.sub(r"//[^\n]*", "", m.group(1))
    return [int(t, 0) for t in re.findall(r"0x[0-9a-fA-F]+|\d+", body)]


def main():
    if not os.path.exists(REF):
        print("reference not present; nothing to check")
        return 0
    src = open(REF).read()
    d = G.derive()
    checks = [
        ("g_partitionMap", G.P2), ("g_partitionMap2", G.P3), ("g_fixupIndexes2", G.ANCHOR2),
        ("g_fixupIndexes3", [v for t in G.ANCHOR3 for v in t]),
        ("g_fragments", d["frags"]),
        ("g_shapeRanges", [v for r in d["ranges"] for v in r]),
        ("g_shapes2", [v for t in d["shapes2"] for v in t]),
        ("g_shapes3", [v for t in d["shapes3"] for v in t]),
        ("g_shapeList3", d["list3"]), ("g_shapeList3Short", d["list3short"]),
        ("g_shapeList2", list(range(1, 129))), ("g_shapeList12", list(range(0, 129))),
        ("g_weight2", G.WEIGHTS[2]), ("g_weight3", G.WEIGHTS[3]), ("g_weight4", G.WEIGHTS[4]),
    ]
    bad = 0
    for name, ours in checks:
        ref = grab(src, name)
        ok = ref == list(ours)
        print("%-20s %5d entries  %s" % (name, len(ref), "OK" if ok else "MISMATCH"))
        bad += not ok
    return bad


if __name__ == "__main__":
    sys.exit(main())
