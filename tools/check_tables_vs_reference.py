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
    bad += check_single_colour()
    bad += check_s3tc_single_colour()
    bad += check_fake709()
    return bad


def check_single_colour():
    """BC7 single-colour tables (ConvectionKernels_BC7_SingleColor.h) vs tools/gen_bc7_single_color.py"""
    import gen_bc7_single_color as SC
    from bc7_sc_overrides import OVERRIDES
    path = "/root/reference/ConvectionKernels_BC7_SingleColor.h"
    txt = open(path).read()
    ref = {}
    for m in re.finditer(r"Table (\w+)=\s*\{\s*(\d+),\s*(\d+),\s*\{(.*?)\}\s*\};", txt, re.S):
        nums = [int(x) for x in re.findall(r"\d+", m.group(4))]
        ref[m.group(1)] = (int(m.group(2)), int(m.group(3)), [tuple(nums[i:i + 3]) for i in range(0, 768, 3)])
    names = []
    for p in ("p00", "p01", "p10", "p11"):
        names += ["g_mode0_%s_i%d" % (p, i) for i in (1, 2, 3)]
    for p in ("p0", "p1"):
        names += ["g_mode1_%s_i%d" % (p, i) for i in (1, 2, 3)]
    names += ["g_mode2", "g_mode3_p0", "g_mode3_p1"]
    for p in ("p0", "p1"):
        names += ["g_mode6_%s_i%d" % (p, i) for i in range(1, 8)]
    names += ["g_mode7_p00", "g_mode7_p01", "g_mode7_p10", "g_mode7_p11"]
    bad = 0
    for n, (mode, idx, pb, ent) in zip(names, SC.build(OVERRIDES)):
        ok = ref[n] == (idx, pb, ent)
        bad += not ok
        if not ok:
            print("single-colour table %s MISMATCH" % n)
    print("%-20s %5d tables   %s" % ("BC7 single colour", len(names), "OK" if not bad else "MISMATCH"))
    return bad


def check_s3tc_single_colour():
    """BC1-family single-colour tables (ConvectionKernels_S3TC_SingleColor.h) vs tools/gen_s3tc_single_color.py"""
    import gen_s3tc_single_color as SC
    txt = open("/root/reference/ConvectionKernels_S3TC_SingleColor.h").read()
    names = ["g_singleColor5_3", "g_singleColor6_3", "g_singleColor5_2", "g_singleColor6_2",
             "g_singleColor5_3_p", "g_singleColor6_3_p", "g_singleColor5_2_p", "g_singleColor6_2_p"]
    bad = 0
    for n, ours in zip(names, SC.build()):
        m = re.search(r"\b%s\[256\]\s*=\s*\{(.*?)\};" % n, txt, re.S)
        nums = [int(x) for x in re.findall(r"\d+", m.group(1))]
        ref = [tuple(nums[i:i + 4]) for i in range(0, 1024, 4)]
        ok = ref == ours
        bad += not ok
        if not ok:
            print("S3TC single-colour table %s MISMATCH" % n)
    print("%-20s %5d tables   %s" % ("S3TC single colour", len(names), "OK" if not bad else "MISMATCH"))
    return bad


def check_fake709():
    """ETC fake-BT.709 rounding table (ConvectionKernels_FakeBT709_Rounding.h) vs tools/gen_fake709_rounding.py"""
    import gen_fake709_rounding as F
    txt = open("/root/reference/ConvectionKernels_FakeBT709_Rounding.h").read()
    m = re.search(r"g_rounding16\[\]\s*=\s*\{(.*?)\};", txt, re.S)
    ref = [int(x) for x in re.findall(r"\d+", m.group(1))]
    ok = ref == F.build()
    print("%-20s %5d entries  %s" % ("fake BT.709 rounding", len(ref), "OK" if ok else "MISMATCH"))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
