#!/usr/bin/env python3
"""Regenerate every constant table the BC7/BC1 paths need, from the block-format
specification data alone (no text is taken from the reference's table files).

Inputs (format-spec constants, BC7 a.k.a. BPTC):
  * the 64 two-subset partition bitmaps and the 64 three-subset partition maps,
  * the anchor ("fix-up") pixel of the 2nd/3rd subset of every partition,
  * interpolation weights for 2/3/4-bit indices.

Derived here (rules verified against the reference by tools/check_tables_vs_reference.py,
SURVEY.md App. G; reference arrays: ConvectionKernels_BC67.cpp:255-649):
  * the 243 distinct "shapes" (pixel subsets): [all 16] + sorted(two-subset masks and
    their complements) + sorted(three-subset masks not already present),
  * per shape the ascending pixel list, (offset,length) into the concatenated list,
  * per partition the shape ids of its subsets, and the per-mode shape lists.
  Shape numbering is part of the API: BC7EncodingPlan arrays are indexed by it
  (ConvectionKernels.h:157-164).

Writes:  oracle/cvtt_oracle_tables.h               (checker)
         convectionkernels_amd/csrc/bc7_tables.h   (product; host + device)
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# --- format-spec data -------------------------------------------------------------
# Two-subset partitions: bit p set <=> pixel p (row-major) belongs to subset 1.
P2 = [
    0xCCCC, 0x8888, 0xEEEE, 0xECC8, 0xC880, 0xFEEC, 0xFEC8, 0xEC80,
    0xC800, 0xFFEC, 0xFE80, 0xE800, 0xFFE8, 0xFF00, 0xFFF0, 0xF000,
    0xF710, 0x008E, 0x7100, 0x08CE, 0x008C, 0x7310, 0x3100, 0x8CCE,
    0x088C, 0x3110, 0x6666, 0x366C, 0x17E8, 0x0FF0, 0x718E, 0x399C,
    0xAAAA, 0xF0F0, 0x5A5A, 0x33CC, 0x3C3C, 0x55AA, 0x9696, 0xA55A,
    0x73CE, 0x13C8, 0x324C, 0x3BDC, 0x6996, 0xC33C, 0x9966, 0x0660,
    0x0272, 0x04E4, 0x4E40, 0x2720, 0xC936, 0x936C, 0x39C6, 0x639C,
    0x9336, 0x9CC6, 0x817E, 0xE718, 0xCCF0, 0x0FCC, 0x7744, 0xEE22,
]

# Three-subset partitions: bits [2p+1:2p] = subset of pixel p.
P3 = [
    0xAA685050, 0x6A5A5040, 0x5A5A4200, 0x5450A0A8, 0xA5A50000, 0xA0A05050, 0x5555A0A0, 0x5A5A5050,
    0xAA550000, 0xAA555500, 0xAAAA5500, 0x90909090, 0x94949494, 0xA4A4A4A4, 0xA9A59450, 0x2A0A4250,
    0xA5945040, 0x0A425054, 0xA5A5A500, 0x55A0A0A0, 0xA8A85454, 0x6A6A4040, 0xA4A45000, 0x1A1A0500,
    0x0050A4A4, 0xAAA59090, 0x14696914, 0x69691400, 0xA08585A0, 0xAA821414, 0x50A4A450, 0x6A5A0200,
    0xA9A58000, 0x5090A0A8, 0xA8A09050, 0x24242424, 0x00AA5500, 0x24924924, 0x24499224, 0x50A50A50,
    0x500AA550, 0xAAAA4444, 0x66660000, 0xA5A0A5A0, 0x50A050A0, 0x69286928, 0x44AAAA44, 0x66666600,
    0xAA444444, 0x54A854A8, 0x95809580, 0x96969600, 0xA85454A8, 0x80959580, 0xAA141414, 0x96960000,
    0xAAAA1414, 0xA05050A0, 0xA0A5A5A0, 0x96000000, 0x40804080, 0xA9A8A9A8, 0xAAAAAA44, 0x2A4A5254,
]

# Anchor pixel of subset 1 for two-subset partitions.
ANCHOR2 = [
    15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15,
    15, 2, 8, 2, 2, 8, 8, 15, 2, 8, 2, 2, 8, 8, 2, 2,
    15, 15, 6, 8, 2, 8, 15, 15, 2, 8, 2, 2, 2, 15, 15, 6,
    6, 2, 6, 8, 15, 15, 2, 2, 15, 15, 15, 15, 15, 2, 2, 15,
]

# Anchor pixels of subsets 1 and 2 for three-subset partitions.
ANCHOR3 = [
    (3, 15), (3, 8), (15, 8), (15, 3), (8, 15), (3, 15), (15, 3), (15, 8),
    (8, 15), (8, 15), (6, 15), (6, 15), (6, 15), (5, 15), (3, 15), (3, 8),
    (3, 15), (3, 8), (8, 15), (15, 3), (3, 15), (3, 8), (6, 15), (10, 8),
    (5, 3), (8, 15), (8, 6), (6, 10), (8, 15), (5, 15), (15, 10), (15, 8),
    (8, 15), (15, 3), (3, 15), (5, 10), (6, 10), (10, 8), (8, 9), (15, 10),
    (15, 6), (3, 15), (15, 8), (5, 15), (15, 3), (15, 6), (15, 6), (15, 8),
    (3, 15), (15, 3), (5, 15), (5, 15), (5, 15), (8, 15), (5, 15), (10, 15),
    (5, 15), (10, 15), (8, 15), (13, 15), (15, 3), (12, 15), (3, 15), (3, 8),
]

WEIGHTS = {2: [0, 21, 43, 64], 3: [0, 9, 18, 27, 37, 46, 55, 64],
           4: [0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64]}


def subset_masks3(code):
    m = [0, 0, 0]
    for p in range(16):
        m[(code >> (2 * p)) & 3] |= 1 << p
    return m


def derive():
    m2 = set()
    for bits in P2:
        m2.add(bits)
        m2.add(~bits & 0xFFFF)
    m3 = set()
    for code in P3:
        for m in subset_masks3(code):
            m3.add(m)
    shapes = [0xFFFF] + sorted(m2) + sorted(m3 - m2)
    assert len(shapes) == 243 and len(m2) == 128
    shape_id = {m: i for i, m in enumerate(shapes)}
    frags, ranges = [], []
    for m in shapes:
        px = [p for p in range(16) if (m >> p) & 1]
        ranges.append((len(frags), len(px)))
        frags.extend(px)
    assert len(frags) == 1612
    shapes2 = [(shape_id[~b & 0xFFFF], shape_id[b]) for b in P2]
    shapes3 = [tuple(shape_id[m] for m in subset_masks3(c)) for c in P3]
    list3 = sorted({s for t in shapes3 for s in t})
    list3short = sorted({s for t in shapes3[:16] for s in t})
    assert len(list3) == 140 and len(list3short) == 36
    return dict(shapes=shapes, frags=frags, ranges=ranges, shapes2=shapes2, shapes3=shapes3,
                list3=list3, list3short=list3short)


def carr(ctype, name, vals, per_line=12, fmt="%d", qual="static const"):
    out = ["%s %s %s[%d] = {" % (qual, ctype, name, len(vals))]
    for i in range(0, len(vals), per_line):
        out.append("    " + ", ".join(fmt % v for v in vals[i:i + per_line]) + ",")
    out.append("};")
    return "\n".join(out)


def emit(prefix, guard, qual, banner):
    d = derive()
    L = ["// GENERATED by tools/gen_tables.py -- do not edit.", "// " + banner,
         "#ifndef %s" % guard, "#define %s" % guard, "#include <stdint.h>", ""]
    L.append("#define %sNUM_SHAPES 243" % prefix.upper())
    L.append("#define %sNUM_FRAGMENTS 1612" % prefix.upper())
    L.append("")
    L.append(carr("uint16_t", prefix + "partition2", P2, 8, "0x%04x", qual))
    L.append(carr("uint32_t", prefix + "partition3", P3, 4, "0x%08xu", qual))
    L.append(carr("uint8_t", prefix + "anchor2", ANCHOR2, 16, "%d", qual))
    L.append(carr("uint8_t", prefix + "anchor3", [v for t in ANCHOR3 for v in t], 16, "%d", qual))
    L.append(carr("uint16_t", prefix + "shape_mask", d["shapes"], 8, "0x%04x", qual))
    L.append(carr("uint8_t", prefix + "fragments", d["frags"], 24, "%d", qual))
    L.append(carr("uint16_t", prefix + "shape_start", [r[0] for r in d["ranges"]], 16, "%d", qual))
    L.append(carr("uint8_t", prefix + "shape_len", [r[1] for r in d["ranges"]], 24, "%d", qual))
    L.append(carr("uint8_t", prefix + "shapes2", [v for t in d["shapes2"] for v in t], 16, "%d", qual))
    L.append(carr("uint8_t", prefix + "shapes3", [v for t in d["shapes3"] for v in t], 15, "%d", qual))
    L.append(carr("uint8_t", prefix + "shape_list3", d["list3"], 20, "%d", qual))
    L.append(carr("uint8_t", prefix + "shape_list3_short", d["list3short"], 20, "%d", qual))
    for bits, w in WEIGHTS.items():
        L.append(carr("uint8_t", prefix + "weights%d" % bits, w, 16, "%d", qual))
    L.append("")
    L.append("#endif")
    return "\n".join(L) + "\n"


def main():
    with open(os.path.join(ROOT, "oracle", "cvtt_oracle_tables.h"), "w") as f:
        f.write(emit("orc_", "CVTT_ORACLE_TABLES_H", "static const",
                     "TEST INFRASTRUCTURE: tables for the CPU oracle."))
    with open(os.path.join(ROOT, "convectionkernels_amd", "csrc", "bc7_tables.h"), "w") as f:
        f.write(emit("k_", "CVTTMI_BC7_TABLES_H", "static const",
                     "BC7 partition/shape tables for the host shim (uploaded to HBM at init)."))


if __name__ == "__main__":
    main()
