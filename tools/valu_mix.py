#!/usr/bin/env python3
"""Instruction-class mix of the kernels and what that mix can issue at best (developer / measurement tool).

The guide's VALU peak -- one wave64 instruction per SIMD per 2 cycles -- is a DUAL-ISSUE figure.  Measured on MI355X
(tools/valu_peak.hip, profiles/r02/valu_peak.json: 291 instruction kinds and pairs, 1-8 waves per SIMD): a wave64 VALU instruction
occupies its SIMD for 4.2-4.3 cycles (rcp / sqrt 8.3), and a second instruction OF ANOTHER WAVE shares that slot (2.3 cycles per
instruction) only for certain pairs:
    F   plain v_add / v_sub / v_mul_f32, v_mov_b32                      pairs with F, I and S1
    I   integer add / sub, and / or / xor / not, right shifts          pairs with F and I            (add_u32 + lshl: 4.2)
    S1  fma, min / max, compares, selects, conversions, left shifts,    pairs with F only             (mul_f32 + max_f32: 2.3,
        bfe, perm, three-operand integer ops, med3, rndne, mul_lo        max_f32 + max_f32: 4.3)
    S2  24-bit multiplies, dot, SDWA, DPP, packed, lane access          pairs with nothing            (mul_f32 + mad_i24: 4.2)
    T   rcp / sqrt / rsq                                                8.3 cycles, alone
So the fewest 4.3-cycle slots a mix needs is a matching problem: S1 with F first, then what is left of F with I and itself.

The SQ class counters (SQ_INSTS_VALU_ADD_F32, _MUL_F32, _FMA_F32, _TRANS_F32, _INT32, _CVT; what each counts:
tools/valu_mix_calibrate.sh, profiles/r05/valu_mix_calibration.txt) give the DYNAMIC size of six classes and, by difference, of
"other" (logic, shifts, moves, float compares / min / max, selects, perm, lane access).  How a counter class splits into F / I / S1 /
S2 is taken from the kernel's ISA (static count: the hot loops are unrolled and dominate the text).  That, the perfect interleaving
of the waves the matching assumes, and 2.3 / 4.3 as the only two costs make the result an ESTIMATE of a floor, reported as such.

    python tools/valu_mix.py [lib.so]                  static split per kernel (no GPU)
    valu_mix.floor(dynamic_counters, static_split)     used by tools/summarize_fmt_pmc.py / summarize_pmc.py
"""
import json
import math
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_resources  # noqa: E402

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
SLOT, PAIRED_SLOT, TRANS_SLOT = 4.3, 4.6, 8.3  # cycles: one instruction alone in its slot / two sharing it / rcp, sqrt

TRANS = ("v_rcp_", "v_sqrt_", "v_rsq_", "v_log_", "v_exp_", "v_sin_", "v_cos_")
F_OPS = ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_mov_b32")
I_OPS = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_add_co_u32", "v_sub_co_u32", "v_subrev_co_u32", "v_addc_co_u32", "v_subb_co_u32", "v_add_i32", "v_sub_i32",
         "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_lshrrev_b32", "v_ashrrev_i32")
S2_PREFIX = ("v_mad_i32_i24", "v_mad_u32_u24", "v_mul_i32_i24", "v_mul_u32_u24", "v_mul_hi_", "v_pk_", "v_dot", "v_readlane", "v_readfirstlane", "v_writelane",
             "v_mbcnt", "v_permlane")
INT_CLASS = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_add_co", "v_sub_co", "v_subrev_co", "v_addc_co", "v_subb_co", "v_add_i32", "v_sub_i32", "v_mul_", "v_mad_",
             "v_min_u", "v_min_i", "v_max_u", "v_max_i", "v_med3_i", "v_med3_u", "v_min3_i", "v_min3_u", "v_max3_i", "v_max3_u", "v_add3_", "v_lshl_add", "v_add_lshl",
             "v_bfe_", "v_dot", "v_sad_", "v_mbcnt", "v_cmp_lt_u", "v_cmp_gt_u", "v_cmp_le_u", "v_cmp_ge_u", "v_cmp_eq_u", "v_cmp_ne_u", "v_cmp_lg_u",
             "v_cmp_lt_i", "v_cmp_gt_i", "v_cmp_le_i", "v_cmp_ge_i", "v_cmp_eq_i", "v_cmp_ne_i", "v_cmp_lg_i", "v_cmpx_", "v_xad_")


def classify(op):
    """(SQ counter class, pairing class F / I / S1 / S2 / T) of one VALU mnemonic"""
    base = re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", op)
    modified = op.endswith(("_sdwa", "_dpp"))
    # --- counter class (profiles/r05/valu_mix_calibration.txt) ---
    if base.startswith(("v_pk_add_f32",)) or base in ("v_add_f32", "v_sub_f32", "v_subrev_f32"):
        cls = "ADD_F32"
    elif base.startswith("v_pk_mul_f32") or base in ("v_mul_f32", "v_mul_legacy_f32"):
        cls = "MUL_F32"
    elif base.startswith(("v_fma_f32", "v_fmac_f32", "v_mad_f32", "v_mac_f32", "v_pk_fma_f32")):
        cls = "FMA_F32"
    elif base.startswith(TRANS):
        cls = "TRANS_F32"
    elif base.startswith("v_cvt"):
        cls = "CVT"
    elif base.startswith(INT_CLASS) and not base.startswith(("v_mul_f", "v_mad_f", "v_mul_legacy")):
        cls = "INT32"
    else:
        cls = "OTHER"
    # --- pairing class (profiles/r02/valu_peak.json) ---
    if base.startswith(TRANS):
        pair = "T"
    elif modified or base.startswith(S2_PREFIX):
        pair = "S2"
    elif base in F_OPS:
        pair = "F"
    elif base in I_OPS:
        pair = "I"
    else:
        pair = "S1"
    return cls, pair


def static_split(path=None):
    """{kernel: {counter class: {pairing class: n}}} from the ISA of the library's code objects"""
    path = path or os.path.join(kernel_resources.ROOT, "convectionkernels_amd", "lib", "libcvtt_mi355x.so")
    res = {}
    for obj in kernel_resources.code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(obj)
            f.flush()
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout
        cur = None
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                cur = res.setdefault(m.group(1), {})
                continue
            m = re.match(r"^\s+(v_[a-z0-9_]+)", line)
            if cur is None or not m:
                continue
            cls, pair = classify(m.group(1))
            e = cur.setdefault(cls, {})
            e[pair] = e.get(pair, 0) + 1
    names = [n for n in res if n.startswith("_Z")]
    try:
        dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
        if len(dem) == len(names):
            for n, d in zip(names, dem):
                res[re.sub(r"\(.*", "", d).replace("void ", "")] = res.pop(n)
    except OSError:
        pass
    return {k: v for k, v in res.items() if v}


def slots(n):
    """fewest issue slots for n = {F, I, S1, S2, T} instructions: (alone, shared, trans)"""
    f, i, s1, s2, t = (float(n.get(k, 0.0)) for k in ("F", "I", "S1", "S2", "T"))
    p1 = min(s1, f)                # every S1 that finds an F shares its slot with it
    rest = (f - p1) + i            # what is left of F, and I, pair among themselves
    shared = p1 + rest / 2.0
    alone = (s1 - p1) + s2
    return alone, shared, t


def floor(counters, split):
    """counters: SQ_INSTS_VALU and the six class counters of one dispatch; split: static_split()[kernel].
    Returns the mix and the estimated cycles per VALU instruction it needs when two or more waves interleave perfectly."""
    total = float(counters["SQ_INSTS_VALU"])
    if total <= 0:
        return None
    cls = {k: float(counters.get("SQ_INSTS_VALU_" + k, 0.0)) for k in ("ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "INT32", "CVT")}
    cls["OTHER"] = max(0.0, total - sum(cls.values()))
    default = {"ADD_F32": {"F": 1}, "MUL_F32": {"F": 1}, "FMA_F32": {"S1": 1}, "TRANS_F32": {"T": 1}, "INT32": {"I": 1, "S1": 1, "S2": 1}, "CVT": {"S1": 1},
               "OTHER": {"F": 1, "I": 1, "S1": 2}}
    n = {}
    for name, dyn in cls.items():
        e = (split or {}).get(name) or default[name]
        tot = float(sum(e.values())) or 1.0
        for pair, cnt in e.items():
            n[pair] = n.get(pair, 0.0) + dyn * cnt / tot
    alone, shared, t = slots(n)
    cycles = alone * SLOT + shared * PAIRED_SLOT + t * TRANS_SLOT
    # no S1-F sharing at all (an S1 instruction always alone): the pessimistic end of the bracket
    pess = (n.get("S1", 0.0) + n.get("S2", 0.0)) * SLOT + (n.get("F", 0.0) + n.get("I", 0.0)) / 2.0 * PAIRED_SLOT + t * TRANS_SLOT
    return {"fraction_of_valu_instructions": {k: round(v / total, 4) for k, v in cls.items()},
            "pairing_classes": {k: round(v / total, 4) for k, v in sorted(n.items())},
            "issue_floor_cycles_per_inst": round(cycles / total, 3),
            "issue_floor_without_S1_F_sharing": round(pess / total, 3),
            "two_cycle_peak_cycles_per_inst": 2.0,
            "note": "estimate: dynamic class sizes from SQ_INSTS_VALU_*, their split into pairing classes from the static ISA, slot costs "
                    "4.3 (alone) / 4.6 (two instructions of two waves) / 8.3 (rcp, sqrt) from profiles/r02/valu_peak.json, perfect interleaving assumed"}


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    s = static_split(args[0] if args else None)
    if "--json" in sys.argv:
        print(json.dumps(s, indent=1))
    else:
        for k in sorted(s):
            n = {}
            for e in s[k].values():
                for pair, cnt in e.items():
                    n[pair] = n.get(pair, 0) + cnt
            tot = sum(n.values())
            alone, shared, t = slots(n)
            print("%-58s VALU %6d  %s  static floor %.2f cycles/inst" % (k[:58], tot, " ".join("%s %.2f" % (p, n.get(p, 0) / max(tot, 1)) for p in ("F", "I", "S1", "S2", "T")),
                  (alone * SLOT + shared * PAIRED_SLOT + t * TRANS_SLOT) / max(tot, 1)))
