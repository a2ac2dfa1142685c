#!/usr/bin/env python3
"""Instruction-class mix of the kernels and what that mix can issue at best (developer / measurement tool).

The guide's VALU peak -- one wave64 instruction per SIMD per 2 cycles -- holds for a few instruction kinds only.  Measured on
MI355X with two or more waves per SIMD (tools/valu_peak.hip, profiles/r02/valu_peak.json): plain f32 add / sub / mul, moves,
integer add / sub, and / or / xor, right shifts: 2.3 cycles; fma, min / max, compares, selects, conversions, 24-bit and 32-bit
multiplies, left shifts, three-operand integer ops, v_perm, dot, SDWA / DPP, lane access, packed f32: 4.2-4.3; rcp / sqrt: 8.3.

The SQ class counters (SQ_INSTS_VALU_ADD_F32, _MUL_F32, _FMA_F32, _TRANS_F32, _INT32, _CVT; what each counts:
tools/valu_mix_calibrate.sh, profiles/r05/valu_mix_calibration.txt) give the DYNAMIC size of six classes and, by difference, of
"other" (logic, shifts, moves, float compares / min / max, selects, perm, lane access).  Three of the classes mix 2.3- and
4.3-cycle kinds (ADD/MUL: plain or packed; INT32: add/sub or the rest; other: logic / right shift / move or the rest); their
split is taken from the kernel's ISA (static count: the hot loops are unrolled and dominate the text), which makes the result an
ESTIMATE, reported as such:  floor = sum(class size x cycles of its kinds) / SQ_INSTS_VALU  cycles per instruction.

    python tools/valu_mix.py [lib.so]                  static split per kernel (no GPU)
    valu_mix.floor(dynamic_counters, static_split)     used by tools/summarize_fmt_pmc.py / summarize_pmc.py
"""
import json
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_resources  # noqa: E402

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
C2, C4, C8 = 2.3, 4.3, 8.3  # cycles per wave64 instruction, two or more waves per SIMD (profiles/r02/valu_peak.json)

TRANS = ("v_rcp_", "v_sqrt_", "v_rsq_", "v_log_", "v_exp_", "v_sin_", "v_cos_")
INT_FAST = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_add_co_u32", "v_sub_co_u32", "v_subrev_co_u32", "v_addc_co_u32", "v_subb_co_u32", "v_add_i32", "v_sub_i32")
INT_SLOW = ("v_mul_", "v_mad_", "v_min_", "v_max_", "v_med3_", "v_min3_", "v_max3_", "v_add3_", "v_lshl_add", "v_add_lshl", "v_bfe_", "v_dot", "v_sad_", "v_mbcnt",
            "v_cmp_", "v_cmpx_", "v_xad_")
OTHER_FAST = ("v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_mov_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_bfrev")


def classify(op):
    """(pmc class, fast?) of one VALU mnemonic"""
    base = re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", op)
    modified = op.endswith(("_sdwa", "_dpp"))
    if base.startswith("v_pk_add_f32") or base.startswith("v_pk_mul_f32"):
        return ("ADD_F32" if "add" in base else "MUL_F32"), False
    if base.startswith("v_pk_fma_f32"):
        return "FMA_F32", False
    if base in ("v_add_f32", "v_sub_f32", "v_subrev_f32"):
        return "ADD_F32", not modified
    if base == "v_mul_f32" or base == "v_mul_legacy_f32":
        return "MUL_F32", not modified
    if base.startswith(("v_fma_f32", "v_fmac_f32", "v_mad_f32", "v_mac_f32", "v_fma_mix")):
        return "FMA_F32", False
    if base.startswith(TRANS):
        return "TRANS_F32", False
    if base.startswith("v_cvt_") or base.startswith("v_cvt"):
        return "CVT", False
    if base.endswith(("_f32", "_f16", "_f64")) or "_f32_" in base:  # float min / max / compares / rndne / med3 ...
        return "OTHER", False
    if base in INT_FAST:
        return "INT32", not modified
    if base.startswith(INT_SLOW):
        return "INT32", False
    if base in OTHER_FAST:
        return "OTHER", not modified
    return "OTHER", False  # shifts left, selects, perm, bfi, lane access, dpp moves, ...


def static_split(path=None):
    """{kernel: {class: {"fast": n, "slow": n}}} from the ISA of the library's code objects"""
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
            cls, fast = classify(m.group(1))
            e = cur.setdefault(cls, {"fast": 0, "slow": 0})
            e["fast" if fast else "slow"] += 1
    names = [n for n in res if n.startswith("_Z")]
    try:
        dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
        if len(dem) == len(names):
            for n, d in zip(names, dem):
                res[re.sub(r"\(.*", "", d).replace("void ", "")] = res.pop(n)
    except OSError:
        pass
    return {k: v for k, v in res.items() if v}


def floor(counters, split):
    """counters: SQ_INSTS_VALU and the six class counters of one dispatch; split: static_split()[kernel].
    Returns the mix and the estimated cycles per VALU instruction this mix needs at full overlap."""
    total = float(counters["SQ_INSTS_VALU"])
    if total <= 0:
        return None
    cls = {k: float(counters.get("SQ_INSTS_VALU_" + k, 0.0)) for k in ("ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "INT32", "CVT")}
    cls["OTHER"] = max(0.0, total - sum(cls.values()))

    def fast_frac(name):
        e = (split or {}).get(name) or {}
        n = e.get("fast", 0) + e.get("slow", 0)
        return e.get("fast", 0) / n if n else 0.5

    cycles = cls["FMA_F32"] * C4 + cls["CVT"] * C4 + cls["TRANS_F32"] * C8
    lo = hi = cycles
    for name in ("ADD_F32", "MUL_F32", "INT32", "OTHER"):
        f = fast_frac(name)
        cycles += cls[name] * (f * C2 + (1.0 - f) * C4)
        lo += cls[name] * C2
        hi += cls[name] * C4
    return {"fraction_of_valu_instructions": {k: round(v / total, 4) for k, v in cls.items()},
            "static_fast_fraction_within_class": {k: round(fast_frac(k), 3) for k in ("ADD_F32", "MUL_F32", "INT32", "OTHER")},
            "issue_floor_cycles_per_inst": round(cycles / total, 3),
            "issue_floor_bracket": [round(lo / total, 3), round(hi / total, 3)],
            "note": "estimate: dynamic class sizes (SQ_INSTS_VALU_*), the 2.3 / 4.3-cycle split inside the mixed classes from the static ISA; "
                    "bracket = all of them fast / all slow; costs from profiles/r02/valu_peak.json"}


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    s = static_split(args[0] if args else None)
    if "--json" in sys.argv:
        print(json.dumps(s, indent=1))
    else:
        for k in sorted(s):
            tot = sum(e["fast"] + e["slow"] for e in s[k].values())
            fast = sum(e["fast"] for e in s[k].values())
            print("%-60s VALU %6d  2.3-cycle kinds %.2f  %s" % (k[:60], tot, fast / max(tot, 1),
                  " ".join("%s %d/%d" % (c, e["fast"], e["fast"] + e["slow"]) for c, e in sorted(s[k].items()))))
