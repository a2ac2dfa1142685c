#!/usr/bin/env python3
"""Condense gpurun_out/valu_peak (tools/valu_peak.sh) into profiles/rNN/valu_peak.json.

    python tools/summarize_valu_peak.py gpurun_out/valu_peak profiles/r02/valu_peak.json
"""
import collections
import csv
import glob
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
rows = [json.loads(l) for l in open(os.path.join(src, "valu_peak.jsonl")) if l.startswith("{")]
ops = collections.OrderedDict()
for r in rows:
    o = ops.setdefault(r["op"], {"by_waves_per_simd": {}})
    ghz = r["memtime_ghz"]
    # chip-wide, from the event time: the trustworthy figure (see tools/gen_valu_peak.py)
    o["by_waves_per_simd"][str(r["waves_per_simd"])] = {
        "simd_cycles_per_wave_inst(event)": round(ghz * 1e9 / r["wave_insts_per_s_per_simd(event)"], 3),
        "wave_insts_per_s_per_simd(event)": r["wave_insts_per_s_per_simd(event)"],
        "cycles_per_wave_inst_one_wave(memtime)": r["cycles_per_wave_inst_per_wave"],
        "simd_cycles_per_wave_inst(max wave memtime / W)": round(r["memtime_cycles_max"] / r["wave_insts"] / r["waves_per_simd"], 3),
        "shader_clock_ghz": ghz, "event_ms": r["event_ms"]}
pmc = glob.glob(os.path.join(src, "pmc", "*counter_collection.csv"))
if pmc:
    d = collections.OrderedDict()
    for r in csv.DictReader(open(pmc[0])):
        k = r["Dispatch_Id"]
        e = d.setdefault(k, {"kernel": r["Kernel_Name"], "dur_ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    disp = list(d.values())
    names = list(ops.keys())
    # three launches per op (two warm-ups + the measured one), in op order
    for i, name in enumerate(names):
        if 3 * i + 2 >= len(disp):
            break
        c = disp[3 * i + 2]
        xcd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0
        inst = c.get("SQ_INSTS_VALU", 0.0)
        ops[name]["pmc_waves_per_simd_4"] = {
            "SQ_INSTS_VALU": inst, "GRBM_GUI_ACTIVE_per_xcd": xcd_cycles,
            "wave_insts_per_cycle_per_simd": inst / 1024.0 / xcd_cycles if xcd_cycles else None,
            "simd_cycles_per_wave_inst": xcd_cycles * 1024.0 / inst if inst else None,
            "SQ_ACTIVE_INST_VALU*4/SQ_INSTS_VALU": c["SQ_ACTIVE_INST_VALU"] * 4 / inst if inst else None,
            "clock_ghz": xcd_cycles / c["dur_ns"]}
# classes
def rate(o):
    return o["by_waves_per_simd"]["8"]["simd_cycles_per_wave_inst(event)"]
classes = {"about_2_cycles": [], "about_4_cycles": [], "about_8_cycles": [], "about_16_cycles": [], "other": []}
for n, o in ops.items():
    if n.startswith("mix_") or n.startswith("s_") or n.startswith("ds_"):
        continue
    r = rate(o)
    key = "about_2_cycles" if r < 2.8 else "about_4_cycles" if 3.4 < r < 5.2 else "about_8_cycles" if 7 < r < 9.5 else "about_16_cycles" if 14 < r < 18 else "other"
    classes[key].append("%s (%.2f)" % (n, r))
out = {"what": "VALU issue rate per SIMD on MI355X (gfx950), tools/valu_peak.hip: cycles one SIMD needs per wave64 instruction "
               "(8 waves per SIMD, 8 independent chains per wave; event time over the whole chip at the s_memtime/s_memrealtime clock)",
       "classes(simd cycles per wave64 instruction at 8 waves/SIMD)": classes, "ops": ops}
os.makedirs(os.path.dirname(dst), exist_ok=True)
json.dump(out, open(dst, "w"), indent=1)
for k, v in classes.items():
    print(k, ":", ", ".join(v))
for n, o in ops.items():
    if n.startswith("mix_") or n.startswith("s_") or n.startswith("ds_"):
        print(n, {w: x["simd_cycles_per_wave_inst(event)"] for w, x in o["by_waves_per_simd"].items()})
