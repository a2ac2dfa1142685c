#!/usr/bin/env python3
"""Condense the rocprofv3 CSVs written by tools/profile_round.sh into one small JSON summary (the raw
counter_collection CSVs have one row per counter per dispatch).

Per PMC pass and kernel: the LAST dispatch (the run does 5 warm-ups and 20 timed steps, so that one is warm) with its
counters, and `mean_last20` = the counters averaged over the last 20 dispatches.  `headline` = the entry of the dominant
kernel (largest total duration); its derived figures and the corrected HBM traffic are what bench.py quotes."""
import csv
import glob
import json
import os
import sys

out_dir = sys.argv[1]
summary = {}


def load(path):
    rows = list(csv.DictReader(open(path)))
    return [r for r in rows if "cvttmi" in r["Kernel_Name"]]


def per_kernel(rows):
    by_disp = {}
    for r in rows:
        d = by_disp.setdefault(int(r["Dispatch_Id"]), {"kernel": r["Kernel_Name"].split("(")[0], "vgpr": r["VGPR_Count"], "sgpr": r["SGPR_Count"],
                                                      "scratch": r["Scratch_Size"], "lds": r.get("LDS_Block_Size", ""), "grid": r["Grid_Size"], "wg": r["Workgroup_Size"],
                                                      "dur_us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "counters": {}})
        c = d["counters"]
        c[r["Counter_Name"]] = c.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    by_kernel = {}
    for k in sorted(by_disp):
        by_kernel.setdefault(by_disp[k]["kernel"], []).append(by_disp[k])
    out = []
    for name, ds in by_kernel.items():
        e = dict(ds[-1])
        tail = ds[-20:]
        e["dispatches"] = len(ds)
        e["mean_last20"] = {"n": len(tail), "dur_us": sum(d["dur_us"] for d in tail) / len(tail),
                            "counters": {k: sum(d["counters"].get(k, 0.0) for d in tail) / len(tail) for k in e["counters"]}}
        e["total_dur_us"] = sum(d["dur_us"] for d in ds)
        out.append(e)
    out.sort(key=lambda e: -e["total_dur_us"])
    return out


for sub in ("pmc_sq", "pmc_fetch", "pmc_write", "pmc_mix"):
    f = glob.glob(os.path.join(out_dir, sub, "*counter_collection.csv"))
    if not f:
        continue
    rows = load(f[0])
    if rows:
        summary[sub] = per_kernel(rows)

stats = glob.glob(os.path.join(out_dir, "trace", "*kernel_stats.csv"))
if stats:
    summary["kernel_stats"] = [r for r in csv.DictReader(open(stats[0]))]

# derived numbers for the dominant kernel, from the mean over its last 20 dispatches
try:
    d = summary["pmc_sq"][0]
    c = d["mean_last20"]["counters"]
    xcd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0           # summed over the 8 XCDs
    simd_cycles = xcd_cycles * 1024                    # 256 CUs x 4 SIMDs
    d["derived"] = {
        "clock_ghz": xcd_cycles / (d["mean_last20"]["dur_us"] * 1e3),
        "valu_insts_per_wave": c["SQ_INSTS_VALU"] / (int(d["grid"]) / 64),
        # (Rounds 2-5 also derived "valu_busy_frac" = SQ_ACTIVE_INST_VALU * 4 / simd_cycles and got 1.1-1.36: on gfx950
        # SQ_ACTIVE_INST_VALU counts the same events as SQ_INSTS_VALU -- equal to the last digit on the single-instruction
        # launches of tools/valu_fma_probe.hip, profiles/r06/valu_fma_probe_pmc.json -- so that number was 4 x instructions per
        # SIMD cycle, not a busy fraction.  The issue fraction below is the one figure these two counters give.)
        "avg_waves_per_simd(WAVE_CYCLES*4/simd_cycles)": c["SQ_WAVE_CYCLES"] * 4 / simd_cycles,
        "simd_cycles_per_valu_inst": simd_cycles / c["SQ_INSTS_VALU"],
        # MI355X_MICROARCH.md: a wave64 VALU instruction occupies a SIMD for 2 cycles -> issue peak = simd_cycles / 2
        "valu_issue_frac_of_2cycle_peak": c["SQ_INSTS_VALU"] * 2 / simd_cycles,
    }
    summary["headline_kernel"] = d["kernel"]
except Exception as e:  # noqa
    summary["derived_error"] = str(e)
try:
    name = summary["pmc_sq"][0]["kernel"] if "pmc_sq" in summary else summary["pmc_fetch"][0]["kernel"]
    fe = [e for e in summary["pmc_fetch"] if e["kernel"] == name][0]
    we = [e for e in summary["pmc_write"] if e["kernel"] == name][0]
    fetch = fe["mean_last20"]["counters"]["FETCH_SIZE"]
    write = we["mean_last20"]["counters"]["WRITE_SIZE"]
    # MI355X_MICROARCH.md: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the
    # bytes of a wide coalesced stream -> doubled
    summary["hbm_traffic_bytes_per_launch"] = {"kernel": name, "fetch_kib_raw": fetch, "write_kib_raw": write,
                                               "bytes_corrected": (2 * fetch + write) * 1024, "dispatches_averaged": fe["mean_last20"]["n"]}
except Exception as e:  # noqa
    summary["traffic_error"] = str(e)
# instruction-class mix of the dominant kernel and the issue rate that mix allows (tools/valu_mix.py)
try:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import valu_mix
    name = summary["pmc_sq"][0]["kernel"]
    me = [e for e in summary["pmc_mix"] if e["kernel"] == name][0]
    # (the counter classes only: the "issue floor" of rounds 4-5 rested on round 2's pairing model, retired by
    # profiles/r06/valu_peak_reconciled.md)
    m = valu_mix.floor(me["mean_last20"]["counters"], valu_mix.static_split().get(name.replace("void ", "")))
    summary["valu_classes"] = {"fraction_of_valu_instructions": m["fraction_of_valu_instructions"], "kernel": name}
except Exception as e:  # noqa
    summary["valu_classes_error"] = str(e)
# which kernel objects these counters belong to: bench.py quotes them only while the loaded library still holds the same ones
try:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from convectionkernels_amd import api
    summary["kernel_object_sha256"] = api.library_fatbin_sha256()
    summary["source_sha256"] = api.library_source_sha256()
except Exception as e:  # noqa
    summary["kernel_object_error"] = str(e)
json.dump(summary, open(os.path.join(out_dir, "summary.json"), "w"), indent=1)
print(json.dumps(summary.get("pmc_sq", [{}])[0].get("derived", {}), indent=1))
print(json.dumps(summary.get("hbm_traffic_bytes_per_launch", {})))
