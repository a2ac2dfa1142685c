#!/usr/bin/env python3
"""Condense the per-format rocprofv3 CSVs of tools/profile_formats.sh into fmt_summary.json."""
import csv, glob, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import valu_mix  # noqa: E402
try:
    STATIC = valu_mix.static_split()
except Exception:  # noqa
    STATIC = {}
out_dir = sys.argv[1]
summary = {"fmt_bench": [json.loads(l) for l in open(os.path.join(out_dir, "fmt_bench.jsonl")) if l.strip()]}
for d in sorted(glob.glob(os.path.join(out_dir, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    fmt = os.path.basename(d)[4:]
    f = glob.glob(os.path.join(d, "*counter_collection.csv"))
    if not f:
        continue
    disp = {}
    for r in csv.DictReader(open(f[0])):
        if "cvttmi" not in r["Kernel_Name"]:
            continue
        e = disp.setdefault(r["Dispatch_Id"], {"kernel": r["Kernel_Name"].split("(")[0], "vgpr": r["VGPR_Count"], "sgpr": r["SGPR_Count"],
                                               "scratch": r["Scratch_Size"], "lds": r.get("LDS_Block_Size", ""), "grid": int(r["Grid_Size"]),
                                               "wg": int(r["Workgroup_Size"]),
                                               "dur_us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "counters": {}})
        e["counters"][r["Counter_Name"]] = e["counters"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    # keep the last dispatch of each kernel (the timed one)
    last = {}
    for e in disp.values():
        last[e["kernel"]] = e
    for e in last.values():
        c = e["counters"]
        try:
            xcd = c["GRBM_GUI_ACTIVE"] / 8.0
            simd = xcd * 1024
            e["derived"] = {"valu_insts_per_wave": c["SQ_INSTS_VALU"] / (e["grid"] / 64),
                            # (SQ_ACTIVE_INST_VALU == SQ_INSTS_VALU on gfx950 -- profiles/r06/valu_fma_probe_pmc.json -- so the
                            # "valu_busy_frac" / "cycles_per_valu_inst" of rounds 2-5 said nothing: dropped)
                            "avg_waves_per_simd": c["SQ_WAVE_CYCLES"] * 4 / simd,
                            "simd_cycles_per_valu_inst": simd / c["SQ_INSTS_VALU"],
                            "valu_issue_frac_of_2cycle_peak": c["SQ_INSTS_VALU"] * 2 / simd}
        except Exception as ex:  # noqa
            e["derived_error"] = str(ex)
    # second pass: LDS / memory instruction counters of the same kernel
    f2 = glob.glob(os.path.join(out_dir, "pmc2_" + fmt, "*counter_collection.csv"))
    if f2:
        d2 = {}
        for r in csv.DictReader(open(f2[0])):
            if "cvttmi" not in r["Kernel_Name"]:
                continue
            e2 = d2.setdefault((r["Kernel_Name"].split("(")[0], r["Dispatch_Id"]), {})
            e2[r["Counter_Name"]] = e2.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        for (kname, _), c2 in d2.items():
            if kname in last:
                waves = last[kname]["grid"] / 64
                last[kname]["mem"] = {k: round(v / waves, 1) for k, v in c2.items()}
                last[kname]["mem_note"] = "per wave"
    # third / fourth pass: HBM traffic (FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of a
    # wide coalesced stream -> doubled, as tools/summarize_pmc.py does for the headline)
    tr = {}
    for sub, cname in (("pmcf_", "FETCH_SIZE"), ("pmcw_", "WRITE_SIZE")):
        f3 = glob.glob(os.path.join(out_dir, sub + fmt, "*counter_collection.csv"))
        if not f3:
            continue
        d3 = {}
        for r in csv.DictReader(open(f3[0])):
            if "cvttmi" in r["Kernel_Name"] and r["Counter_Name"] == cname:
                k3 = (r["Kernel_Name"].split("(")[0], r["Dispatch_Id"])
                d3[k3] = d3.get(k3, 0.0) + float(r["Counter_Value"])
        for (kname, _), v in d3.items():  # dispatches in order: the last one of each kernel stays
            tr.setdefault(kname, {})[cname] = v
    for kname, c3 in tr.items():
        if kname in last and "FETCH_SIZE" in c3 and "WRITE_SIZE" in c3:
            last[kname]["hbm"] = {"fetch_kib_raw": c3["FETCH_SIZE"], "write_kib_raw": c3["WRITE_SIZE"],
                                  "bytes_corrected": (2 * c3["FETCH_SIZE"] + c3["WRITE_SIZE"]) * 1024}
    # fifth pass: instruction-class counters -> the mix and what it can issue at best (tools/valu_mix.py)
    f5 = glob.glob(os.path.join(out_dir, "pmcm_" + fmt, "*counter_collection.csv"))
    if f5:
        d5 = {}
        for r in csv.DictReader(open(f5[0])):
            if "cvttmi" in r["Kernel_Name"]:
                e5 = d5.setdefault((r["Kernel_Name"].split("(")[0], r["Dispatch_Id"]), {})
                e5[r["Counter_Name"]] = e5.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        for (kname, _), c5 in d5.items():  # the last dispatch of each kernel stays
            if kname in last and c5.get("SQ_INSTS_VALU"):
                try:
                    # (the counter classes only: the "issue floor" that rounds 4-5 derived from them rested on round 2's
                    # pairing model, which profiles/r06/valu_peak_reconciled.md retires)
                    m = valu_mix.floor(c5, STATIC.get(kname.replace("void ", "")))
                    last[kname]["valu_classes"] = {"fraction_of_valu_instructions": m["fraction_of_valu_instructions"]} if m else None
                except Exception as ex:  # noqa
                    last[kname]["valu_classes_error"] = str(ex)
    summary[fmt] = list(last.values())
    st = glob.glob(os.path.join(out_dir, "trace_" + fmt, "*kernel_stats.csv"))
    if st:
        summary[fmt + "_kernel_stats"] = [r for r in csv.DictReader(open(st[0])) if "cvttmi" in r["Name"]]
try:  # which library these counters belong to (bench.py quotes them only for that one)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from convectionkernels_amd import api
    summary["source_sha256"] = api.library_source_sha256()
except Exception as ex:  # noqa
    summary["source_sha256_error"] = str(ex)
json.dump(summary, open(os.path.join(out_dir, "fmt_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1)[:6000])
