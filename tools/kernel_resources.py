#!/usr/bin/env python3
"""Registers, scratch and LDS of every kernel in the shipped library (developer tool, no GPU needed):
   python tools/kernel_resources.py [lib.so] [--json]
Reads the gfx950 code objects out of the library's .hip_fatbin and their AMDGPU metadata notes (llvm-readelf)."""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
CXXFILT = "c++filt"


def code_objects(path):
    data = open(path, "rb").read()
    for m in re.finditer(b"\x7fELF", data):
        o = m.start()
        if o == 0 or struct.unpack_from("<H", data, o + 18)[0] != 224:  # EM_AMDGPU
            continue
        shoff, = struct.unpack_from("<Q", data, o + 0x28)
        shentsize, shnum, _ = struct.unpack_from("<HHH", data, o + 0x3A)
        yield data[o:o + shoff + shentsize * shnum]


def kernels(path=None):
    path = path or os.path.join(ROOT, "convectionkernels_amd", "lib", "libcvtt_mi355x.so")
    res = {}
    for obj in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(obj)
            f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for blk in txt.split("- .agpr_count:")[1:]:
            def field(name, b=blk):
                m = re.search(r"\.%s:\s+(\S+)" % name, b)
                return m.group(1) if m else None
            name = field("name")
            if not name:
                continue
            res[name] = {"vgpr": int(field("vgpr_count")), "sgpr": int(field("sgpr_count")),
                         "scratch_bytes_per_lane": int(field("private_segment_fixed_size")),
                         "lds_bytes": int(field("group_segment_fixed_size")),
                         "vgpr_spills": int(field("vgpr_spill_count") or 0), "sgpr_spills": int(field("sgpr_spill_count") or 0)}
    names = list(res)
    if names:
        try:
            dem = subprocess.run([CXXFILT] + names, capture_output=True, text=True).stdout.splitlines()
        except OSError:
            dem = []
        if len(dem) == len(names):
            res = {re.sub(r"\(.*", "", d).replace("void ", ""): res[n] for d, n in zip(dem, names)}
    return res


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    r = kernels(args[0] if args else None)
    if "--json" in sys.argv:
        print(json.dumps(r, indent=1))
    else:
        for k in sorted(r):
            v = r[k]
            print("%-64s vgpr %3d  scratch %3d B  lds %5d B  (vgpr spills %d, sgpr spills %d)" % (k[:64], v["vgpr"], v["scratch_bytes_per_lane"], v["lds_bytes"], v["vgpr_spills"], v["sgpr_spills"]))
