#!/usr/bin/env python3
"""Developer tool (GPU box): wave-cycle share of the ETC colour kernel's stages, from a library built with
   make -C convectionkernels_amd/csrc VARIANT=profe EXTRA=-DCVTT_ETC_PROFILE
   CVTTMI_LIB=convectionkernels_amd/lib/variants/libcvtt_mi355x_profe.so python tools/etc_phase_profile.py [size]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from convectionkernels_amd import api, synth
size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
names = ["load + planar", "sector split + T mode 1", "T mode 2", "H mode", "cluster fit: pair search + rest (+ punch-through stages)", "cluster fit: TestHalfBlock", "(counter)", "cluster fit: base colours"]
ctx = api.Context(0); lib = api.load_library()
t = torch.from_numpy(synth.tile_blocks(synth.image_rgba8(4, size, size))).cuda()
buf = (ctypes.c_ulonglong * 16)()
for label, fn in (("etc2", ctx.encode_etc2), ("etc1", ctx.encode_etc1), ("punch-through", ctx.encode_etc2_punchthrough_alpha)):
    lib.cvttmi_etc_prof_read(buf)
    fn(t); torch.cuda.synchronize()
    lib.cvttmi_etc_prof_read(buf)
    tot = float(sum(buf[i] for i in (0, 1, 2, 3, 4, 5, 7))) or 1.0
    print(label, {names[i]: round(buf[i] / tot, 3) for i in (0, 1, 2, 3, 4, 5, 7)}, "cycles/block", tot / t.shape[0])
    n = float(t.shape[0])
    print("   pair walk per block: walks %.2f, steps by scanning %.2f, dealt to the lanes %.2f attempts (%.2f succeeded; sets %.1f / %.1f entries), steps on lanes %.2f" %
          (buf[12] / n, buf[8] / n, buf[9] / n, buf[10] / n, buf[13] / max(1, buf[9]), buf[14] / max(1, buf[9]), buf[11] / n))
