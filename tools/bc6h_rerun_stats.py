#!/usr/bin/env python3
"""Developer tool (GPU box; library built with -DCVTT_BC6H_PROFILE): how often the BC6H kernel's parallel seed-point chains meet a
LATE duplicate (a round that all eight blocks of a group repeat, known only after the last pass) and run a subset's passes again.
    CVTTMI_LIB=.../libcvtt_mi355x_prof.so python tools/bc6h_rerun_stats.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from convectionkernels_amd import api, synth
ctx = api.Context(0)
lib = api.load_library()
for name, b in [("noise (config 3, 1024^2)", synth.tile_blocks(synth.image_f16bits(3, 1024, 1024)))] + list(synth.hdr_content_families(1 << 16).items()):
    buf = (ctypes.c_ulonglong * 32)()
    lib.cvttmi_bc6h_prof_read(buf)
    ctx.encode_bc6h(torch.from_numpy(b).cuda(), signed=False); torch.cuda.synchronize()
    lib.cvttmi_bc6h_prof_read(buf)
    v = list(buf)
    print("%-26s %7d blocks: partition searches (per wave) %d, subset passes run again %d (%.4f per search), lazy %d (with replay %d, %d round slots), eager %d"
          % (name, len(b), v[0], v[1], v[1] / max(1, v[0]), v[2], v[3], v[4], v[5]))
    print("    several-mode commits worked out without the pair loop %d, with it %d" % (v[6], v[7]))
    print("    forced round (tweak, pass) histogram:", {"(%d,%d)" % (m // 3, m % 3): v[8 + m] for m in range(12) if v[8 + m]})
    print("    new drops found per run again:", {k: v[20 + k] for k in range(8) if v[20 + k]})
