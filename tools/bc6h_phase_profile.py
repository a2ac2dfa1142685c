#!/usr/bin/env python3
"""Developer tool (GPU box): wave-cycle share of the BC6H kernel's phases, from a library built with
   make -C convectionkernels_amd/csrc VARIANT=prof6 EXTRA=-DCVTT_BC6H_PROFILE"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from convectionkernels_amd import api, synth
size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
names = ["PCA seeds", "endpoints+quantise", "index selection", "duplicate test", "error+refiner sums", "legality/commit", "zero meta", "-"]
ctx = api.Context(0); lib = api.load_library()
t = torch.from_numpy(synth.tile_blocks(synth.image_f16bits(3, size, size))).cuda()
buf = (ctypes.c_ulonglong * 16)()
lib.cvttmi_bc6h_prof_read(buf)
ctx.encode_bc6h(t); torch.cuda.synchronize()
lib.cvttmi_bc6h_prof_read(buf)
tot = float(sum(buf[:8]))
print({names[i]: round(buf[i] / tot, 4) for i in range(7)}, "cycles/wave", tot / (t.shape[0] / 64))
waves = t.shape[0] / 64
print("per wave: lazy partitions %.1f, eager partitions (prec >= 8) %.1f, pair searches %.2f, their rounds %.2f (lanes with a candidate %.2f), replays %.3f" % (buf[13] / waves, buf[11] / waves, buf[8] / waves, buf[9] / waves, buf[12] / waves, buf[10] / waves))
