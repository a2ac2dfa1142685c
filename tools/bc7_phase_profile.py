#!/usr/bin/env python3
"""Developer tool (GPU box): wave-cycle share of the BC7 kernel's phases, from a library built
with  make -C convectionkernels_amd/csrc VARIANT=prof EXTRA=-DCVTT_BC7_PROFILE .
   CVTTMI_LIB=convectionkernels_amd/lib/variants/libcvtt_mi355x_prof.so python tools/bc7_phase_profile.py [size] [opaque]"""
import ctypes, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from convectionkernels_amd import api, synth

size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
opaque = len(sys.argv) > 2 and sys.argv[2] == "opaque"
names = ["dual seeds", "dual-plane search", "partition bounds", "seed PCA", "single-plane search", "pack", "load+block bounds", "projection"]
ctx = api.Context(0)
lib = api.load_library()
t = torch.from_numpy(synth.tile_blocks(synth.image_rgba8(2, size, size, opaque=opaque))).cuda()
buf = (ctypes.c_ulonglong * 48)()
for exhaustive in (False, True):
    ctx.set_exhaustive(exhaustive)
    lib.cvttmi_bc7_prof_read(buf)
    ctx.encode_bc7(t); torch.cuda.synchronize()
    lib.cvttmi_bc7_prof_read(buf)
    tot = float(sum(buf[:8]))
    print("exhaustive" if exhaustive else "pruned", {names[i]: round(buf[i] / tot, 4) for i in range(8)}, "cycles/wave", tot / (t.shape[0] / 16))
    print("   wave-cycle histogram (log2 buckets from 2^12):", list(buf[16:32]))
    print("   dual configs: quad-evaluations executed", buf[32], "needed by the block itself", buf[33],
          "| single-plane shapes: executed", buf[34], "needed", buf[35])
    print("   dual configs an exact alpha-plane error would have pruned:", buf[38], "| colour planes an alpha-first, best-key-first order would evaluate:", buf[39])
    print("   waves by number of single-plane chain passes (log2 buckets 1, 2-3, 4-7, ... >=128):", list(buf[40:48]))
    print("   candidate partitions alive at stage start (all stages):", buf[36], "block-stages:", buf[37])
