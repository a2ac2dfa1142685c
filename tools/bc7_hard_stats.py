#!/usr/bin/env python3
"""Developer tool (GPU box): how many BC7 blocks the first launch hands to the second one (bc7_kernel.hip, HARD), by
CVTTMI_BC7_HARD_MIN = the number of live mode-7 partitions a wave at the very end of the grid may keep (one more for
every CVTTMI_BC7_HARD_DIV waves that follow it); the settings are read when a context is created.  Inputs below 2^19
blocks leave the hand-over off unless CVTTMI_BC7_HARD_CAP is set.   python tools/bc7_hard_stats.py [size] [opaque]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from convectionkernels_amd import api, synth

size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
opaque = len(sys.argv) > 2 and sys.argv[2] == "opaque"
lib = api.load_library()
lib.cvttmi_bc7_hard_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
t = torch.from_numpy(synth.tile_blocks(synth.image_rgba8(2, size, size, opaque=opaque))).cuda()
for hard_min in (4, 8, 12, 16, 24, 32, 48, 64):
    os.environ["CVTTMI_BC7_HARD_MIN"] = str(hard_min)
    ctx = api.Context(0)
    ctx.encode_bc7(t)
    n, cap = ctypes.c_uint32(), ctypes.c_uint32()
    lib.cvttmi_bc7_hard_stats(ctx._h, ctypes.byref(n), ctypes.byref(cap))
    print("allowance %2d: %7d of %d blocks handed over (slots %d)" % (hard_min, n.value, t.shape[0], cap.value))
