#!/usr/bin/env python3
"""Developer tool (GPU box): ETC2 / punch-through parity against the oracle on dark content, where the T mode's
"zero slot" (hazard H2) can win and the kernel has to take its rare exact path (group counts).
   python tools/etc_dark_parity.py"""
import sys, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from convectionkernels_amd import api
from oracle import pyref
rng = np.random.default_rng(7)
ctx = api.Context(0); orc = pyref.OracleLib()
fam = []
n = 8192
b = rng.integers(0, 24, (n, 16, 4)).astype(np.uint8); b[..., 3] = 255; fam.append(b)                 # very dark noise
b = rng.integers(0, 8, (n, 16, 4)).astype(np.uint8); m = rng.random((n, 16)) < 0.15
b[m] = rng.integers(100, 256, (int(m.sum()), 4)); b[..., 3] = 255; fam.append(b)                      # dark with bright outliers
b = np.zeros((n, 16, 4), np.uint8); m = rng.random((n, 16)) < 0.3
b[m] = rng.integers(0, 256, (int(m.sum()), 4)); b[..., 3] = rng.integers(0, 2, (n, 16)) * 255; fam.append(b)  # black + random, binary alpha
b = rng.integers(0, 256, (n, 1, 4)).astype(np.uint8).repeat(16, 1); b[::3, ::2, :3] = 0; b[..., 3] = 255; fam.append(b)  # flat with black pixels
blocks = np.concatenate(fam)
bad = 0
for mode, fn in ((0, ctx.encode_etc2), (4, ctx.encode_etc2_punchthrough_alpha)):
    for flags in (api.Flags.Default, api.Flags.Default | api.Flags.Uniform, api.Flags.Default | api.Flags.ETC_UseFakeBT709):
        exp = orc.encode_etc2(blocks, pyref.make_options(flags=flags), mode, threads=64)
        got = fn(torch.from_numpy(blocks).cuda(), api.Options(flags=flags)).cpu().numpy()
        k = int((got != exp).any(axis=1).sum()); bad += k
        # share of T-mode blocks with black line colour
        print("mode", mode, "flags", hex(flags), "blocks", blocks.shape[0], "mismatches", k, flush=True)
print("TOTAL", bad)
try:  # profile builds count how often the exact path ran
    import ctypes
    buf = (ctypes.c_ulonglong * 8)()
    api.load_library().cvttmi_etc_prof_read(buf)
    print("T-mode exact-path executions:", buf[6])
except AttributeError:
    pass
