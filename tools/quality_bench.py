#!/usr/bin/env python3
"""Developer tool (GPU box): BC7 throughput with the plans of ConfigureBC7EncodingPlanFromQuality (2048^2 random RGBA / opaque)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from convectionkernels_amd import api, synth
ctx = api.Context(0)
for opaque in (False, True):
    b = synth.tile_blocks(synth.image_rgba8(2, 2048, 2048, opaque=opaque))
    t = torch.from_numpy(b).cuda()
    for q in (1, 20, 50, 80, 100, None):
        plan = api.BC7EncodingPlan()
        if q is not None:
            api.ConfigureBC7EncodingPlanFromQuality(plan, q)
        o = ctx.encode_bc7(t, api.Options(), plan); torch.cuda.synchronize()
        ms = []
        for _ in range(3):
            a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            a.record(); ctx.encode_bc7(t, api.Options(), plan, out=o); e.record(); torch.cuda.synchronize(); ms.append(a.elapsed_time(e))
        print("opaque" if opaque else "alpha", "quality", q, "%.2f ms  %.1f Mblocks/s" % (min(ms), b.shape[0] / min(ms) / 1e3), flush=True)
