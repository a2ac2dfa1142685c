#!/usr/bin/env python3
"""Developer tool (GPU box, profile build -- see bc7_stage_profile.py): how many chain rounds of the BC7 single-plane search
start from end points another seed point of the same (unit, p-bit combination) already has in this round / had in an earlier one."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from convectionkernels_amd import api, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 14
want = sys.argv[2:]
ctx = api.Context(0)
lib = api.load_library()
buf = (ctypes.c_ulonglong * 8)()
for name, b in synth.content_families(N).items():
    if want and name not in want:
        continue
    t = torch.from_numpy(b).cuda()
    lib.cvttmi_bc7_dup_read(buf)
    ctx.encode_bc7(t); torch.cuda.synchronize()
    lib.cvttmi_bc7_dup_read(buf)
    r = [int(v) for v in buf]
    print("%-20s chain rounds %10d | same as a lower seed now %.3f | seen earlier %.3f | dup by round %s" % (name, r[0], r[1] / max(1, r[0]), r[2] / max(1, r[0]), r[3:7]))
