#!/usr/bin/env python3
"""Developer tool (GPU box): how many single-plane chain rounds of the BC7 search repeat endpoints that another seed point
of the same (shape, p-bits) already has in this round, or that were evaluated in an earlier round (profile build):
   CVTTMI_LIB=convectionkernels_amd/lib/variants/libcvtt_mi355x_prof.so python tools/bc7_dup_profile.py [blocks]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from convectionkernels_amd import api, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 15
ctx = api.Context(0)
lib = api.load_library()
buf = (ctypes.c_ulonglong * 8)()
for name, b in synth.content_families(N).items():
    t = torch.from_numpy(b).cuda()
    lib.cvttmi_bc7_dup_read(buf)
    ctx.encode_bc7(t); torch.cuda.synchronize()
    lib.cvttmi_bc7_dup_read(buf)
    tot = max(1, buf[0])
    print("%-20s chain rounds per block %8.0f | same as a lower seed point now %.3f | seen in an earlier round %.3f | duplicates by round %s" %
          (name, buf[0] / N, buf[1] / tot, buf[2] / tot, [round(buf[3 + i] / tot, 3) for i in range(4)]), flush=True)
