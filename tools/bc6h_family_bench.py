#!/usr/bin/env python3
"""Developer tool (GPU box): EncodeBC6HU on the three HDR content families of synth.hdr_content_families, 2^19 blocks each
(CVTTMI_LIB selects a library variant; the digest shows that variants agree)."""
import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from convectionkernels_amd import api, synth
ctx = api.Context(0)
n = 1 << 19
for name, b in synth.hdr_content_families(n).items():
    t = torch.from_numpy(b).cuda()
    o = ctx.encode_bc6h(t, signed=False); torch.cuda.synchronize()
    ms = []
    for _ in range(2):
        a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        a.record(); ctx.encode_bc6h(t, signed=False, out=o); e.record(); torch.cuda.synchronize(); ms.append(a.elapsed_time(e))
    import hashlib
    print(os.path.basename(os.environ.get("CVTTMI_LIB", "current")), name, round(n / min(ms) / 1e3, 2), "Mblocks/s", hashlib.sha256(o.cpu().numpy().tobytes()).hexdigest()[:10])
