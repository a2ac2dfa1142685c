#!/usr/bin/env python3
"""A/B timing of library variants (developer tool, GPU box):
   python tools/ab_bench.py lib1.so lib2.so ...   -> kernel ms for BC7 4096^2 (alpha + opaque)
Each variant runs in a subprocess (CVTTMI_LIB) so the dlopen'ed code objects never mix."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, os, json, hashlib
sys.path.insert(0, %r)
import numpy as np, torch
from convectionkernels_amd import api, synth
size = int(os.environ.get("AB_SIZE", "2048"))
ctx = api.Context(0)
h = json.load(open(os.path.join(%r, "tests", "golden", "config_hashes.json")))
ctx.set_rcp_table(np.array(h["rcp_hex"], np.uint32).view(np.float32))
res = {}
for name, opaque in (("alpha", False), ("opaque", True)):
    blocks = synth.tile_blocks(synth.image_rgba8(2, size, size, opaque=opaque))
    t = torch.from_numpy(blocks).cuda()
    out = ctx.encode_bc7(t); torch.cuda.synchronize()
    ms = []
    for _ in range(3):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); ctx.encode_bc7(t, out=out); b.record(); torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    res[name] = {"ms": min(ms), "mblocks_s": blocks.shape[0] / min(ms) / 1e3,
                 "sha": hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:12]}
print(json.dumps(res))
''' % (ROOT, ROOT)

for lib in sys.argv[1:]:
    env = dict(os.environ, CVTTMI_LIB=os.path.abspath(lib))
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    print(os.path.basename(lib), line[-1] if line else ("FAILED: " + p.stderr[-400:]))
