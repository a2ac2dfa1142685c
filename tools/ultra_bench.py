#!/usr/bin/env python3
"""Developer tool (GPU box): BASELINE config 5b -- EncodeBC7, Flags::Ultra, one 16384^2 image (seed 5); prints Mblocks/s and
the SHA-256 check against tests/golden/config_hashes.json."""
import sys, os, json, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from convectionkernels_amd import api, synth
size = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
h = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "config_hashes.json")))
ctx = api.Context(0)
ctx.set_rcp_table(np.array(h["rcp_hex"], np.uint32).view(np.float32))
t = torch.from_numpy(synth.tile_blocks(synth.image_rgba8(5, size, size))).cuda()
opt, plan = api.Options(flags=api.Flags.Ultra), api.BC7EncodingPlan()
out = ctx.encode_bc7(t, opt, plan); torch.cuda.synchronize()
ms = []
for _ in range(3):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); ctx.encode_bc7(t, opt, plan, out=out); b.record(); torch.cuda.synchronize(); ms.append(a.elapsed_time(b))
d = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()
print(json.dumps({"size": size, "mblocks_s": t.shape[0] / min(ms) / 1e3, "ms": min(ms),
                  "sha256_matches_reference": (d == h.get("config5b_bc7_16384_seed5_ultra")) if size == 16384 else None,
                  "lib": os.path.basename(os.environ.get("CVTTMI_LIB", ""))}))
