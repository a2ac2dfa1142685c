import sys, os, json, hashlib
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from convectionkernels_amd import api, synth
ctx = api.Context(0)
fam = synth.content_families(1 << 18)
res = {}
for k in ("noise", "photo-like", "gradient opaque", "two colours"):
    t = torch.from_numpy(fam[k]).cuda()
    out = ctx.encode_etc2_rgba(t); torch.cuda.synchronize()
    ms = []
    for _ in range(3):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); ctx.encode_etc2_rgba(t, out=out); b.record(); torch.cuda.synchronize(); ms.append(a.elapsed_time(b))
    res[k] = (round(min(ms), 3), hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:10])
print(os.path.basename(os.environ.get("CVTTMI_LIB", "default")), res)
