import sys, os, hashlib
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from convectionkernels_amd import api, synth
ctx = api.Context(0)
b = synth.tile_blocks(synth.image_rgba8(2, 2048, 2048))
t = torch.from_numpy(b).cuda()
for name, fn in (("etc2rgba fake", ctx.encode_etc2_rgba), ("etc2 fake", ctx.encode_etc2), ("etc1 fake", ctx.encode_etc1)):
    opt = api.Options(flags=api.Flags.Default | api.Flags.ETC_UseFakeBT709)
    o = fn(t, opt); torch.cuda.synchronize()
    ms = []
    for _ in range(3):
        a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        a.record(); fn(t, opt, out=o); e.record(); torch.cuda.synchronize(); ms.append(a.elapsed_time(e))
    print(name, "%.2f Mblocks/s" % (b.shape[0] / min(ms) / 1e3), hashlib.sha256(o.cpu().numpy().tobytes()).hexdigest()[:10], os.path.basename(os.environ.get("CVTTMI_LIB", "shipped")))
