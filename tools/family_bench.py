#!/usr/bin/env python3
"""Developer tool (GPU box): BC7 kernel throughput on the content families of tools/stress_parity.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from convectionkernels_amd import api
import importlib.util
spec = importlib.util.spec_from_file_location("sp", os.path.join(os.path.dirname(os.path.abspath(__file__)), "stress_parity.py"))
sp = importlib.util.module_from_spec(spec); sys.argv = [sys.argv[0], "1048576"]; spec.loader.exec_module(sp)
ctx = api.Context(0)
for name, b in sp.families(1 << 20).items():
    t = torch.from_numpy(b).cuda()
    o = ctx.encode_bc7(t); torch.cuda.synchronize()
    ms = []
    for _ in range(2):
        a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        a.record(); ctx.encode_bc7(t, out=o); e.record(); torch.cuda.synchronize(); ms.append(a.elapsed_time(e))
    modes = np.bincount([(int(x) & -int(x)).bit_length() - 1 for x in o[:65536, 0].cpu().numpy()], minlength=8)
    print("%-22s %7.2f ms  %7.1f Mblocks/s   modes %s" % (name, min(ms), b.shape[0] / min(ms) / 1e3, modes.tolist()), flush=True)
