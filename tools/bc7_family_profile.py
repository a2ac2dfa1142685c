#!/usr/bin/env python3
"""Developer tool (GPU box): phase shares of the BC7 kernel per content family of tools/stress_parity.py, from a library
built with  make -C convectionkernels_amd/csrc VARIANT=prof EXTRA=-DCVTT_BC7_PROFILE :
   CVTTMI_LIB=convectionkernels_amd/lib/variants/libcvtt_mi355x_prof.so python tools/bc7_family_profile.py [blocks]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from convectionkernels_amd import api
import importlib.util
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 17
spec = importlib.util.spec_from_file_location("sp", os.path.join(os.path.dirname(os.path.abspath(__file__)), "stress_parity.py"))
sp = importlib.util.module_from_spec(spec); sys.argv = [sys.argv[0], str(N)]; spec.loader.exec_module(sp)
names = ["dual seeds", "dual-plane search", "partition bounds", "seed PCA", "single-plane search", "pack", "load+block bounds", "projection"]
ctx = api.Context(0)
lib = api.load_library()
buf = (ctypes.c_ulonglong * 48)()
for name, b in sp.families(N).items():
    t = torch.from_numpy(b).cuda()
    lib.cvttmi_bc7_prof_read(buf)
    ctx.encode_bc7(t); torch.cuda.synchronize()
    lib.cvttmi_bc7_prof_read(buf)
    tot = float(sum(buf[:8]))
    print("%-20s" % name, {names[i]: round(buf[i] / tot, 3) for i in range(8)}, "cycles/wave %.0f" % (tot / (N / 16)))
    print("      dual quad-evaluations per block %.2f | partitions alive at stage start per block-stage %.2f | chain-pass buckets %s" %
          (buf[32] / N, buf[36] / max(1, buf[37]), list(buf[40:48])), flush=True)
