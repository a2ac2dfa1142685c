#!/usr/bin/env python3
"""Developer tool (GPU box): where the single-plane search of the BC7 kernel spends its chain batches, per stage (mode) and
content family, from a library built with  make -C convectionkernels_amd/csrc VARIANT=prof EXTRA=-DCVTT_BC7_PROFILE :
   CVTTMI_LIB=convectionkernels_amd/lib/variants/libcvtt_mi355x_prof.so python tools/bc7_stage_profile.py [blocks] [family ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from convectionkernels_amd import api, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 16
want = sys.argv[2:]
ctx = api.Context(0)
lib = api.load_library()
buf = (ctypes.c_ulonglong * 96)()
prof = (ctypes.c_ulonglong * 48)()
phase = ["dual seeds", "dual-plane search", "partition bounds", "seed PCA", "single-plane search", "pack", "load+block bounds", "projection",
         "offers", "item list + sort", "seed pass", "probe chains", "probe filter", "full chains", "commit", "-"]
modes = [6, 7, 1, 3, 0, 2]
for name, b in synth.content_families(N).items():
    if want and name not in want:
        continue
    t = torch.from_numpy(b).cuda()
    out = ctx.encode_bc7(t); torch.cuda.synchronize()
    lib.cvttmi_bc7_stage_read(buf); lib.cvttmi_bc7_prof_read(prof)
    ctx.encode_bc7(t, out=out); torch.cuda.synchronize()
    lib.cvttmi_bc7_stage_read(buf); lib.cvttmi_bc7_prof_read(prof)
    tot = float(sum(prof[:16]))
    o = out.cpu().numpy()
    first = o[:, 0].astype(np.uint32)
    mode = np.array([(int(v) & -int(v)).bit_length() - 1 if v else 8 for v in first])
    hist = np.bincount(mode, minlength=9)[:8]
    print("== %-20s winners by mode %s   phases %s" % (name, hist.tolist(), {phase[i]: round(prof[i] / tot, 3) for i in range(16) if prof[i] / tot >= 0.01}))
    for si, m in enumerate(modes):
        r = [int(buf[si * 8 + k]) for k in range(8)]
        if r[6] == 0:
            continue
        print("   mode %d: alive partitions/block %6.2f | units searched/block %7.2f | chain batches/wave %7.2f | lane use %.2f | offer rounds/wave %6.2f | commits/block %.3f" %
              (m, r[2] / N, r[1] / N, r[0] / (N / 16), r[5] / max(1, r[0] * 64), r[3] / (N / 16), r[4] / N))
        q = [int(buf[48 + si * 8 + k]) for k in range(8)]
        if q[0] or q[3]:
            print("           sharper bound: second-tier rounds/wave %.2f, partitions looked at/block %.2f, removed %.3f of them | probe-filter rounds/wave %.2f, looked at/block %.2f, removed %.3f" %
                  (q[0] / (N / 16), q[1] / N, q[2] / max(1, q[1]), q[3] / (N / 16), q[4] / N, q[5] / max(1, q[4])))
        if False:
            print("           what-if, of %d searched partitions: ruled out by exact(subset 0) + bound(rest) %.3f | exact(subset 1) + bound(rest) %.3f | larger subset first %.3f | "
                  "larger bound first %.3f | either %.3f | by the total bound at commit time %.3f | beat the best %.4f" %
                  (q[0], q[1] / q[0], q[2] / q[0], q[3] / q[0], q[4] / q[0], q[5] / q[0], q[6] / q[0], q[7] / q[0]))
