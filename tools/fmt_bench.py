#!/usr/bin/env python3
"""Kernel timing of one format on synthetic BASELINE-style input (developer tool, GPU box).
   python tools/fmt_bench.py bc7|bc7o|bc7b|bc7u|bc1|bc1x|bc2|bc3|bc4|bc5|bc6hu|bc6hs|etc1|etc2|etc2pt|etc2rgba|eac [size] [reps]   (bc1x / bc7b = with Flags::Better, bc7u = Flags::Ultra)
   bc7c5 / bc7c5u = BASELINE config 5a / 5b (seed-5 image, default / Flags::Ultra)
   bc7photo|bc7grad|bc7two = EncodeBC7 on (size/4)^2 blocks of the photo-like / smooth opaque gradient / two-colour family of synth.content_families"""
import sys, os, json, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from convectionkernels_amd import api, synth

fmt = sys.argv[1]
size = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ctx = api.Context(0)
FAMILY = {"bc7photo": "photo-like", "bc7grad": "gradient opaque", "bc7two": "two colours"}
if fmt in FAMILY:
    b = synth.content_families((size // 4) ** 2)[FAMILY[fmt]]
elif fmt in ("bc7c5", "bc7c5u"):  # BASELINE config 5a / 5b: the seed-5 image (16384^2 at full size), default options / Flags::Ultra
    b = synth.tile_blocks(synth.image_rgba8(5, size, size))
elif fmt in ("bc7", "bc7o", "bc7b", "bc7u", "bc1", "bc1x", "bc2", "bc3", "bc4", "bc5", "etc1", "etc2", "etc2pt", "etc2rgba", "eac"):
    b = synth.tile_blocks(synth.image_rgba8(2, size, size, opaque=(fmt == "bc7o")))
else:
    b = synth.tile_blocks(synth.image_f16bits(3, size, size))
t = torch.from_numpy(b).cuda()
enc = {"bc7": ctx.encode_bc7, "bc7c5": ctx.encode_bc7, "bc7c5u": lambda x, out=None: ctx.encode_bc7(x, api.Options(flags=api.Flags.Ultra), out=out), "bc7o": ctx.encode_bc7, "bc7photo": ctx.encode_bc7, "bc7grad": ctx.encode_bc7, "bc7two": ctx.encode_bc7, "bc7b": lambda x, out=None: ctx.encode_bc7(x, api.Options(flags=api.Flags.Better), out=out),
       "bc7u": lambda x, out=None: ctx.encode_bc7(x, api.Options(flags=api.Flags.Ultra), out=out), "bc1": ctx.encode_bc1,
       "bc1x": lambda x, out=None: ctx.encode_bc1(x, api.Options(flags=api.Flags.Better), out=out),
       "bc6hu": lambda x, out=None: ctx.encode_bc6h(x, signed=False, out=out),
       "bc6hs": lambda x, out=None: ctx.encode_bc6h(x, signed=True, out=out),
       "bc2": ctx.encode_bc2, "bc3": ctx.encode_bc3, "bc4": ctx.encode_bc4, "bc5": ctx.encode_bc5,
       "etc1": ctx.encode_etc1, "etc2pt": ctx.encode_etc2_punchthrough_alpha, "etc2": ctx.encode_etc2, "etc2rgba": ctx.encode_etc2_rgba, "eac": ctx.encode_etc2_alpha}[fmt]
o = enc(t); torch.cuda.synchronize()
ms = []
for _ in range(reps):
    a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    a.record(); enc(t, out=o); e.record(); torch.cuda.synchronize(); ms.append(a.elapsed_time(e))
print(json.dumps({"fmt": fmt, "size": size, "blocks": int(b.shape[0]), "ms": min(ms), "mblocks_s": b.shape[0] / min(ms) / 1e3,
                  "sha": hashlib.sha256(o.cpu().numpy().tobytes()).hexdigest()[:12], "lib": os.path.basename(os.environ.get("CVTTMI_LIB", ""))}))
