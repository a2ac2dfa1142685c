#!/usr/bin/env python3
"""A/B timing of library variants (developer tool, GPU box):
   python tools/ab_bench.py [--set bc7|bc7all|fmt] lib1.so lib2.so ...
Each variant runs in a subprocess (CVTTMI_LIB) so the dlopen'ed code objects never mix.  Every workload reports the best
kernel time of three launches and the first 12 hex digits of the output's SHA-256; a variant whose digest differs from
the first variant's is flagged (the A/B is between builds that must produce the same bytes).
  bc7     BC7 4096^2 RGBA noise + opaque noise (AB_SIZE overrides the edge)
  bc7all  + Flags::Better, the photo-like / gradient / two-colour families (2^18 blocks) and punch-through options
  fmt     BC6HU 2048^2, ETC2 RGBA 2048^2, BC1 4096^2"""
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
size = int(os.environ.get("AB_SIZE", "4096"))
which = os.environ.get("AB_SET", "bc7")
ctx = api.Context(0)
h = json.load(open(os.path.join(%r, "tests", "golden", "config_hashes.json")))
ctx.set_rcp_table(np.array(h["rcp_hex"], np.uint32).view(np.float32))
res = {}
def run(name, blocks, enc, reps=3):
    t = torch.from_numpy(blocks).cuda()
    out = enc(t, None); torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); enc(t, out); b.record(); torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    res[name] = {"ms": round(min(ms), 4), "mblocks_s": round(blocks.shape[0] / min(ms) / 1e3, 2),
                 "sha": hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:12]}
    del t, out
o, p = api.Options(), api.BC7EncodingPlan()
bc7 = lambda t, out: ctx.encode_bc7(t, o, p, out=out)
if which in ("bc7", "bc7all"):
    run("alpha", synth.tile_blocks(synth.image_rgba8(2, size, size)), bc7)
    run("opaque", synth.tile_blocks(synth.image_rgba8(2, size, size, opaque=True)), bc7)
if which == "bc7all":
    better = api.Options(flags=api.Flags.Better)
    run("better", synth.tile_blocks(synth.image_rgba8(2, size, size)), lambda t, out: ctx.encode_bc7(t, better, p, out=out))
    run("better_opaque", synth.tile_blocks(synth.image_rgba8(2, size // 2, size // 2, opaque=True)), lambda t, out: ctx.encode_bc7(t, better, p, out=out))
    pt = api.Options(flags=api.Flags.Default | api.Flags.BC7_RespectPunchThrough)
    fam = synth.content_families(1 << 18)
    run("pt", fam["punch-through alpha"], lambda t, out: ctx.encode_bc7(t, pt, p, out=out))
    for k in ("photo-like", "gradient opaque", "gradient rgba", "two colours", "alpha 248..255"):
        run(k, fam[k], bc7, reps=2)
if which == "fmt":
    run("bc6hu", synth.tile_blocks(synth.image_f16bits(3, 2048, 2048)), lambda t, out: ctx.encode_bc6h(t, o, signed=False, out=out), reps=2)
    run("bc6hs", synth.tile_blocks(synth.image_f16bits(3, 1024, 1024)), lambda t, out: ctx.encode_bc6h(t, o, signed=True, out=out), reps=2)
    run("etc2rgba", synth.tile_blocks(synth.image_rgba8(4, 2048, 2048)), lambda t, out: ctx.encode_etc2_rgba(t, o, out=out))
    run("bc1", synth.tile_blocks(synth.image_rgba8(1, 4096, 4096)), lambda t, out: ctx.encode_bc1(t, o, out=out))
print(json.dumps(res))
''' % (ROOT, ROOT)


def main():
    args = sys.argv[1:]
    which = "bc7"
    if args and args[0] == "--set":
        which = args[1]
        args = args[2:]
    first = None
    for lib in args:
        env = dict(os.environ, CVTTMI_LIB=os.path.abspath(lib), AB_SET=which)
        p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(os.path.basename(lib), "FAILED: " + p.stderr[-600:])
            continue
        r = json.loads(line[-1])
        if first is None:
            first = r
        cells = []
        for k, v in r.items():
            flag = "" if first.get(k, v)["sha"] == v["sha"] else " !!SHA"
            cells.append("%s %.3f ms %.1f M/s%s" % (k, v["ms"], v["mblocks_s"], flag))
        print("%-34s %s" % (os.path.basename(lib), " | ".join(cells)), flush=True)


if __name__ == "__main__":
    main()
