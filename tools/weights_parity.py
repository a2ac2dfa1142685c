#!/usr/bin/env python3
"""Developer tool (GPU box): BC7 / BC1 against the real reference (oracle/_ref) with degenerate channel weights (zeros give
inf / NaN inside the reference, EndpointSelector.h:51-70); every block must still match.
   python tools/weights_parity.py"""
import sys, numpy as np, torch, os, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import content
from convectionkernels_amd import api
from oracle import pyref
ctx = api.Context(0); ref = pyref.RefLib(fast=True)
ctx.set_rcp_table(ref.probe_rcp())
blocks = np.concatenate([content.mixed_ldr_blocks(5, 64), content.config_blocks(2, 256, 256)])
plan = ref.default_plan()
for w in ((1,1,1,0), (0,1,0,1), (1,1,1,1e-3), (100,1,0.01,5), (0,0,0,0), (1,1,1,1)):
    o = api.Options(); o.redWeight, o.greenWeight, o.blueWeight, o.alphaWeight = w
    ob = np.frombuffer(o.tobytes(), np.uint8).copy()
    g = ctx.encode_bc7(torch.from_numpy(blocks).cuda(), o).cpu().numpy()
    r = ref.encode_bc7(blocks, ob, plan)
    print("BC7 weights", w, "mismatches", int((g != r).any(axis=1).sum()), "of", blocks.shape[0], flush=True)
    g = ctx.encode_bc1(torch.from_numpy(blocks).cuda(), o).cpu().numpy()
    r = ref.encode_bc1(blocks, ob)
    print("BC1 weights", w, "mismatches", int((g != r).any(axis=1).sum()), flush=True)

# ETC2 RGB / BC6H with the same weights (canonical reference build)
canon = pyref.RefLib()
hdr = content.mixed_hdr_blocks(9, 16)
etcb = blocks[:1024]
for w in ((1, 1, 1, 0), (0, 1, 0, 1), (100, 1, 0.01, 5), (0, 0, 0, 0)):
    o = api.Options(); o.redWeight, o.greenWeight, o.blueWeight, o.alphaWeight = w
    ob = np.frombuffer(o.tobytes(), np.uint8).copy()
    g = ctx.encode_etc2(torch.from_numpy(etcb).cuda(), o).cpu().numpy()
    r = canon.encode_etc2(etcb, ob, 0)
    print("ETC2 weights", w, "mismatches", int((g != r).any(axis=1).sum()), "of", etcb.shape[0], flush=True)
    g = ctx.encode_bc6h(torch.from_numpy(hdr).cuda(), o, signed=False).cpu().numpy()
    r = canon.encode_bc6h(hdr, ob, False)
    print("BC6HU weights", w, "mismatches", int((g != r).any(axis=1).sum()), "of", hdr.shape[0], flush=True)
