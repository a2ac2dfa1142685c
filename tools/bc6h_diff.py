#!/usr/bin/env python3
"""Developer tool (GPU box): EncodeBC6HU on config-3 noise (and the mixed HDR families) against the CPU oracle; prints the
blocks that differ with their mode bits / partition.   python tools/bc6h_diff.py [edge] [signed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import content
from convectionkernels_amd import api
from oracle import pyref

edge = int(sys.argv[1]) if len(sys.argv) > 1 else 256
signed = len(sys.argv) > 2 and sys.argv[2] == "1"
ctx = api.Context(0)
orc = pyref.OracleLib()
rcp = orc.probe_rcp()
ctx.set_rcp_table(rcp)
MODES = {0: "m0(10:5)", 1: "m1(7:6)", 2: "m2(11:5,4,4)", 6: "m3(11:4,5,4)", 10: "m4(11:4,4,5)", 14: "m5(9:5)", 18: "m6(8:6,5,5)", 22: "m7(8:5,6,5)",
         26: "m8(8:5,5,6)", 30: "m9(6:6)", 3: "m10(10:10)", 7: "m11(11:9)", 11: "m12(12:8)", 15: "m13(16:4)"}
def mode(b):
    m = b[0] & 3
    return MODES.get(m if m < 2 else b[0] & 31, "?")
def part(b):
    v = int.from_bytes(bytes(b[8:12]), "little") | (int(b[12]) << 32)
    return ((int.from_bytes(bytes(b), "little")) >> 77) & 31
mixed = content.mixed_hdr_blocks(4242, 40, signed=signed)
noise7 = content.config_blocks_hdr(7, 64, 80)
shuf = np.concatenate([mixed, noise7])
shuf = np.ascontiguousarray(shuf[np.random.Generator(np.random.PCG64(11)).permutation(len(shuf))])
big = content.config_blocks_hdr(3, 2048, 2048)
cases = [("shuffled", shuf), ("waveA", np.ascontiguousarray(big[177296:177312])), ("waveB", np.ascontiguousarray(big[222064:222080]))]
if edge:
    cases.append(("noise", content.config_blocks_hdr(3, edge, edge)))
for name, blocks in cases:
    ob = pyref.make_options()
    exp = orc.encode_bc6h(blocks, ob, signed, rcp, threads=16)
    got = ctx.encode_bc6h(blocks, api.Options.frombytes(ob), signed=signed)
    bad = np.nonzero((got != exp).any(axis=1))[0]
    print(name, len(blocks), "blocks,", bad.size, "differ")
    for i in bad[:12]:
        print("  block %d (group %d lane %d): exp %s p%d %s | got %s p%d %s" % (i, i // 8, i % 8, mode(exp[i]), part(exp[i]), exp[i].tobytes().hex(),
                                                                          mode(got[i]), part(got[i]), got[i].tobytes().hex()))
