#!/usr/bin/env python3
"""Developer tool: the round history (quantised end points, errors, validity) of one block at the 6-bit two-subset precision, from
the kernel (library built with -DCVTT_BC6H_TRACE=<index in the 16-block wave>, GPU box) or from the oracle (built with
-DORC_BC6H_TRACE=<lane of the first group>, any host), as text lines that can be diffed.
    python tools/bc6h_trace.py gpu|cpu <lib.so> <A|B>      (A / B: the two 16-block waves of 2048^2 config-3 noise that differed)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import content

which, lib, wave = sys.argv[1], sys.argv[2], sys.argv[3]
big = content.config_blocks_hdr(3, 2048, 2048)
if wave in ("A", "B"):
    blocks = np.ascontiguousarray(big[177296:177312] if wave == "A" else big[222064:222080])
else:  # "name:edge:first": 16 blocks of config-3 noise of that edge from block `first`
    _, edge, first = wave.split(":")
    blocks = np.ascontiguousarray(content.config_blocks_hdr(3, int(edge), int(edge))[int(first):int(first) + 16])
if which == "cpu":
    from oracle import pyref
    os.environ["ORC_BC6H_TRACE_ON"] = "1"
    orc = pyref.OracleLib.__new__(pyref.OracleLib)
    orc.lib = ctypes.CDLL(lib)
    rcp = np.array([0] + [1.0 / i for i in range(1, 17)], np.float32)
    sys.stdout.flush()
    orc.encode_bc6h(blocks[:8], pyref.make_options(), False, rcp, threads=1)
else:
    os.environ["CVTTMI_LIB"] = lib
    from convectionkernels_amd import api
    ctx = api.Context(0)
    ctx.set_rcp_table(np.array([0] + [1.0 / i for i in range(1, 17)], np.float32))
    ctx.encode_bc6h(blocks, api.Options(), signed=False)
    buf = (ctypes.c_uint * (32 * 74))()
    assert api.load_library().cvttmi_bc6h_trace_read(buf) == 0
    buf2 = (ctypes.c_ulonglong * 16)()
    api.load_library().cvttmi_bc6h_trace2_read(buf2)
    sys.stderr.write("TRACE2 " + " ".join("%x" % v for v in buf2) + "\n")
    d = np.frombuffer(buf, np.uint32).reshape(32, 74)
    for p in range(32):
        print("TRACE p %d rv %x %x" % (p, d[p, 0], d[p, 1]))
        for s in range(2):
            for m in range(12):
                wa, wb, err = (int(x) for x in d[p, 2 + (s * 12 + m) * 3: 5 + (s * 12 + m) * 3])
                ep = [wa & 0x7ff, (wa >> 11) & 0x7ff, wa >> 22, wb & 0x7ff, (wb >> 11) & 0x7ff, wb >> 22]
                print("TRACE p %d s %d m %d ep %d %d %d %d %d %d err %08x" % (p, s, m, *ep, err))
