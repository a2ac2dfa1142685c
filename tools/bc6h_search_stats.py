"""Developer tool: what the reference's BC6H search does on a kind of content, counted in an instrumented build of the C
oracle (-DORC_BC6H_STATS; test infrastructure, never the product).  Per two-subset precision: rounds run, rounds skipped
because a whole 8-block group repeats an earlier round, rounds in which a single block repeats one (a lane cannot skip on its
own in the reference), how often subset 0's own delta fits a mode, and how many (block, partition) searches commit anything.
Usage: python tools/bc6h_search_stats.py [groups]"""
import ctypes
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from convectionkernels_amd import synth  # noqa: E402
from oracle import pyref  # noqa: E402

NAMES = ["rounds", "groupdup", "lanerounds", "lanedup", "lanedup_exact", "ownfit0", "lanepart", "lanepart_canbeat",
         "lanepart_commit", "commits", "pairs_better", "pairs_better_legal", "grouppart", "grouppart_canbeat", "grouppart_commit"]


def main():
    groups = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    so = os.path.join(tempfile.gettempdir(), "libcvtt_oracle_stats.so")
    subprocess.check_call(["gcc", "-O2", "-std=c99", "-msse2", "-mfpmath=sse", "-ffp-contract=off", "-fno-fast-math", "-frounding-math",
                           "-fPIC", "-shared", "-pthread", "-DORC_BC6H_STATS", "-o", so, os.path.join(ROOT, "oracle", "cvtt_oracle.c"), "-lm"])
    lib = ctypes.CDLL(so)
    stats = (ctypes.c_ulonglong * (17 * len(NAMES))).in_dll(lib, "orc_bc6h_stats")

    opt = pyref.make_options()
    for name, b in synth.hdr_content_families(groups * 8).items():
        ctypes.memset(stats, 0, ctypes.sizeof(stats))
        out = np.zeros((groups * 8, 16), np.uint8)
        bb = np.ascontiguousarray(b)
        rcp = pyref.OracleLib().probe_rcp()
        lib.orc_encode_bc6h.restype = ctypes.c_int
        rc = lib.orc_encode_bc6h(out.ctypes.data_as(ctypes.c_void_p), bb.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(groups * 8),
                                 opt.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(0), rcp.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(1))
        assert rc == 0, rc
        a = np.array(stats[:], np.float64).reshape(17, len(NAMES))
        print("== %s (%d groups)" % (name, groups))
        for prec in (11, 10, 9, 8, 7, 6):
            r = dict(zip(NAMES, a[prec]))
            tot = r["rounds"] + r["groupdup"]
            print("  prec %2d: group-rounds %6d  group-dup %5.1f%%  lane-dup %5.1f%% (exact %5.1f%%)  own-fit0 %5.1f%%  "
                  "lane-part canbeat %5.1f%% commit %5.2f%%  group-part canbeat %5.1f%% commit %5.1f%%  commits/block %.2f  pairs better/lane-part %.1f"
                  % (prec, tot, 100 * r["groupdup"] / max(tot, 1), 100 * r["lanedup"] / max(r["lanerounds"], 1),
                     100 * r["lanedup_exact"] / max(r["lanerounds"], 1), 100 * r["ownfit0"] / max(r["lanerounds"] / 2, 1),
                     100 * r["lanepart_canbeat"] / max(r["lanepart"], 1), 100 * r["lanepart_commit"] / max(r["lanepart"], 1),
                     100 * r["grouppart_canbeat"] / max(r["grouppart"], 1), 100 * r["grouppart_commit"] / max(r["grouppart"], 1),
                     r["commits"] / (groups * 8), r["pairs_better"] / max(r["lanepart"], 1)))


if __name__ == "__main__":
    main()
