#!/usr/bin/env python3
"""One-off differential stress test (GPU box): large vectorised content families through the HIP kernels and
through the real reference (oracle/_ref, all host threads), every block compared.  Catches events too rare for
the unit tests (the sqrt rounding case was 1 block in 2 million).
    python tools/stress_parity.py [blocks_per_family] [all|bc7|ldr|hdr|rows] [seed]
"rows" = the formats / flags added after the first four: BC1-BC3 with S3TC_Exhaustive, BC2-BC5, ETC1, ETC2 punch-through, R11."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from convectionkernels_amd import api
from oracle import pyref

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 17
STAGE = sys.argv[2] if len(sys.argv) > 2 else "all"
SEED = int(sys.argv[3]) if len(sys.argv) > 3 else 2026
rng = np.random.Generator(np.random.PCG64(SEED))
yy, xx = np.divmod(np.arange(16), 4)


def families(n):
    from convectionkernels_amd import synth
    return synth.content_families(n, SEED)


def ref_parallel(fn, blocks, per_out, threads):
    n = blocks.shape[0]
    chunk = 512
    out = np.zeros((n, per_out), np.uint8)
    nxt = [0]; lock = threading.Lock()

    def work():
        while True:
            with lock:
                i = nxt[0]; nxt[0] += chunk
            if i >= n:
                return
            out[i:i + chunk] = fn(blocks[i:i + chunk])
    ts = [threading.Thread(target=work) for _ in range(threads)]
    [t.start() for t in ts]; [t.join() for t in ts]
    return out


def main():
    ctx = api.Context(0)
    ref = pyref.RefLib(fast=True)
    canon = pyref.RefLib()
    ctx.set_rcp_table(ref.probe_rcp())
    threads = os.cpu_count() or 8
    plan = ref.default_plan()
    total_bad = 0
    fam = families(N)
    bc7_opts = {"default": api.Options(), "better": api.Options(flags=api.Flags.Better), "uniform": api.Options(flags=api.Flags.Default | api.Flags.Uniform),
                "ultra+pt": api.Options(flags=api.Flags.Ultra | api.Flags.BC7_RespectPunchThrough)}
    for oname, opt in (bc7_opts.items() if STAGE in ("all", "bc7") else ()):
        ob = np.frombuffer(opt.tobytes(), np.uint8).copy()
        for name, b in fam.items():
            if oname != "default" and name in ("noise", "opaque noise"):
                b = b[: N // 4]
            t0 = time.time()
            g = ctx.encode_bc7(torch.from_numpy(b).cuda(), opt).cpu().numpy()
            r = ref_parallel(lambda x: ref.encode_bc7(x, ob, plan), b, 16, threads)
            bad = int((g != r).any(axis=1).sum())
            total_bad += bad
            print("BC7 %-9s %-20s %8d blocks  mismatches %d  (%.1f s)" % (oname, name, b.shape[0], bad, time.time() - t0), flush=True)
    o = pyref.make_options()
    for name, b in (fam.items() if STAGE in ("all", "ldr") else ()):
        b = b[: N // 2]
        g = ctx.encode_bc1(torch.from_numpy(b).cuda()).cpu().numpy()
        r = ref_parallel(lambda x: ref.encode_bc1(x, o), b, 8, threads)
        bad = int((g != r).any(axis=1).sum()); total_bad += bad
        print("BC1 %-30s %8d blocks  mismatches %d" % (name, b.shape[0], bad), flush=True)
        g = ctx.encode_etc2_rgba(torch.from_numpy(b).cuda()).cpu().numpy()
        r = ref_parallel(lambda x: canon.encode_etc2(x, o, 1), b, 16, threads)
        bad = int((g != r).any(axis=1).sum()); total_bad += bad
        print("ETC2 RGBA %-24s %8d blocks  mismatches %d" % (name, b.shape[0], bad), flush=True)
    # HDR: the LDR families re-read as half bit patterns (finite positives), plus wide-range noise
    n6 = N // 8
    h = rng.integers(0, 0x7C00, (n6, 16, 4)).astype(np.uint16); h[:, :, 3] = 0x3C00
    base = rng.integers(0x3000, 0x7000, (n6, 1, 3)); nar = (base + rng.integers(-60, 61, (n6, 16, 3))).astype(np.uint16)
    h2 = np.zeros((n6, 16, 4), np.uint16); h2[:, :, :3] = nar; h2[:, :, 3] = 0x3C00
    # + the three families of bench.py's bc6h_content_families, and everything shuffled block by block: the BC6H search decides
    # wave by wave (64 blocks) what the delta coding rules out, so blocks whose deltas fit must also sit next to blocks whose
    # deltas never do, inside waves and inside the reference's 8-block groups
    hdr = []
    if STAGE in ("all", "hdr"):
        from convectionkernels_amd import synth
        hdr = [("wide noise", h), ("narrow", h2)] + [(k, v.view(np.uint16)) for k, v in synth.hdr_content_families(n6, SEED).items()]
        allb = np.concatenate([v for _, v in hdr])
        hdr.append(("shuffled mix", np.ascontiguousarray(allb[rng.permutation(allb.shape[0])[: 2 * n6]])))
    fastopt = pyref.make_options(flags=pyref.FLAG_BC6H_FAST_INDEXING)
    gpu_fast = api.Options(flags=api.Flags.BC6H_FastIndexing)
    for name, hb in hdr:
        for sg in (False, True):
            hb2 = hb.copy()
            if sg:
                hb2[:, :, :3] |= (rng.integers(0, 2, hb2[:, :, :3].shape) << 15).astype(np.uint16)
            b = hb2.view(np.int16)
            for label, go, ro in (("", None, o), (" fast", gpu_fast, fastopt)):
                if label and name not in ("shuffled mix", "noise"):
                    continue
                g = (ctx.encode_bc6h(torch.from_numpy(b).cuda(), signed=sg) if go is None else ctx.encode_bc6h(torch.from_numpy(b).cuda(), go, signed=sg)).cpu().numpy()
                r = ref_parallel(lambda x: canon.encode_bc6h(x, ro, sg), b, 16, threads)
                bad = int((g != r).any(axis=1).sum()); total_bad += bad
                print("BC6H%s%s %-24s %8d blocks  mismatches %d" % ("S" if sg else "U", label, name, b.shape[0], bad), flush=True)
    if STAGE in ("all", "rows"):
        better = pyref.make_options(flags=pyref.FLAGS_BETTER)
        gpu_better = api.Options(flags=api.Flags.Better)
        for name, b in fam.items():
            b = b[: N // 4]
            t = torch.from_numpy(b).cuda()
            checks = [
                ("BC1 exhaustive", lambda: ctx.encode_bc1(t, gpu_better), lambda x: ref.encode_bc1(x, better), 8, N // 16),
                ("BC3 exhaustive", lambda: ctx.encode_bc3(t, gpu_better), lambda x: ref.encode_s3tc(x, better, 3), 16, N // 16),
                ("BC2", lambda: ctx.encode_bc2(t), lambda x: ref.encode_s3tc(x, o, 2), 16, None),
                ("BC3", lambda: ctx.encode_bc3(t), lambda x: ref.encode_s3tc(x, o, 3), 16, None),
                ("BC4U", lambda: ctx.encode_bc4(t), lambda x: ref.encode_s3tc(x, o, 4), 8, None),
                ("BC5S", lambda: ctx.encode_bc5(t, signed=True), lambda x: ref.encode_s3tc(x, o, 7), 16, None),
                ("ETC1", lambda: ctx.encode_etc1(t), lambda x: canon.encode_etc2(x, o, 3), 8, None),
                ("ETC2 punch-through", lambda: ctx.encode_etc2_punchthrough_alpha(t), lambda x: canon.encode_etc2(x, o, 4), 8, None),
            ]
            for label, gfn, rfn, per, limit in checks:
                g = gfn().cpu().numpy()
                m = b.shape[0] if limit is None else min(limit, b.shape[0])
                m -= m % 8
                r = ref_parallel(rfn, b[:m], per, threads)
                bad = int((g[:m] != r).any(axis=1).sum()); total_bad += bad
                print("%-20s %-22s %8d blocks  mismatches %d" % (label, name, m, bad), flush=True)
        r11 = rng.integers(-1500, 2600, (N // 4, 16)).astype(np.int16)
        for sg in (False, True):
            g = ctx.encode_etc2_alpha11(r11, signed=sg)
            r = ref_parallel(lambda x: ref.encode_eac11(x, o, sg), r11, 8, threads)
            bad = int((np.asarray(g) != r).any(axis=1).sum()); total_bad += bad
            print("R11 %-37s %8d blocks  mismatches %d" % ("signed" if sg else "unsigned", r11.shape[0], bad), flush=True)
    print("TOTAL MISMATCHES", total_bad)


if __name__ == "__main__":
    main()
