"""Reduced differential stress (the driver's `-m gpu` run repeats what tools/stress_parity.py shows in builder logs):
every content family of synth.content_families through the HIP kernels and through the REAL reference (oracle/_ref, all
usable host threads; the C restatement when the reference build did not travel), every block compared.
8 families x 65 536 blocks with the default options and x 32 768 with Flags::Better and Flags::Ultra + BC7_RespectPunchThrough
for BC7; 16 384 blocks each for BC6HU / BC6HS (wide-range noise, narrow-range blocks and the three families of
synth.hdr_content_families) and for ETC2 RGBA.  These are the families on which the branch-and-bound prunes hardest (smooth, two-colour,
photo-like) and least (noise)."""
import os
import time

import numpy as np
import pytest

from oracle import pyref

pytestmark = pytest.mark.gpu

N_BC7 = {"default": 65536, "better": 32768, "ultra_pt": 32768}
N_OTHER = 16384


def _threads():
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(round(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return n


@pytest.fixture(scope="module")
def cpu_side(oracle_lib):
    """(kind, bc7(blocks, opt_bytes, plan_bytes), other(fmt, blocks, opt_bytes)) + the RCPPS table the GPU must use"""
    threads = _threads()
    if pyref.RefLib.available(fast=True) and pyref.RefLib.available():
        fast, canon = pyref.RefLib(fast=True), pyref.RefLib()
        rcp = fast.probe_rcp()

        def bc7(b, ob, pb):
            out, done, _ = fast.encode_mt("bc7", b, ob, pb, threads=threads, budget_s=300.0, chunk_blocks=64)
            assert done == b.shape[0]
            return out

        def other(fmt, b, ob):
            out, done, _ = canon.encode_mt(fmt, b, ob, None, threads=threads, budget_s=300.0, chunk_blocks=64)
            assert done == b.shape[0]
            return out
        return "reference", rcp, bc7, other
    rcp = oracle_lib.probe_rcp()
    bc7 = lambda b, ob, pb: oracle_lib.encode_bc7(b, ob, pb, rcp, threads)
    other = lambda fmt, b, ob: (oracle_lib.encode_bc6h(b, ob, fmt == "bc6hs", rcp, threads) if fmt.startswith("bc6h")
                                else oracle_lib.encode_etc2(b, ob, 1, threads))
    return "port", rcp, bc7, other


@pytest.mark.parametrize("variant", ["default", "better", "ultra_pt"])
def test_bc7_families_vs_reference(gpu_ctx, cpu_side, variant):
    from convectionkernels_amd import api, synth
    kind, rcp, bc7, _ = cpu_side
    gpu_ctx.set_rcp_table(rcp)
    opt = {"default": api.Options(), "better": api.Options(flags=api.Flags.Better),
           "ultra_pt": api.Options(flags=api.Flags.Ultra | api.Flags.BC7_RespectPunchThrough)}[variant]
    plan = api.BC7EncodingPlan()
    ob = np.frombuffer(opt.tobytes(), np.uint8).copy()
    pb = np.frombuffer(plan.tobytes(), np.uint8).copy()
    t0 = time.time()
    report = {}
    for name, b in synth.content_families(N_BC7[variant], seed=20260930 + len(variant)).items():
        got = gpu_ctx.encode_bc7(b, opt, plan)
        exp = bc7(b, ob, pb)
        report[name] = int((got != exp).any(axis=1).sum())
    print("BC7 %s vs %s: %s (%.1f s)" % (variant, kind, report, time.time() - t0))
    assert all(v == 0 for v in report.values()), report


def test_bc6h_and_etc2_vs_reference(gpu_ctx, cpu_side):
    from convectionkernels_amd import api, synth
    kind, rcp, _, other = cpu_side
    gpu_ctx.set_rcp_table(rcp)
    opt = api.Options()
    ob = np.frombuffer(opt.tobytes(), np.uint8).copy()
    rng = np.random.Generator(np.random.PCG64(77))
    report = {}
    # HDR: wide-range noise and narrow-range blocks (close exponents), unsigned and signed
    q = N_OTHER // 4
    wide = rng.integers(0, 0x7C00, (q, 16, 4)).astype(np.uint16)
    base = rng.integers(0x3000, 0x7000, (q, 1, 3))
    narrow = np.zeros((q, 16, 4), np.uint16)
    narrow[:, :, :3] = (base + rng.integers(-60, 61, (q, 16, 3))).astype(np.uint16)
    fam_hdr = synth.hdr_content_families(q - q % 24, seed=4242)  # smooth ramps / narrow range: where the deltas do fit
    third = (q - q % 24) // 3
    mixed = np.concatenate([v[:third].view(np.uint16) for v in fam_hdr.values()])
    hdr = np.concatenate([wide, narrow, mixed, mixed[::-1]])[:N_OTHER]
    hdr = np.ascontiguousarray(hdr[:hdr.shape[0] // 64 * 64])  # whole 64-block chunks of the reference's worker threads
    hdr[:, :, 3] = 0x3C00
    for sg in (False, True):
        h = hdr.copy()
        if sg:
            h[:, :, :3] |= (rng.integers(0, 2, (h.shape[0], 16, 3)) << 15).astype(np.uint16)
        b = h.view(np.int16)
        got = gpu_ctx.encode_bc6h(b, opt, signed=sg)
        exp = other("bc6hs" if sg else "bc6hu", b, ob)
        report["bc6h" + ("s" if sg else "u")] = int((got != exp).any(axis=1).sum())
    fam = synth.content_families(N_OTHER // 8, seed=99)
    ldr = np.concatenate(list(fam.values()))
    got = gpu_ctx.encode_etc2_rgba(ldr, opt)
    exp = other("etc2rgba", ldr, ob)
    report["etc2rgba"] = int((got != exp).any(axis=1).sum())
    print("BC6H / ETC2 RGBA vs %s: %s" % (kind, report))
    assert all(v == 0 for v in report.values()), report
