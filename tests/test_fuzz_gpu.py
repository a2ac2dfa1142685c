"""Options fuzz on the MI355X: 32 seeded random `cvtt::Options` sets (any flag bits; zero, tiny, huge and
negative channel weights; thresholds outside [0,1]; refine rounds -1..9; seed points -1..6), random `BC7FineTuningParams`
plans and hand-written `BC7EncodingPlan` byte patterns, every output block compared with the REAL reference on the box
(oracle/_ref; the C restatement where the reference build did not travel).  BC7 runs with the branch-and-bound on AND off:
the bounds' soundness argument is written for positive weights, the reference divides by the raw weight
(EndpointSelector.h:61-66), so every weight sign / magnitude goes through both searches here.
Content: 4 096 mixed LDR blocks (tests/content.py: noise, gradients, solid, two-colour, punch-through alpha, the 250 / 251
alpha threshold, saturated values, mixes inside a group) and 4 096 mixed HDR blocks (normals, ramps, denormals, inf / nan
patterns, signed)."""
import os
import time

import numpy as np
import pytest

import content
import fuzz_options as fo
from oracle import pyref

pytestmark = pytest.mark.gpu

N_SETS = int(os.environ.get("CVTT_FUZZ_SETS", "32"))  # the driver's run: 32; builder runs with more (profiles/r05/options_fuzz_gpu_160.log)
GROUPS = 512  # 4 096 blocks per format and option set


def _threads():
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(round(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return n


class _Cpu:
    """the reference (or the restatement) on all host threads"""

    def __init__(self, oracle_lib):
        self.threads = _threads()
        self.orc = oracle_lib
        self.ref = pyref.RefLib() if pyref.RefLib.available() else None
        self.kind = "reference" if self.ref else "port"
        self.rcp = (self.ref or self.orc).probe_rcp()

    def run(self, fmt, blocks, ob, pb=None):
        if self.ref:
            out, done, _ = self.ref.encode_mt(fmt, blocks, ob, pb, threads=self.threads, budget_s=600.0, chunk_blocks=64)
            assert done == blocks.shape[0]
            return out
        if fmt == "bc7":
            return self.orc.encode_bc7(blocks, ob, pb, self.rcp, self.threads)
        if fmt == "bc1":
            return self.orc.encode_bc1(blocks, ob, self.rcp, self.threads)
        if fmt.startswith("bc6h"):
            return self.orc.encode_bc6h(blocks, ob, fmt == "bc6hs", self.rcp, self.threads)
        return self.orc.encode_etc2(blocks, ob, 1, self.threads)


@pytest.fixture(scope="module")
def cpu(oracle_lib):
    return _Cpu(oracle_lib)


@pytest.fixture(scope="module")
def ldr():
    return content.mixed_ldr_blocks(424242, GROUPS)


def _bytes(s):
    return np.frombuffer(s.tobytes(), np.uint8).copy()


def _report(tag, bad, t0):
    worst = {k: v for k, v in bad.items() if v}
    print("%s: %d cases, %d with mismatches (%.1f s)" % (tag, len(bad), len(worst), time.time() - t0))
    for k, v in worst.items():
        print("   MISMATCH %s: %d blocks" % (k, v))
    assert not worst, worst


def test_bc7_options_fuzz_pruned_and_exhaustive(gpu_ctx, cpu, ldr):
    from convectionkernels_amd import api
    gpu_ctx.set_rcp_table(cpu.rcp)
    plan = api.BC7EncodingPlan()
    pb = _bytes(plan)
    bad, t0 = {}, time.time()
    try:
        for i, o in enumerate(fo.options_sets(N_SETS)):
            exp = cpu.run("bc7", ldr, _bytes(o), pb)
            for ex in (False, True):
                gpu_ctx.set_exhaustive(ex)
                got = gpu_ctx.encode_bc7(ldr, o, plan)
                bad["#%d %s %s" % (i, "exhaustive" if ex else "pruned", fo.describe(o))] = int((got != exp).any(axis=1).sum())
    finally:
        gpu_ctx.set_exhaustive(False)
    _report("BC7 options fuzz vs %s" % cpu.kind, bad, t0)


def test_bc7_plan_fuzz(gpu_ctx, cpu, ldr):
    from convectionkernels_amd import api
    gpu_ctx.set_rcp_table(cpu.rcp)
    plans = []
    for i, ft in enumerate(fo.fine_tuning_sets(12)):
        p = api.BC7EncodingPlan()
        api.ConfigureBC7EncodingPlanFromFineTuningParams(p, ft)
        plans.append(("fine_tuning_%d" % i, p))
    plans += fo.hand_written_plans()
    opts = [api.Options(), api.Options(flags=api.Flags.Ultra | api.Flags.BC7_RespectPunchThrough, refineRoundsBC7=3),
            fo.options_sets(N_SETS)[13]]
    bad, t0 = {}, time.time()
    try:
        for name, p in plans:
            for j, o in enumerate(opts):
                exp = cpu.run("bc7", ldr, _bytes(o), _bytes(p))
                for ex in (False, True):
                    gpu_ctx.set_exhaustive(ex)
                    got = gpu_ctx.encode_bc7(ldr, o, p)
                    bad["%s opt%d %s" % (name, j, "exhaustive" if ex else "pruned")] = int((got != exp).any(axis=1).sum())
    finally:
        gpu_ctx.set_exhaustive(False)
    _report("BC7 plan fuzz vs %s" % cpu.kind, bad, t0)


def test_bc1_options_fuzz(gpu_ctx, cpu, ldr):
    gpu_ctx.set_rcp_table(cpu.rcp)
    bad, t0 = {}, time.time()
    for i, o in enumerate(fo.options_sets(N_SETS)):
        got = gpu_ctx.encode_bc1(ldr, o)
        exp = cpu.run("bc1", ldr, _bytes(o))
        bad["#%d %s" % (i, fo.describe(o))] = int((got != exp).any(axis=1).sum())
    _report("BC1 options fuzz vs %s" % cpu.kind, bad, t0)


@pytest.mark.parametrize("signed", [False, True])
def test_bc6h_options_fuzz(gpu_ctx, cpu, signed):
    gpu_ctx.set_rcp_table(cpu.rcp)
    hdr = content.mixed_hdr_blocks(31337 + int(signed), GROUPS, signed=signed)
    bad, t0 = {}, time.time()
    for i, o in enumerate(fo.options_sets(N_SETS)):
        got = gpu_ctx.encode_bc6h(hdr, o, signed=signed)
        exp = cpu.run("bc6hs" if signed else "bc6hu", hdr, _bytes(o))
        bad["#%d %s" % (i, fo.describe(o))] = int((got != exp).any(axis=1).sum())
    _report("BC6H%s options fuzz vs %s" % ("S" if signed else "U", cpu.kind), bad, t0)


def test_etc2_rgba_options_fuzz(gpu_ctx, cpu, ldr):
    gpu_ctx.set_rcp_table(cpu.rcp)
    bad, t0 = {}, time.time()
    for i, o in enumerate(fo.options_sets(N_SETS)):
        got = gpu_ctx.encode_etc2_rgba(ldr, o)
        exp = cpu.run("etc2rgba", ldr, _bytes(o))
        bad["#%d %s" % (i, fo.describe(o))] = int((got != exp).any(axis=1).sum())
    _report("ETC2 RGBA options fuzz vs %s" % cpu.kind, bad, t0)


# ---- the (f)4 formats and the decoders: 10 fuzzed Options sets each (5 corners + 5 random), against the real reference on the box ----
N_WIDE = int(os.environ.get("CVTT_FUZZ_WIDE_SETS", "10"))


def _wide_sets():
    """the corners that matter for these formats (zero / negative weights, refine rounds and seed points out of range, thresholds
    outside [0, 1], FakeBT709 / Uniform / paranoid bits through the random flags) + seeded random sets"""
    sets = fo.options_sets(32)
    pick = [2, 5, 7, 8, 9] + list(range(10, 32)) + [0, 1, 3, 4, 6]
    return [sets[i] for i in pick[:N_WIDE]]


def _ref_or_skip(cpu):
    if cpu.ref is None:
        pytest.skip("oracle/_ref did not travel: the reference build is the checker of this test")
    return cpu.ref


@pytest.mark.parametrize("fmt", ["bc2", "bc3", "bc4u", "bc4s", "bc5u", "bc5s"])
def test_s3tc_family_options_fuzz(gpu_ctx, cpu, ldr, fmt):
    ref = _ref_or_skip(cpu)
    gpu_ctx.set_rcp_table(cpu.rcp)
    code = {"bc2": 2, "bc3": 3, "bc4u": 4, "bc4s": 5, "bc5u": 6, "bc5s": 7}[fmt]
    enc = {"bc2": lambda b, o: gpu_ctx.encode_bc2(b, o), "bc3": lambda b, o: gpu_ctx.encode_bc3(b, o),
           "bc4u": lambda b, o: gpu_ctx.encode_bc4(b, o, signed=False), "bc4s": lambda b, o: gpu_ctx.encode_bc4(b, o, signed=True),
           "bc5u": lambda b, o: gpu_ctx.encode_bc5(b, o, signed=False), "bc5s": lambda b, o: gpu_ctx.encode_bc5(b, o, signed=True)}[fmt]
    bad, t0 = {}, time.time()
    for i, o in enumerate(_wide_sets()):
        exp = ref.encode_s3tc(ldr, _bytes(o), code)
        bad["#%d %s" % (i, fo.describe(o))] = int((enc(ldr, o) != exp).any(axis=1).sum())
    _report("%s options fuzz vs reference" % fmt.upper(), bad, t0)


@pytest.mark.parametrize("mode,name", [(0, "etc2"), (3, "etc1"), (4, "etc2_punchthrough"), (2, "etc2_alpha")])
def test_etc_family_options_fuzz(gpu_ctx, cpu, ldr, mode, name):
    """EncodeETC2 / EncodeETC1 / EncodeETC2PunchthroughAlpha / EncodeETC2Alpha; every second set allocates the scratch with OTHER
    Options than it encodes with (the chroma axes belong to AllocETC2Data's, ETC.cpp:3117-3145)"""
    ref = _ref_or_skip(cpu)
    from convectionkernels_amd import api
    enc = {0: lambda b, o, ao: gpu_ctx.encode_etc2(b, o, compression_data=ao), 3: lambda b, o, ao: gpu_ctx.encode_etc1(b, o),
           4: lambda b, o, ao: gpu_ctx.encode_etc2_punchthrough_alpha(b, o, compression_data=ao),
           2: lambda b, o, ao: gpu_ctx.encode_etc2_alpha(b, o)}[mode]
    sets = _wide_sets()
    bad, t0 = {}, time.time()
    for i, o in enumerate(sets):
        ao = sets[(i + 3) % len(sets)] if (i & 1) and mode in (0, 4) else None
        exp = ref.encode_etc2(ldr, _bytes(o), mode, alloc_options=None if ao is None else _bytes(ao))
        got = enc(ldr, o, ao)
        bad["#%d %s%s" % (i, fo.describe(o), " alloc#%d" % ((i + 3) % len(sets)) if ao is not None else "")] = int((got != exp).any(axis=1).sum())
    _report("%s options fuzz vs reference" % name, bad, t0)


@pytest.mark.parametrize("signed", [False, True])
def test_eac_r11_options_fuzz(gpu_ctx, cpu, signed):
    """EncodeETC2Alpha11: 16 int16 per block -- in range, at the clamps and far outside (the reference clamps, ETC.cpp:2087-2114)"""
    ref = _ref_or_skip(cpu)
    rng = np.random.Generator(np.random.PCG64(99 + int(signed)))
    n = GROUPS * 8
    lo, hi = (-1023, 1023) if signed else (0, 2047)
    blocks = rng.integers(lo, hi + 1, (n, 16)).astype(np.int16)
    blocks[n // 4:n // 2] = (blocks[n // 4:n // 2] // 64) * 64                      # few distinct values
    blocks[n // 2:n // 2 + n // 8] = rng.integers(-32768, 32768, (n // 8, 16))       # far outside the range
    blocks[n // 2 + n // 8:n // 2 + n // 4] = rng.choice([lo, hi, lo + 1, hi - 1, 0], (n // 8, 16))
    base = rng.integers(lo, hi + 1, (n // 4, 1))
    blocks[3 * n // 4:] = np.clip(base + rng.integers(-6, 7, (n // 4, 16)), -32768, 32767)  # narrow ranges: small multipliers
    bad, t0 = {}, time.time()
    for i, o in enumerate(_wide_sets()):
        exp = ref.encode_eac11(blocks, _bytes(o), signed=signed)
        got = gpu_ctx.encode_etc2_alpha11(blocks, signed=signed, options=o)
        bad["#%d %s" % (i, fo.describe(o))] = int((got != exp).any(axis=1).sum())
    _report("EAC R11 %s options fuzz vs reference" % ("signed" if signed else "unsigned"), bad, t0)


def test_decoders_on_random_bytes(gpu_ctx, cpu):
    """DecodeBC7 / DecodeBC6HU / DecodeBC6HS take no Options: the fuzz is over the packed bytes -- uniform random blocks (every
    mode, reserved modes and illegal headers included), each mode forced in turn, and what the encoders produce"""
    ref = _ref_or_skip(cpu)
    rng = np.random.Generator(np.random.PCG64(2718))
    n = 8192
    raw = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    bc7 = raw.copy()
    for m in range(9):  # mode m = m zero bits then a one (m = 8: the reserved all-zero mode byte)
        sl = slice(n // 2 + m * 256, n // 2 + (m + 1) * 256)
        bc7[sl, 0] = ((bc7[sl, 0].astype(np.uint16) << (m + 1)) | (1 << m)).astype(np.uint8) if m < 8 else 0
    got = gpu_ctx.decode_bc7(bc7)
    exp = ref.decode_bc7(bc7)
    assert int((got != exp).any(axis=(1, 2)).sum()) == 0
    bc6 = raw.copy()
    modes = [0, 1, 2, 6, 10, 14, 18, 22, 26, 30, 3, 7, 11, 15, 19, 23, 27, 31]  # the 14 real 5-bit / 2-bit headers and the reserved ones
    for k, hdr in enumerate(modes):
        sl = slice(n // 2 + k * 128, n // 2 + (k + 1) * 128)
        bc6[sl, 0] = (bc6[sl, 0] & (0xFC if hdr < 2 else 0xE0)) | hdr
    for signed in (False, True):
        got = gpu_ctx.decode_bc6h(bc6, signed=signed)
        exp = ref.decode_bc6h(bc6, signed=signed)
        assert int((got != exp).any(axis=(1, 2)).sum()) == 0, "bc6h signed=%s" % signed
    # round trip of real encoder output, fuzzed options
    ldr = content.mixed_ldr_blocks(5150, 128)
    hdr = content.mixed_hdr_blocks(5151, 128)
    from convectionkernels_amd import api
    for o in _wide_sets()[:4]:
        packed = gpu_ctx.encode_bc7(ldr, o, api.BC7EncodingPlan())
        assert (gpu_ctx.decode_bc7(packed) == ref.decode_bc7(packed)).all()
        packed = gpu_ctx.encode_bc6h(hdr, o, signed=False)
        assert (gpu_ctx.decode_bc6h(packed, signed=False) == ref.decode_bc6h(packed, signed=False)).all()
