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
