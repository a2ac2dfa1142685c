"""Seeded random `cvtt::Options` sets, `BC7FineTuningParams` plans and hand-written `BC7EncodingPlan` byte patterns for the
options fuzz (tests/test_fuzz_gpu.py on the MI355X against oracle/_ref; tests/test_fuzz_cpu.py: the C restatement against
oracle/_ref on a few blocks).  The value sets are the ones the reference does not guard against: it takes the weights
as they come (Util.cpp:62-73, EndpointSelector.h:61-66 divides by them), clamps refine rounds and seed points only where
BC67.cpp:1044-1045 / 2667-2670 / S3TC.cpp say so, and reads the plan bytes without validation."""
import numpy as np

from convectionkernels_amd import api

WEIGHTS = (0.0, 1e-3, 0.1, 1.0, 3.0, 100.0, -1.0)
THRESHOLDS = (-1.0, 0.0, 0.25, 1.0, 2.0)
ALL_FLAGS = 0xFF8  # every bit of cvtt::Flags (ConvectionKernels.h:33-69)


def options_sets(n, seed=20261001):
    """n Options: flags any bits, weights / threshold from the sets above, every refineRounds* in -1..9, seedPoints -1..6.
    The first sets are hand-picked corners (all-zero weights would make every error 0 == every candidate ties: kept, it is
    what the reference computes too)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sets = []
    corners = [
        dict(redWeight=0.0, greenWeight=1.0, blueWeight=1.0, alphaWeight=1.0),
        dict(redWeight=1.0, greenWeight=1.0, blueWeight=1.0, alphaWeight=0.0),
        dict(redWeight=-1.0, greenWeight=1.0, blueWeight=3.0, alphaWeight=1.0),
        dict(redWeight=100.0, greenWeight=1e-3, blueWeight=0.1, alphaWeight=100.0),
        dict(redWeight=1e-3, greenWeight=1e-3, blueWeight=1e-3, alphaWeight=1e-3),
        dict(redWeight=0.0, greenWeight=0.0, blueWeight=0.0, alphaWeight=0.0),
        dict(refineRoundsBC7=-1, refineRoundsBC6H=-1, refineRoundsIIC=-1, refineRoundsS3TC=-1, seedPoints=-1),
        dict(refineRoundsBC7=9, refineRoundsBC6H=9, refineRoundsIIC=9, refineRoundsS3TC=9, seedPoints=6),
        dict(threshold=-1.0, flags=api.Flags.Default | api.Flags.BC7_RespectPunchThrough),
        dict(threshold=2.0, flags=api.Flags.Ultra | api.Flags.BC7_RespectPunchThrough | api.Flags.Uniform),
    ]
    for c in corners[:n]:
        sets.append(api.Options(**c))
    while len(sets) < n:
        o = api.Options()
        o.flags = int(rng.integers(0, 1 << 12)) & ALL_FLAGS
        o.threshold = float(rng.choice(THRESHOLDS))
        o.redWeight, o.greenWeight, o.blueWeight, o.alphaWeight = (float(rng.choice(WEIGHTS)) for _ in range(4))
        o.refineRoundsBC7, o.refineRoundsBC6H, o.refineRoundsIIC, o.refineRoundsS3TC = (int(rng.integers(-1, 10)) for _ in range(4))
        o.seedPoints = int(rng.integers(-1, 7))
        sets.append(o)
    return sets


def describe(o):
    return ("flags=0x%03x thr=%g w=(%g,%g,%g,%g) refine=(%d,%d,%d,%d) sp=%d"
            % (o.flags, o.threshold, o.redWeight, o.greenWeight, o.blueWeight, o.alphaWeight,
               o.refineRoundsBC7, o.refineRoundsBC6H, o.refineRoundsIIC, o.refineRoundsS3TC, o.seedPoints))


def fine_tuning_sets(n, seed=515):
    """random BC7FineTuningParams: seed points 0..4 per mode / partition with different densities (0 = partition off),
    occasionally above MaxTweakRounds (the encoder clamps, BC67.cpp:1244, 1732)"""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = []
    for i in range(n):
        p = api.BC7FineTuningParams()
        density = (0.05, 0.3, 0.7, 1.0)[i % 4]
        raw = rng.integers(1, 5, 285).astype(np.uint8)
        raw[rng.random(285) >= density] = 0
        if i % 3 == 2:
            raw[rng.random(285) < 0.05] = rng.integers(5, 256)
        if i % 5 == 4:
            raw[16 + 64 + 64 + 64:16 + 64 + 64 + 64 + 13] = 0  # no mode 4 / 5 / 6 at all
        out.append(api.BC7FineTuningParams.frombytes(raw.tobytes()))
    return out


def hand_written_plans(seed=77):
    """BC7EncodingPlan byte patterns no Configure* call produces.  Every shape an enabled partition needs stays in the shape
    lists (the reference would read uninitialised seeds otherwise), everything else is arbitrary: seed counts 0..255, sparse
    and empty partition masks, permuted shape lists, mode 6 off, mode-4/5 seed points mixed with zeros and values above 4,
    an RGB mode-7 mask that is not `rgba & ~mode3`."""
    rng = np.random.Generator(np.random.PCG64(seed))
    plans = []

    p = api.BC7EncodingPlan()  # seed counts of every size, including 0 (shape skipped) and > MaxTweakRounds
    for i in range(243):
        p.seedPointsForShapeRGB[i] = int(rng.choice([0, 1, 2, 3, 4, 5, 17, 255]))
    for i in range(129):
        p.seedPointsForShapeRGBA[i] = int(rng.choice([0, 1, 2, 3, 4, 6, 200]))
    plans.append(("seed_counts", p))

    p = api.BC7EncodingPlan()  # sparse masks, mode 6 off, permuted lists
    p.mode0PartitionEnabled = 0x8421
    p.mode1PartitionEnabled = int(rng.integers(0, 1 << 63))
    p.mode2PartitionEnabled = 0x00000000FFFF0000
    p.mode3PartitionEnabled = int(rng.integers(0, 1 << 63)) & 0x0F0F0F0F0F0F0F0F
    p.mode7RGBAPartitionEnabled = int(rng.integers(0, 1 << 63))
    p.mode7RGBPartitionEnabled = int(rng.integers(0, 1 << 63))
    p.mode6Enabled = 0
    perm = rng.permutation(243)
    for i in range(243):
        p.rgbShapeList[i] = int(perm[i])
    perm = rng.permutation(129)
    for i in range(129):
        p.rgbaShapeList[i] = int(perm[i])
    plans.append(("sparse_permuted", p))

    p = api.BC7EncodingPlan()  # only the dual-plane modes, uneven seed points
    p.mode0PartitionEnabled = 0
    p.mode1PartitionEnabled = p.mode2PartitionEnabled = p.mode3PartitionEnabled = 0
    p.mode7RGBAPartitionEnabled = p.mode7RGBPartitionEnabled = 0
    p.mode6Enabled = 0
    vals = [0, 1, 2, 3, 4, 9, 0, 2, 255, 1, 0, 4]
    for r in range(4):
        p.mode4SP[r][0] = vals[r]
        p.mode4SP[r][1] = vals[4 + r]
        p.mode5SP[r] = vals[8 + r]
    plans.append(("dual_plane_only", p))

    p = api.BC7EncodingPlan()  # mode 6 and mode 7 alone (mode6Enabled is a C++ bool in the reference: bytes other than 0 / 1 are undefined there)
    p.mode0PartitionEnabled = 0
    p.mode1PartitionEnabled = p.mode2PartitionEnabled = p.mode3PartitionEnabled = 0
    p.mode7RGBAPartitionEnabled = 0xFFFFFFFF00000000
    p.mode7RGBPartitionEnabled = 0x00000000FFFFFFFF
    p.mode6Enabled = 1
    for r in range(4):
        p.mode4SP[r][0] = p.mode4SP[r][1] = p.mode5SP[r] = 0
    plans.append(("mode6_mode7", p))

    p = api.BC7EncodingPlan()  # nothing but one two-subset partition: tiny candidate sets
    p.mode0PartitionEnabled = 0
    p.mode1PartitionEnabled = 1 << 13
    p.mode2PartitionEnabled = p.mode3PartitionEnabled = 0
    p.mode7RGBAPartitionEnabled = p.mode7RGBPartitionEnabled = 0
    p.mode6Enabled = 0
    for r in range(4):
        p.mode4SP[r][0] = p.mode4SP[r][1] = p.mode5SP[r] = 0
    plans.append(("single_partition", p))
    return plans
