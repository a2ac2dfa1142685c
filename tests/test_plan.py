"""ConfigureBC7EncodingPlanFromQuality / FromFineTuningParams: host-side, no device needed.
Plans must be byte-identical to the reference's (goldens: tests/golden/bc7_plans.npz, made with oracle/_ref)."""
import os

import numpy as np
import pytest

from convectionkernels_amd import api

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _plan_bytes(p):
    return np.frombuffer(p.tobytes(), np.uint8)


def test_every_quality_matches_the_reference():
    g = np.load(os.path.join(GOLD, "bc7_plans.npz"))
    for q in range(1, 101):
        p = api.BC7EncodingPlan()
        api.ConfigureBC7EncodingPlanFromQuality(p, q)
        assert (_plan_bytes(p) == g["quality"][q - 1]).all(), q
    for q, same in ((0, 1), (-7, 1), (101, 100), (1000, 100)):  # clamped, BC67.cpp:3295-3298
        p = api.BC7EncodingPlan()
        api.ConfigureBC7EncodingPlanFromQuality(p, q)
        assert (_plan_bytes(p) == g["quality"][same - 1]).all()


def test_quality_plans_are_the_ones_the_encoder_goldens_used():
    g = np.load(os.path.join(GOLD, "bc7_mixed.npz"))
    for q in (1, 20, 60, 100):
        p = api.BC7EncodingPlan()
        api.ConfigureBC7EncodingPlanFromQuality(p, q)
        assert p.tobytes() == g["plan_quality%d" % q].tobytes()


def test_fine_tuning_params():
    g = np.load(os.path.join(GOLD, "bc7_plans.npz"))
    for ft, exp in zip(g["finetune"], g["finetune_plans"]):
        p = api.BC7EncodingPlan()
        assert api.ConfigureBC7EncodingPlanFromFineTuningParams(p, api.BC7FineTuningParams.frombytes(ft.tobytes())) is True
        assert (_plan_bytes(p) == exp).all()
    # default parameters give the default plan's search space (every partition, 4 seed points)
    p, d = api.BC7EncodingPlan(), api.BC7EncodingPlan()
    api.ConfigureBC7EncodingPlanFromFineTuningParams(p, api.BC7FineTuningParams())
    assert p.mode1PartitionEnabled == d.mode1PartitionEnabled and p.mode0PartitionEnabled == 0xFFFF
    assert p.rgbNumShapesToEvaluate == 242 and p.rgbaNumShapesToEvaluate == 129  # no RGB mode uses the whole-block shape
    assert p.mode7RGBPartitionEnabled == 0  # = RGBA mask & ~mode-3 mask (BC67.cpp:3480), unlike BC7EncodingPlan()


def test_quality_ladder_is_monotone():
    prev = -1
    for q in range(1, 101):
        p = api.BC7EncodingPlan()
        api.ConfigureBC7EncodingPlanFromQuality(p, q)
        n = p.rgbNumShapesToEvaluate + p.rgbaNumShapesToEvaluate
        assert n >= prev
        prev = n


def test_against_reference_on_fresh_parameters(ref_lib):
    rng = np.random.default_rng(77)
    for _ in range(64):
        ft = rng.integers(0, 5, 285).astype(np.uint8)
        p = api.BC7EncodingPlan()
        api.ConfigureBC7EncodingPlanFromFineTuningParams(p, api.BC7FineTuningParams.frombytes(ft.tobytes()))
        assert p.tobytes() == ref_lib.plan_from_finetune(ft).tobytes()


@pytest.mark.gpu
def test_gpu_encode_with_configured_plan(gpu_ctx):
    g = np.load(os.path.join(GOLD, "bc7_mixed.npz"))
    gpu_ctx.set_rcp_table(g["rcp"])
    p = api.BC7EncodingPlan()
    api.ConfigureBC7EncodingPlanFromQuality(p, 20)
    out = gpu_ctx.encode_bc7(g["blocks"], api.Options.frombytes(g["opt_quality20"]), p)
    assert (out == g["out_quality20"]).all()
