"""The option / plan sets of the GPU fuzz (tests/fuzz_options.py), on a few blocks on the CPU: the C restatement against the
compiled reference -- the checker of tests/test_fuzz_gpu.py has to agree with the reference on exactly these inputs --
and the generators themselves (deterministic, every value set hit)."""
import numpy as np
import pytest

import content
import fuzz_options as fo
from convectionkernels_amd import api


def _bytes(s):
    return np.frombuffer(s.tobytes(), np.uint8).copy()


def test_generators_are_deterministic_and_cover_the_value_sets():
    a, b = fo.options_sets(32), fo.options_sets(32)
    assert [x.tobytes() for x in a] == [x.tobytes() for x in b]
    ws = {round(float(w), 6) for o in a for w in (o.redWeight, o.greenWeight, o.blueWeight, o.alphaWeight)}
    assert {0.0, 0.001, 0.1, 1.0, 3.0, 100.0, -1.0} <= ws
    assert {o.seedPoints for o in a} >= set(range(-1, 7))
    assert min(o.refineRoundsBC7 for o in a) == -1 and max(o.refineRoundsBC7 for o in a) == 9
    assert {round(o.threshold, 2) for o in a} >= {-1.0, 0.0, 0.25, 1.0, 2.0}
    assert len(fo.fine_tuning_sets(12)) == 12 and len(fo.hand_written_plans()) == 5


@pytest.mark.parametrize("first", [0, 16])
def test_restatement_equals_reference_on_fuzz_options(oracle_lib, ref_lib, first):
    rcp = ref_lib.probe_rcp()
    ldr = content.mixed_ldr_blocks(5, 12)
    hdr = content.mixed_hdr_blocks(6, 8)
    hdrs = content.mixed_hdr_blocks(7, 8, signed=True)
    plan = _bytes(api.BC7EncodingPlan())
    for i, o in enumerate(fo.options_sets(32)[first:first + 16]):
        ob = _bytes(o)
        tag = "#%d %s" % (first + i, fo.describe(o))
        assert (ref_lib.encode_bc7(ldr, ob, plan) == oracle_lib.encode_bc7(ldr, ob, plan, rcp, 4)).all(), tag
        assert (ref_lib.encode_bc1(ldr, ob) == oracle_lib.encode_bc1(ldr, ob, rcp, 4)).all(), tag
        assert (ref_lib.encode_bc6h(hdr, ob, False) == oracle_lib.encode_bc6h(hdr, ob, False, rcp, 4)).all(), tag
        assert (ref_lib.encode_bc6h(hdrs, ob, True) == oracle_lib.encode_bc6h(hdrs, ob, True, rcp, 4)).all(), tag
        assert (ref_lib.encode_etc2(ldr, ob, 1) == oracle_lib.encode_etc2(ldr, ob, 1, 4)).all(), tag


def test_restatement_equals_reference_on_fuzz_plans(oracle_lib, ref_lib):
    rcp = ref_lib.probe_rcp()
    ldr = content.mixed_ldr_blocks(5, 12)
    opt = _bytes(api.Options())
    for i, ft in enumerate(fo.fine_tuning_sets(6)):
        p = api.BC7EncodingPlan()
        api.ConfigureBC7EncodingPlanFromFineTuningParams(p, ft)
        pb = _bytes(p)
        assert (ref_lib.plan_from_finetune(_bytes(ft)) == pb).all(), i
        assert (ref_lib.encode_bc7(ldr, opt, pb) == oracle_lib.encode_bc7(ldr, opt, pb, rcp, 4)).all(), i
    for name, p in fo.hand_written_plans():
        pb = _bytes(p)
        assert (ref_lib.encode_bc7(ldr, opt, pb) == oracle_lib.encode_bc7(ldr, opt, pb, rcp, 4)).all(), name
