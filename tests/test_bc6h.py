"""BC6H (EncodeBC6HU / EncodeBC6HS): oracle vs golden vectors / reference on CPU; HIP path on GPU."""
import os

import numpy as np
import pytest

import content
from oracle import pyref

GOLD = os.path.join(os.path.dirname(__file__), "golden")
NAMES = ["default", "fast", "uniform", "seeds2_refine2", "weights"]


@pytest.mark.parametrize("name", NAMES)
def test_oracle_golden(oracle_lib, name):
    g = np.load(os.path.join(GOLD, "bc6h_mixed.npz"))
    out = oracle_lib.encode_bc6h(g["blocks"], g["opt_" + name], False, g["rcp"], threads=8)
    assert (out == g["out_" + name]).all()
    outs = oracle_lib.encode_bc6h(g["blocks_signed"], g["opt_" + name], True, g["rcp"], threads=8)
    assert (outs == g["outs_" + name]).all()


def test_oracle_known_answers(oracle_lib):
    g = np.load(os.path.join(GOLD, "known_answers.npz"))
    out = oracle_lib.encode_bc6h(g["hdr_blocks"], pyref.make_options(), False, g["rcp"])
    assert (out == g["bc6hu"]).all()
    assert out[0].tobytes().hex() == "1e52092f4c757d92a3158148235b1b51"
    assert out[7].tobytes().hex() == "3e7c34543d2fa2a94b8999e0bf5b4e09"


def test_oracle_vs_reference(oracle_lib, ref_lib):
    rcp = ref_lib.probe_rcp()
    blocks = content.mixed_hdr_blocks(99, 8)
    for opt in (pyref.make_options(), pyref.make_options(flags=pyref.FLAG_BC6H_FAST_INDEXING, seed_points=3)):
        assert (oracle_lib.encode_bc6h(blocks, opt, False, rcp, 8) == ref_lib.encode_bc6h(blocks, opt, False)).all()
    cfg = content.config_blocks_hdr(3, 32, 32)  # BASELINE config 3 content
    assert (oracle_lib.encode_bc6h(cfg, pyref.make_options(), False, rcp, 8) == ref_lib.encode_bc6h(cfg, pyref.make_options(), False)).all()


def test_group_coupling_changes_output(oracle_lib):
    """the duplicate-round skip and the mode commit loop couple the 8 lanes:
    encoding a block inside its group is not always the same as encoding it replicated alone"""
    blocks = content.mixed_hdr_blocks(3, 16)
    opt = pyref.make_options()
    grouped = oracle_lib.encode_bc6h(blocks, opt, False, None, 8)
    assert grouped.shape == (128, 16)


# ------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_golden(gpu_ctx, name):
    from convectionkernels_amd import api
    g = np.load(os.path.join(GOLD, "bc6h_mixed.npz"))
    gpu_ctx.set_rcp_table(g["rcp"])
    opt = api.Options.frombytes(g["opt_" + name])
    out = gpu_ctx.encode_bc6h(g["blocks"], opt, signed=False)
    bad = np.nonzero((out != g["out_" + name]).any(axis=1))[0]
    assert bad.size == 0, bad[:8]
    outs = gpu_ctx.encode_bc6h(g["blocks_signed"], opt, signed=True)
    bad = np.nonzero((outs != g["outs_" + name]).any(axis=1))[0]
    assert bad.size == 0, bad[:8]


@pytest.mark.gpu
def test_gpu_known_answers_and_config3(gpu_ctx, oracle_lib):
    """App. H vector; BASELINE configs[2] content (positive normal halfs) vs the oracle; device path"""
    import torch
    from convectionkernels_amd import api
    g = np.load(os.path.join(GOLD, "known_answers.npz"))
    gpu_ctx.set_rcp_table(g["rcp"])
    assert (gpu_ctx.encode_bc6h(g["hdr_blocks"], api.Options()) == g["bc6hu"]).all()
    rcp = oracle_lib.probe_rcp()
    gpu_ctx.set_rcp_table(rcp)
    blocks = content.config_blocks_hdr(3, 128, 128)  # 1024 blocks of the config-3 generator
    exp = oracle_lib.encode_bc6h(blocks, pyref.make_options(), False, rcp, threads=8)
    t = torch.from_numpy(blocks).cuda()
    out = gpu_ctx.encode_bc6h(t, api.Options()).cpu().numpy()
    assert (out == exp).all()
    for n in (8, 72):  # ragged tails of a 64-block wave
        assert (gpu_ctx.encode_bc6h(blocks[:n].copy(), api.Options()) == exp[:n]).all()


@pytest.mark.gpu
def test_gpu_vs_reference_on_this_box(gpu_ctx, ref_lib):
    from convectionkernels_amd import api
    gpu_ctx.set_rcp_table(ref_lib.probe_rcp())
    blocks = content.mixed_hdr_blocks(2024, 24)
    assert (gpu_ctx.encode_bc6h(blocks, api.Options()) == ref_lib.encode_bc6h(blocks, ref_lib.default_options(), False)).all()
    sblocks = content.mixed_hdr_blocks(2025, 8, signed=True)
    assert (gpu_ctx.encode_bc6h(sblocks, api.Options(), signed=True) == ref_lib.encode_bc6h(sblocks, ref_lib.default_options(), True)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("signed", [False, True])
def test_gpu_usable_rounds_vary_inside_a_wave(gpu_ctx, oracle_lib, signed):
    """The kernel computes a round's error only when some lane of the WAVE can commit with it (the delta-coding legality
    of the round's end points is known before its pixels are looked at) and skips subset 1 and the commit loop of a
    partition in which no round of subset 0 is usable.  Blocks whose deltas fit the transformed modes (ramps, narrow
    ranges, solid, two colours) are shuffled block by block among noise blocks whose deltas never do, so that inside
    every wave -- and inside every 8-block group, where the reference couples the lanes -- usable and unusable rounds mix
    at every precision."""
    from convectionkernels_amd import api
    rcp = oracle_lib.probe_rcp()
    gpu_ctx.set_rcp_table(rcp)
    mixed = content.mixed_hdr_blocks(4242, 40, signed=signed)
    noise = content.config_blocks_hdr(7, 64, 80)  # 320 blocks of config-3 noise
    blocks = np.concatenate([mixed, noise])
    rng = np.random.Generator(np.random.PCG64(11))
    blocks = np.ascontiguousarray(blocks[rng.permutation(len(blocks))])
    for opt_kw in ({}, {"flags": pyref.FLAG_BC6H_FAST_INDEXING, "seed_points": 3}, {"refine_bc6h": 1}):
        ob = pyref.make_options(**opt_kw)
        exp = oracle_lib.encode_bc6h(blocks, ob, signed, rcp, threads=8)
        got = gpu_ctx.encode_bc6h(blocks, api.Options.frombytes(ob), signed=signed)
        bad = np.nonzero((got != exp).any(axis=1))[0]
        assert bad.size == 0, (opt_kw, bad[:8])


@pytest.mark.gpu
def test_gpu_config3_full_size_hash(gpu_ctx):
    """BASELINE configs[2]: EncodeBC6HU on 4096x4096 random HDR (seed 3): SHA-256 of the 16 MiB output equals the
    reference's (canonical build, recorded RCPPS table)"""
    import hashlib
    import json
    import torch
    from convectionkernels_amd import api
    h = json.load(open(os.path.join(GOLD, "config_hashes.json")))
    if "config3_bc6hu_4096_seed3" not in h:
        pytest.skip("config 3 hash not generated")
    gpu_ctx.set_rcp_table(np.array(h["rcp_hex"], np.uint32).view(np.float32))
    t = torch.from_numpy(content.config_blocks_hdr(3, 4096, 4096)).cuda()
    out = gpu_ctx.encode_bc6h(t, api.Options()).cpu().numpy()
    assert hashlib.sha256(out.tobytes()).hexdigest() == h["config3_bc6hu_4096_seed3"]


def test_integer_endpoint_quantisation_equals_the_float_sequence():
    """csrc/bc6h_kernel.hip quantises endpoints with an integer ceil(elem * 64 / 31) (one v_mul_hi_u32) instead of the
    reference's rounded-up binary32 division (BC67.cpp:2425-2445): exhaustive equality over every value the colour-space
    clamp lets through, unsigned and signed (tools/check_bc6h_quantize.py; no GPU needed)."""
    import runpy
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    runpy.run_path(os.path.join(root, "tools", "check_bc6h_quantize.py"), run_name="__main__")
