"""BC1 (EncodeBC1): oracle vs golden vectors / reference on CPU; HIP path vs all of them on GPU."""
import hashlib
import json
import os

import numpy as np
import pytest

import content
from oracle import pyref

GOLD = os.path.join(os.path.dirname(__file__), "golden")
NAMES = ["default", "plain", "uniform", "threshold09", "threshold0", "refine1_seeds2", "refine3", "weights",
         "better", "exhaustive_plain", "exhaustive_weights"]  # the last three: S3TC_Exhaustive


@pytest.mark.parametrize("name", NAMES)
def test_oracle_golden(oracle_lib, name):
    g = np.load(os.path.join(GOLD, "bc1_mixed.npz"))
    out = oracle_lib.encode_bc1(g["blocks"], g["opt_" + name], g["rcp"], threads=4)
    assert (out == g["out_" + name]).all()


def test_oracle_known_answers_and_config1_hash(oracle_lib):
    g = np.load(os.path.join(GOLD, "known_answers.npz"))
    out = oracle_lib.encode_bc1(g["blocks"], pyref.make_options(), g["rcp"])
    assert (out == g["bc1"]).all()
    assert out[0].tobytes().hex() == "706e5650bf422f2d"
    h = json.load(open(os.path.join(GOLD, "config_hashes.json")))
    rcp = np.array(h["rcp_hex"], np.uint32).view(np.float32)
    out = oracle_lib.encode_bc1(content.config_blocks(1, 256, 256), pyref.make_options(), rcp, threads=4)
    assert hashlib.sha256(out.tobytes()).hexdigest() == h["config1_bc1_256_seed1"]


def test_oracle_vs_reference(oracle_lib, ref_lib):
    blocks = content.mixed_ldr_blocks(4321, 36)
    for opt in (pyref.make_options(), pyref.make_options(flags=0), pyref.make_options(threshold=0.25, seed_points=3)):
        assert (oracle_lib.encode_bc1(blocks, opt, ref_lib.probe_rcp(), 4) == ref_lib.encode_bc1(blocks, opt)).all()


def test_oracle_exhaustive_vs_reference(oracle_lib, ref_lib):
    """S3TC_Exhaustive (Flags::Better): sorted-cluster enumeration + single-colour tables, incl. the group-coupled
    TestCounts escape (groups mixing transparent and opaque blocks under the alpha test)"""
    blocks = np.concatenate([content.mixed_ldr_blocks(1357, 10), content.alpha_structure_blocks(12, 96)])
    for opt in (pyref.make_options(flags=pyref.FLAGS_BETTER), pyref.make_options(flags=pyref.FLAG_S3TC_EXHAUSTIVE, threshold=0.7)):
        assert (oracle_lib.encode_bc1(blocks, opt, ref_lib.probe_rcp(), 4) == ref_lib.encode_bc1(blocks, opt)).all()


def test_single_colour_tables_follow_the_rule():
    """oracle/cvtt_oracle_s3tcsc.h and csrc/s3tc_sc_tables.h are what tools/gen_s3tc_single_color.py emits"""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_s3sc", os.path.join(root, "tools", "gen_s3tc_single_color.py"))
    sc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sc)
    for target, text in sc.render().items():
        assert open(os.path.join(root, target)).read() == text, target


# ------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_golden(gpu_ctx, name):
    from convectionkernels_amd import api
    g = np.load(os.path.join(GOLD, "bc1_mixed.npz"))
    gpu_ctx.set_rcp_table(g["rcp"])
    out = gpu_ctx.encode_bc1(g["blocks"], api.Options.frombytes(g["opt_" + name]))
    bad = np.nonzero((out != g["out_" + name]).any(axis=1))[0]
    assert bad.size == 0, bad[:8]


@pytest.mark.gpu
def test_gpu_config1_and_device_path(gpu_ctx, oracle_lib):
    """BASELINE configs[0]: EncodeBC1 on 256x256 random RGBA (seed 1), plus a 4096^2 run vs the oracle sample"""
    import torch
    from convectionkernels_amd import api
    h = json.load(open(os.path.join(GOLD, "config_hashes.json")))
    rcp = np.array(h["rcp_hex"], np.uint32).view(np.float32)
    gpu_ctx.set_rcp_table(rcp)
    blocks = content.config_blocks(1, 256, 256)
    out = gpu_ctx.encode_bc1(torch.from_numpy(blocks).cuda(), api.Options())
    torch.cuda.synchronize()
    assert hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest() == h["config1_bc1_256_seed1"]
    big = content.config_blocks(7, 2048, 2048)
    out = gpu_ctx.encode_bc1(torch.from_numpy(big).cuda(), api.Options()).cpu().numpy()
    exp = oracle_lib.encode_bc1(big, pyref.make_options(), rcp, threads=8)
    assert (out == exp).all()
    for n in (8, 72):  # ragged tails of a 64-block wave
        assert (gpu_ctx.encode_bc1(big[:n].copy(), api.Options()) == exp[:n]).all()


@pytest.mark.gpu
def test_gpu_exhaustive_vs_oracle(gpu_ctx, oracle_lib):
    """Flags::Better / S3TC_Exhaustive on the device: ~1100 end-point pairs per block"""
    import torch
    from convectionkernels_amd import api
    rcp = oracle_lib.probe_rcp()
    gpu_ctx.set_rcp_table(rcp)
    blocks = np.concatenate([content.config_blocks(3, 256, 256), content.alpha_structure_blocks(77, 1024), content.mixed_ldr_blocks(15, 24)])
    for flags, thr in ((pyref.FLAGS_BETTER, 0.5), (pyref.FLAG_S3TC_EXHAUSTIVE | pyref.FLAG_UNIFORM, 0.25)):
        exp = oracle_lib.encode_bc1(blocks, pyref.make_options(flags=flags, threshold=thr), rcp, threads=8)
        out = gpu_ctx.encode_bc1(torch.from_numpy(blocks).cuda(), api.Options(flags=flags, threshold=thr)).cpu().numpy()
        bad = np.nonzero((out != exp).any(axis=1))[0]
        assert bad.size == 0, bad[:8]
        assert (gpu_ctx.encode_bc1(blocks[:72].copy(), api.Options(flags=flags, threshold=thr)) == exp[:72]).all()


@pytest.mark.gpu
def test_gpu_vs_reference_on_this_box(gpu_ctx, ref_lib):
    from convectionkernels_amd import api
    gpu_ctx.set_rcp_table(ref_lib.probe_rcp())
    blocks = content.mixed_ldr_blocks(2468, 64)
    assert (gpu_ctx.encode_bc1(blocks, api.Options()) == ref_lib.encode_bc1(blocks, ref_lib.default_options())).all()
