"""BC2 / BC3 / BC4U / BC4S / BC5U / BC5S: oracle against golden vectors of the reference, HIP
kernels against goldens and oracle."""
import os

import numpy as np
import pytest

import content
from oracle import pyref

GOLD = os.path.join(os.path.dirname(__file__), "golden")
FORMATS = ((2, "bc2"), (3, "bc3"), (4, "bc4u"), (5, "bc4s"), (6, "bc5u"), (7, "bc5s"))
VARIANTS = ("default", "uniform_seeds2_refine1", "refine3_seeds3", "better", "exhaustive_plain")


@pytest.mark.parametrize("name", VARIANTS)
def test_oracle_golden(oracle_lib, name):
    g = np.load(os.path.join(GOLD, "s3tc_mixed.npz"))
    for fmt, tag in FORMATS:
        out = oracle_lib.encode_s3tc(g["blocks"], g["opt_" + name], fmt, g["rcp"], threads=4)
        bad = np.nonzero((out != g["out_%s_%s" % (tag, name)]).any(axis=1))[0]
        assert bad.size == 0, (tag, bad[:8])


def test_oracle_vs_reference_fresh(oracle_lib, ref_lib):
    blocks = np.concatenate([content.mixed_ldr_blocks(4711, 12), content.alpha_structure_blocks(9, 128)])
    rcp = ref_lib.probe_rcp()
    for fmt, tag in FORMATS:
        assert (oracle_lib.encode_s3tc(blocks, pyref.make_options(), fmt, rcp, threads=4) == ref_lib.encode_s3tc(blocks, pyref.make_options(), fmt)).all(), tag


def _gpu_encode(ctx, fmt, blocks, opt):
    if fmt == 2:
        return ctx.encode_bc2(blocks, opt)
    if fmt == 3:
        return ctx.encode_bc3(blocks, opt)
    if fmt in (4, 5):
        return ctx.encode_bc4(blocks, opt, signed=(fmt == 5))
    return ctx.encode_bc5(blocks, opt, signed=(fmt == 7))


@pytest.mark.gpu
@pytest.mark.parametrize("name", VARIANTS)
def test_gpu_golden(gpu_ctx, name):
    from convectionkernels_amd import api
    g = np.load(os.path.join(GOLD, "s3tc_mixed.npz"))
    gpu_ctx.set_rcp_table(g["rcp"])
    opt = api.Options.frombytes(g["opt_" + name])
    for fmt, tag in FORMATS:
        out = _gpu_encode(gpu_ctx, fmt, g["blocks"], opt)
        bad = np.nonzero((out != g["out_%s_%s" % (tag, name)]).any(axis=1))[0]
        assert bad.size == 0, (tag, bad[:8])


@pytest.mark.gpu
def test_gpu_vs_oracle_and_device_path(gpu_ctx, oracle_lib):
    import torch
    from convectionkernels_amd import api
    rcp = oracle_lib.probe_rcp()
    gpu_ctx.set_rcp_table(rcp)
    blocks = np.concatenate([content.config_blocks(1, 256, 256), content.alpha_structure_blocks(31, 2048), content.mixed_ldr_blocks(8, 48)])
    t = torch.from_numpy(blocks).cuda()
    for fmt, tag in FORMATS:
        exp = oracle_lib.encode_s3tc(blocks, pyref.make_options(), fmt, rcp, threads=8)
        out = _gpu_encode(gpu_ctx, fmt, t, api.Options()).cpu().numpy()
        bad = np.nonzero((out != exp).any(axis=1))[0]
        assert bad.size == 0, (tag, bad[:8])
    for fmt, tag in FORMATS[:2]:  # Flags::Better only changes the colour half of BC2 / BC3
        exp = oracle_lib.encode_s3tc(blocks[:4096], pyref.make_options(flags=pyref.FLAGS_BETTER), fmt, rcp, threads=8)
        out = _gpu_encode(gpu_ctx, fmt, t[:4096], api.Options(flags=api.Flags.Better)).cpu().numpy()
        assert (out == exp).all(), tag
