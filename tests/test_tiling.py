"""Image -> PixelBlock tiling on the device against a numpy statement of the
reference caller's loop (etc2packer/etc2packer.cpp:215-247, 275-281)."""
import numpy as np
import pytest

import content
from convectionkernels_amd import synth


def test_numpy_tiling_statement():
    """the test-side statement itself: multiples of 32 agree with the plain reshape, ragged sizes clamp"""
    img = synth.image_rgba8(7, 64, 32)
    assert (content.tile_clamped(img) == synth.tile_blocks(img)).all()
    img = synth.image_rgba8(8, 40, 12)[:10, :37]
    t = content.tile_clamped(img)
    assert t.shape == (3 * 16, 16, 4)
    assert (t[0, 5] == img[1, 1]).all()
    assert (t[9, 3] == img[0, 36]).all() and (t[9, 0] == img[0, 36]).all()      # block 9 starts at x=36: columns clamp to 36
    assert (t[15, 0] == img[0, 36]).all()                                         # a pure padding block repeats the last column
    assert (t[2 * 16 + 1, 15] == img[9, 7]).all()                                 # rows clamp to y=9
    assert content.compact_rows(np.arange(48 * 2).reshape(48, 2), 37, 10).shape == (30, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("w,h", [(64, 32), (37, 10), (1, 1), (33, 5), (256, 4), (100, 100)])
def test_tile_rgba8(gpu_ctx, w, h):
    import torch
    img = synth.image_rgba8(11, (w + 3) // 4 * 4, (h + 3) // 4 * 4)[:h, :w].copy()
    got = gpu_ctx.tile_image(torch.from_numpy(img).cuda()).cpu().numpy()
    assert (got == content.tile_clamped(img)).all()


@pytest.mark.gpu
def test_tile_strided_rows_and_f16(gpu_ctx):
    import torch
    big = torch.from_numpy(synth.image_rgba8(12, 128, 64)).cuda()
    view = big[3:50, 5:90]                      # strided rows, unaligned start
    got = gpu_ctx.tile_image(view).cpu().numpy()
    assert (got == content.tile_clamped(view.cpu().numpy())).all()
    hdr = synth.image_f16bits(3, 64, 48)[:45, :50].copy()
    got = gpu_ctx.tile_image(torch.from_numpy(hdr).cuda()).cpu().numpy()
    assert (got == content.tile_clamped(hdr)).all()


@pytest.mark.gpu
def test_encode_image_matches_blocks_path(gpu_ctx, oracle_lib):
    """image in HBM -> packed rows == oracle on the numpy-tiled blocks, ragged size (groups with padding blocks)"""
    import torch
    from oracle import pyref
    from convectionkernels_amd import api
    rcp = oracle_lib.probe_rcp()
    gpu_ctx.set_rcp_table(rcp)
    w, h = 75, 22
    img = synth.image_rgba8(13, 76, 24)[:h, :w].copy()
    blocks = content.tile_clamped(img)
    d_img = torch.from_numpy(img).cuda()
    plan = np.frombuffer(api.BC7EncodingPlan().tobytes(), np.uint8).copy()
    exp = content.compact_rows(oracle_lib.encode_bc7(blocks, pyref.make_options(), plan, rcp, threads=4), w, h)
    got = gpu_ctx.encode_image("bc7", d_img).cpu().numpy()
    assert got.shape == exp.shape and (got == exp).all()
    exp = content.compact_rows(oracle_lib.encode_bc1(blocks, pyref.make_options(), rcp, threads=4), w, h)
    got = gpu_ctx.encode_image("bc1", d_img).cpu().numpy()
    assert (got == exp).all()
    exp = content.compact_rows(oracle_lib.encode_etc2(blocks, pyref.make_options(), 1, threads=4), w, h)
    got = gpu_ctx.encode_image("etc2rgba", d_img).cpu().numpy()
    assert (got == exp).all()
