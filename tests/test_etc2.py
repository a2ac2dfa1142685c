"""ETC2 RGB / RGBA / EAC alpha and ETC1: oracle vs golden vectors / reference on CPU; HIP path on GPU."""
import os

import numpy as np
import pytest

import content
from oracle import pyref

GOLD = os.path.join(os.path.dirname(__file__), "golden")
NAMES = ["default", "uniform", "weights"]
MODES = [(0, "rgb"), (1, "rgba"), (2, "alpha"), (3, "etc1")]  # 3 = EncodeETC1: individual + differential modes only


@pytest.mark.parametrize("name", NAMES)
def test_oracle_golden(oracle_lib, name):
    g = np.load(os.path.join(GOLD, "etc2_mixed.npz"))
    for mode, tag in MODES:
        out = oracle_lib.encode_etc2(g["blocks"], g["opt_" + name], mode, threads=8)
        bad = np.nonzero((out != g["out_%s_%s" % (tag, name)]).any(axis=1))[0]
        assert bad.size == 0, (tag, bad[:8])


PT_NAMES = ["default", "uniform_t025", "weights_t0", "t1"]


@pytest.mark.parametrize("name", PT_NAMES)
def test_oracle_golden_punchthrough(oracle_lib, name):
    g = np.load(os.path.join(GOLD, "etc2_mixed.npz"))
    out = oracle_lib.encode_etc2(g["pt_blocks"], g["pt_opt_" + name], 4, threads=8)
    bad = np.nonzero((out != g["pt_out_" + name]).any(axis=1))[0]
    assert bad.size == 0, bad[:8]


def test_punchthrough_goldens_cover_the_modes():
    """opaque and non-opaque blocks, and among the latter differential, T and H layouts"""
    g = np.load(os.path.join(GOLD, "etc2_mixed.npz"))
    out = g["pt_out_default"]
    opaque = (out[:, 3] >> 1) & 1
    assert (opaque == 1).any() and (opaque == 0).any()
    # non-opaque blocks: R / G overflow patterns select T and H mode (ETC.cpp:2414-2563)
    r = out[:, 0].astype(int)
    r_sum = (r >> 3) + ((((r & 7) ^ 4) - 4))
    assert ((opaque == 0) & ((r_sum < 0) | (r_sum > 31))).any()


def test_oracle_punchthrough_vs_reference(oracle_lib, ref_lib):
    blocks = content.punchthrough_blocks(23, 1)
    for opt in (pyref.make_options(), pyref.make_options(flags=pyref.FLAG_UNIFORM, threshold=0.75)):
        assert (oracle_lib.encode_etc2(blocks, opt, 4, 8) == ref_lib.encode_etc2(blocks, opt, 4)).all()


def test_oracle_known_answers(oracle_lib):
    g = np.load(os.path.join(GOLD, "known_answers.npz"))
    out = oracle_lib.encode_etc2(g["blocks"], pyref.make_options(), 1)
    assert (out == g["etc2rgba"]).all()
    assert out[0].tobytes().hex() == "93bac95c23f747d33107ad56df223d0a"
    assert out[7].tobytes().hex() == "ff1092492492492425f963c745d519f7"


def test_oracle_vs_reference(oracle_lib, ref_lib):
    blocks = np.concatenate([content.mixed_ldr_blocks(5150, 24), content.config_blocks(4, 32, 32)])
    for opt in (pyref.make_options(), pyref.make_options(flags=pyref.FLAG_UNIFORM)):
        for mode, _ in MODES:
            assert (oracle_lib.encode_etc2(blocks, opt, mode, 8) == ref_lib.encode_etc2(blocks, opt, mode)).all()


def _alloc_cases():
    """(encode options, AllocETC2Data options): the chroma axes follow the second (reference ETC.cpp:3117-3145)"""
    a = pyref.make_options(weights=(0.9, 0.3, 0.6, 1.0))
    b = pyref.make_options(weights=(0.1, 1.0, 0.8, 1.0), flags=pyref.FLAGS_DEFAULT | 0x400)  # + ETC_UseFakeBT709
    return [(pyref.make_options(), a), (a, pyref.make_options()), (pyref.make_options(flags=pyref.FLAG_UNIFORM), b)]


def test_oracle_alloc_time_axes_vs_reference(oracle_lib, ref_lib):
    """AllocETC2Data(options A) + EncodeETC2*(options B): the sector split uses A's axes, the error metric B's weights"""
    blocks = np.concatenate([content.mixed_ldr_blocks(77, 24), content.config_blocks(4, 32, 32)])
    pt = content.punchthrough_blocks(29, 1)
    differs = 0
    for enc, alloc in _alloc_cases():
        for mode in (0, 1, 4):
            b = pt if mode == 4 else blocks
            want = ref_lib.encode_etc2(b, enc, mode, alloc_options=alloc)
            assert (oracle_lib.encode_etc2(b, enc, mode, 8, alloc_options=alloc) == want).all(), mode
            differs += int((want != ref_lib.encode_etc2(b, enc, mode)).any())
    assert differs > 0  # the allocation-time options do matter on this content


FAKE_NAMES = ["fake709", "fake709_accurate", "fake709_uniform"]


@pytest.mark.parametrize("name", FAKE_NAMES)
def test_oracle_golden_fake_bt709(oracle_lib, name):
    """ETC_UseFakeBT709 (+ ETC_FakeBT709Accurate): every colour format, incl. punch-through"""
    g = np.load(os.path.join(GOLD, "etc2_mixed.npz"))
    for mode, tag in ((0, "rgb"), (1, "rgba"), (3, "etc1")):
        out = oracle_lib.encode_etc2(g["blocks"], g["opt_" + name], mode, threads=8)
        assert (out == g["out_%s_%s" % (tag, name)]).all(), tag
    out = oracle_lib.encode_etc2(g["pt_blocks"], g["opt_" + name], 4, threads=8)
    assert (out == g["pt_out_" + name]).all()


def test_fake_bt709_rounding_table_follows_the_rule():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_f709", os.path.join(root, "tools", "gen_fake709_rounding.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    for target, text in m.render().items():
        assert open(os.path.join(root, target)).read() == text, target


def test_t_mode_group_coupling(oracle_lib):
    """hazard H2: a block's T-mode candidates depend on the unique-colour count of its group
    (SURVEY App. B); encoding must therefore take whole groups"""
    blocks = content.mixed_ldr_blocks(8, 12)
    out = oracle_lib.encode_etc2(blocks, pyref.make_options(), 0, 4)
    assert out.shape == (96, 8)


# ------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", PT_NAMES)
def test_gpu_golden_punchthrough(gpu_ctx, name):
    from convectionkernels_amd import api
    g = np.load(os.path.join(GOLD, "etc2_mixed.npz"))
    out = gpu_ctx.encode_etc2_punchthrough_alpha(g["pt_blocks"], api.Options.frombytes(g["pt_opt_" + name]))
    bad = np.nonzero((out != g["pt_out_" + name]).any(axis=1))[0]
    assert bad.size == 0, bad[:8]


@pytest.mark.gpu
@pytest.mark.parametrize("name", FAKE_NAMES)
def test_gpu_golden_fake_bt709(gpu_ctx, name):
    from convectionkernels_amd import api
    g = np.load(os.path.join(GOLD, "etc2_mixed.npz"))
    opt = api.Options.frombytes(g["opt_" + name])
    for mode, tag in ((0, "rgb"), (1, "rgba"), (3, "etc1")):
        out = _enc(gpu_ctx, mode)(g["blocks"], opt)
        bad = np.nonzero((out != g["out_%s_%s" % (tag, name)]).any(axis=1))[0]
        assert bad.size == 0, (tag, bad[:8])
    out = gpu_ctx.encode_etc2_punchthrough_alpha(g["pt_blocks"], opt)
    bad = np.nonzero((out != g["pt_out_" + name]).any(axis=1))[0]
    assert bad.size == 0, bad[:8]


@pytest.mark.gpu
def test_gpu_fake_bt709_vs_oracle(gpu_ctx, oracle_lib):
    import torch
    from convectionkernels_amd import api
    blocks = np.concatenate([content.config_blocks(4, 128, 128), content.mixed_ldr_blocks(44, 32)])
    for flags in (api.Flags.Default | api.Flags.ETC_UseFakeBT709, api.Flags.Ultra | api.Flags.ETC_UseFakeBT709):
        exp = oracle_lib.encode_etc2(blocks, pyref.make_options(flags=flags), 0, threads=8)
        out = gpu_ctx.encode_etc2(torch.from_numpy(blocks).cuda(), api.Options(flags=flags)).cpu().numpy()
        bad = np.nonzero((out != exp).any(axis=1))[0]
        assert bad.size == 0, bad[:8]


@pytest.mark.gpu
def test_gpu_dark_content_zero_slot(gpu_ctx, oracle_lib):
    """dark blocks make the T mode's zero-slot candidate (hazard H2) win, which sends the kernel through its exact path
    (the group's unique-colour counts are only computed then)"""
    import torch
    from convectionkernels_amd import api
    blocks = content.dark_blocks(7, 2048)
    t = torch.from_numpy(blocks).cuda()
    for flags in (api.Flags.Default, api.Flags.Default | api.Flags.Uniform):
        opt = pyref.make_options(flags=flags)
        for mode, fn in ((0, gpu_ctx.encode_etc2), (4, gpu_ctx.encode_etc2_punchthrough_alpha)):
            exp = oracle_lib.encode_etc2(blocks, opt, mode, threads=8)
            out = fn(t, api.Options(flags=flags)).cpu().numpy()
            bad = np.nonzero((out != exp).any(axis=1))[0]
            assert bad.size == 0, (mode, bad[:8])


@pytest.mark.gpu
def test_gpu_punchthrough_vs_oracle(gpu_ctx, oracle_lib):
    """larger run on the device path: cut-out structures + config-4 noise (alpha random -> half the pixels transparent)"""
    import torch
    from convectionkernels_amd import api
    blocks = np.concatenate([content.punchthrough_blocks(29, 8), content.config_blocks(4, 256, 256)])
    for thr in (0.5, 0.02):
        exp = oracle_lib.encode_etc2(blocks, pyref.make_options(threshold=thr), 4, threads=8)
        out = gpu_ctx.encode_etc2_punchthrough_alpha(torch.from_numpy(blocks).cuda(), api.Options(threshold=thr)).cpu().numpy()
        bad = np.nonzero((out != exp).any(axis=1))[0]
        assert bad.size == 0, bad[:8]


def _enc(ctx, mode):
    return {0: ctx.encode_etc2, 1: ctx.encode_etc2_rgba, 2: ctx.encode_etc2_alpha, 3: ctx.encode_etc1}[mode]


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_golden(gpu_ctx, name):
    from convectionkernels_amd import api
    g = np.load(os.path.join(GOLD, "etc2_mixed.npz"))
    opt = api.Options.frombytes(g["opt_" + name])
    for mode, tag in MODES:
        out = _enc(gpu_ctx, mode)(g["blocks"], opt)
        bad = np.nonzero((out != g["out_%s_%s" % (tag, name)]).any(axis=1))[0]
        assert bad.size == 0, (tag, bad[:8])


@pytest.mark.gpu
def test_gpu_known_answers_and_config4(gpu_ctx, oracle_lib):
    """App. H vector; BASELINE configs[3] content (random RGBA, seed 4) vs the oracle; device path"""
    import torch
    from convectionkernels_amd import api
    g = np.load(os.path.join(GOLD, "known_answers.npz"))
    assert (gpu_ctx.encode_etc2_rgba(g["blocks"], api.Options()) == g["etc2rgba"]).all()
    blocks = content.config_blocks(4, 512, 512)  # 16384 blocks
    exp = oracle_lib.encode_etc2(blocks, pyref.make_options(), 1, threads=8)
    out = gpu_ctx.encode_etc2_rgba(torch.from_numpy(blocks).cuda(), api.Options()).cpu().numpy()
    bad = np.nonzero((out != exp).any(axis=1))[0]
    assert bad.size == 0, bad[:8]
    # ETC1 on the same content: both the individual (diff bit 0) and the differential mode must occur
    exp1 = oracle_lib.encode_etc2(blocks, pyref.make_options(), 3, threads=8)
    out1 = gpu_ctx.encode_etc1(torch.from_numpy(blocks).cuda(), api.Options()).cpu().numpy()
    assert (out1 == exp1).all()
    smooth = content.mixed_ldr_blocks(90, 64)
    exp2 = oracle_lib.encode_etc2(smooth, pyref.make_options(), 3, threads=8)
    assert (gpu_ctx.encode_etc1(smooth, api.Options()) == exp2).all()
    diff = np.concatenate([out1[:, 3], exp2[:, 3]]) & 2
    assert (diff == 0).any() and (diff != 0).any()
    smooth = content.mixed_ldr_blocks(99, 96)
    for mode, _ in MODES:
        exp = oracle_lib.encode_etc2(smooth, pyref.make_options(), mode, threads=8)
        assert (_enc(gpu_ctx, mode)(smooth, api.Options()) == exp).all()
    with pytest.raises(api.CvttError):
        gpu_ctx.encode_etc2(blocks[:12].copy(), api.Options())  # not a whole number of 8-block groups


@pytest.mark.gpu
def test_gpu_vs_reference_on_this_box(gpu_ctx, ref_lib):
    from convectionkernels_amd import api
    blocks = np.concatenate([content.mixed_ldr_blocks(777, 48), content.config_blocks(4, 64, 64)])
    for mode, _ in MODES:
        assert (_enc(gpu_ctx, mode)(blocks, api.Options()) == ref_lib.encode_etc2(blocks, ref_lib.default_options(), mode)).all()


@pytest.mark.gpu
def test_gpu_config4_full_size_hash(gpu_ctx):
    """BASELINE configs[3]: EncodeETC2RGBA on 4096x4096 random RGBA (seed 4): SHA-256 of the output equals the reference's"""
    import hashlib
    import json
    import torch
    from convectionkernels_amd import api
    h = json.load(open(os.path.join(GOLD, "config_hashes.json")))
    if "config4_etc2rgba_4096_seed4" not in h:
        pytest.skip("config 4 hash not generated")
    t = torch.from_numpy(content.config_blocks(4, 4096, 4096)).cuda()
    out = gpu_ctx.encode_etc2_rgba(t, api.Options()).cpu().numpy()
    assert hashlib.sha256(out.tobytes()).hexdigest() == h["config4_etc2rgba_4096_seed4"]


def test_oracle_eac11_golden(oracle_lib):
    g = np.load(os.path.join(GOLD, "etc2_mixed.npz"))
    assert (g["r11_blocks"] == content.mixed_r11_blocks(11, 64)).all()
    assert (oracle_lib.encode_eac11(g["r11_blocks"], False) == g["r11_unsigned"]).all()
    assert (oracle_lib.encode_eac11(g["r11_blocks"], True) == g["r11_signed"]).all()


def test_oracle_eac11_vs_reference(oracle_lib, ref_lib):
    b = content.mixed_r11_blocks(12, 256)
    for sg in (False, True):
        assert (oracle_lib.encode_eac11(b, sg) == ref_lib.encode_eac11(b, pyref.make_options(), sg)).all()


@pytest.mark.gpu
def test_gpu_eac11(gpu_ctx, oracle_lib):
    """EAC R11 unsigned / signed: golden (reference) and oracle on fresh content; host and device path"""
    import torch
    g = np.load(os.path.join(GOLD, "etc2_mixed.npz"))
    assert (gpu_ctx.encode_etc2_alpha11(g["r11_blocks"], signed=False) == g["r11_unsigned"]).all()
    assert (gpu_ctx.encode_etc2_alpha11(g["r11_blocks"], signed=True) == g["r11_signed"]).all()
    b = content.mixed_r11_blocks(77, 512)
    t = torch.from_numpy(b).cuda()
    for sg in (False, True):
        assert (gpu_ctx.encode_etc2_alpha11(t, signed=sg).cpu().numpy() == oracle_lib.encode_eac11(b, sg)).all()


@pytest.mark.gpu
def test_gpu_alloc_time_axes(gpu_ctx, oracle_lib):
    """the drop-in keeps the reference's split of responsibilities: axes from the Options of AllocETC2Data, weights from the
    Options of the Encode call (cvttmi_encode_etc2_with_data; ConvectionKernels_ETC.cpp:3117-3145)"""
    from convectionkernels_amd import api
    ref = pyref.RefLib() if pyref.RefLib.available() else None
    blocks = np.concatenate([content.mixed_ldr_blocks(77, 24), content.config_blocks(4, 32, 32)])
    pt = content.punchthrough_blocks(29, 1)
    for enc, alloc in _alloc_cases():
        data = api.AllocETC2Data(api.Options.frombytes(alloc))
        eo = api.Options.frombytes(enc)
        for mode, fn in ((0, gpu_ctx.encode_etc2), (1, gpu_ctx.encode_etc2_rgba), (4, gpu_ctx.encode_etc2_punchthrough_alpha)):
            b = pt if mode == 4 else blocks
            want = ref.encode_etc2(b, enc, mode, alloc_options=alloc) if ref else oracle_lib.encode_etc2(b, enc, mode, 8, alloc_options=alloc)
            got = fn(b, eo, compression_data=data)
            bad = np.nonzero((got != want).any(axis=1))[0]
            assert bad.size == 0, (mode, bad[:8])


def test_eac_magic_number_division_is_exact():
    """csrc/etc2_kernel.hip (EAC alpha / R11) divides a pixel's lookup value by the candidate's multiplier as
    (lookup * ceil(2^20 / multiplier)) >> 20 with a 24-bit multiply: exact for every lookup < 2^12 and multiplier <= 128
    (the ranges of ETC.cpp's EAC search), and the product stays below 2^32."""
    n = np.arange(4096, dtype=np.uint64)
    for d in range(1, 129):
        m = -(-(1 << 20) // d)
        assert m < (1 << 24)
        p = n * np.uint64(m)
        assert int(p.max()) < (1 << 32)
        assert ((p >> np.uint64(20)) == n // np.uint64(d)).all(), d


@pytest.mark.gpu
def test_gpu_launch_shapes_agree(gpu_ctx, oracle_lib):
    """The launch shape depends on the number of blocks -- the EAC search has a sixteen-lanes-per-block form for launches of
    at most 65 536 blocks and a lane-per-block form above, and the colour kernel deals the groups to the eight XCDs in eighths
    (ragged when the number of groups is no multiple of 8) -- the bytes must not: every size gives the prefix of one large
    encode, and the small ones equal the oracle."""
    import torch
    from convectionkernels_amd import api
    big = np.concatenate([content.config_blocks(4, 1024, 1024), content.mixed_ldr_blocks(5, 96), content.config_blocks(11, 64, 64)])
    big = np.ascontiguousarray(big[: 65536 + 264])  # above the EAC threshold, 8 233 groups: no multiple of 8
    t = torch.from_numpy(big).cuda()
    opt = api.Options()
    full_rgba = gpu_ctx.encode_etc2_rgba(t, opt).cpu().numpy()
    full_rgb = gpu_ctx.encode_etc2(t, opt).cpu().numpy()
    full_a = gpu_ctx.encode_etc2_alpha(t, opt).cpu().numpy()
    for n in (8, 24, 72, 1032, 65536, 65544):
        assert (gpu_ctx.encode_etc2_rgba(t[:n], opt).cpu().numpy() == full_rgba[:n]).all(), n
        assert (gpu_ctx.encode_etc2(t[:n], opt).cpu().numpy() == full_rgb[:n]).all(), n
        assert (gpu_ctx.encode_etc2_alpha(t[:n], opt).cpu().numpy() == full_a[:n]).all(), n
    small = big[:1032]
    assert (full_rgba[:1032] == oracle_lib.encode_etc2(small, pyref.make_options(), 1, threads=8)).all()
    # the same for the 11-bit EAC of EncodeETC2Alpha11
    r11 = content.mixed_r11_blocks(5, 8200)[: 65536 + 64]
    if r11.shape[0] > 65536:
        tr = torch.from_numpy(np.ascontiguousarray(r11)).cuda()
        for sg in (False, True):
            full = gpu_ctx.encode_etc2_alpha11(tr, signed=sg).cpu().numpy()
            assert (gpu_ctx.encode_etc2_alpha11(tr[:4096], signed=sg).cpu().numpy() == full[:4096]).all()
