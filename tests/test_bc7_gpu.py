"""GPU parity tests (-m gpu): the HIP BC7 path, called through the C ABI, against
(1) the committed golden vectors from the real reference, (2) the C oracle on seeded inputs,
(3) oracle/_ref on this box when it travelled, (4) whole-image hashes at BASELINE sizes."""
import hashlib
import json
import os

import numpy as np
import pytest

import content
from oracle import pyref

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")

GPU_VARIANTS = ["default", "uniform", "punchthrough", "better", "ultra", "singlecolor", "refine1", "refine3", "weights",
                "quality1", "quality20", "quality60", "quality100"]


def _api():
    from convectionkernels_amd import api
    return api


def _diff(a, b):
    return np.nonzero((np.asarray(a) != np.asarray(b)).any(axis=1))[0]


def test_known_answers(gpu_ctx):
    api = _api()
    g = np.load(os.path.join(GOLD, "known_answers.npz"))
    gpu_ctx.set_rcp_table(g["rcp"])
    out = gpu_ctx.encode_bc7(g["blocks"], api.Options(), api.BC7EncodingPlan())
    assert _diff(out, g["bc7"]).size == 0
    assert out[0].tobytes().hex() == "108a856ce10f2c7dcac90b5c0a5d2acf"


@pytest.mark.parametrize("name", GPU_VARIANTS)
def test_golden_mixed(gpu_ctx, name):
    api = _api()
    g = np.load(os.path.join(GOLD, "bc7_mixed.npz"))
    gpu_ctx.set_rcp_table(g["rcp"])
    out = gpu_ctx.encode_bc7(g["blocks"], api.Options.frombytes(g["opt_" + name]),
                             api.BC7EncodingPlan.frombytes(g["plan_" + name]))
    bad = _diff(out, g["out_" + name])
    assert bad.size == 0, "blocks %s differ" % bad[:8]


def test_vs_oracle_seeded(gpu_ctx, oracle_lib):
    """fresh seeded content, host RCPPS table of THIS box on both sides"""
    api = _api()
    rcp = oracle_lib.probe_rcp()
    gpu_ctx.set_rcp_table(rcp)
    blocks = content.mixed_ldr_blocks(4242, 24)
    for opt in (api.Options(), api.Options(flags=api.Flags.Better), api.Options(flags=api.Flags.Default | api.Flags.Uniform)):
        exp = oracle_lib.encode_bc7(blocks, np.frombuffer(opt.tobytes(), np.uint8).copy(),
                                    np.frombuffer(api.BC7EncodingPlan().tobytes(), np.uint8).copy(), rcp, threads=8)
        out = gpu_ctx.encode_bc7(blocks, opt, api.BC7EncodingPlan())
        bad = _diff(out, exp)
        assert bad.size == 0, "flags %x blocks %s" % (opt.flags, bad[:8])


def test_vs_reference_on_this_box(gpu_ctx, ref_lib):
    """bit-exact vs the reference CPU path on the same box (its own RCPPS)"""
    api = _api()
    gpu_ctx.set_rcp_table(ref_lib.probe_rcp())
    blocks = content.config_blocks(2, 256, 256)
    blocks = np.concatenate([blocks, content.config_blocks(2, 128, 128, opaque=True)])
    exp = ref_lib.encode_bc7(blocks, ref_lib.default_options(), ref_lib.default_plan())
    out = gpu_ctx.encode_bc7(blocks, api.Options(), api.BC7EncodingPlan())
    assert _diff(out, exp).size == 0


def test_pruned_equals_exhaustive(gpu_ctx):
    """the branch-and-bound search (default) and the exhaustive search give identical blocks"""
    import torch
    api = _api()
    blocks = np.concatenate([content.mixed_ldr_blocks(31337, 48), content.config_blocks(9, 128, 128),
                             content.config_blocks(10, 128, 128, opaque=True)])
    t = torch.from_numpy(blocks).cuda()
    try:
        for opt in (api.Options(), api.Options(flags=api.Flags.Better), api.Options(flags=api.Flags.Default | api.Flags.Uniform),
                    api.Options(redWeight=3.0, greenWeight=0.25, blueWeight=1.5, alphaWeight=0.1)):
            gpu_ctx.set_exhaustive(False)
            a = gpu_ctx.encode_bc7(t, opt).cpu().numpy()
            gpu_ctx.set_exhaustive(True)
            b = gpu_ctx.encode_bc7(t, opt).cpu().numpy()
            assert _diff(a, b).size == 0
    finally:
        gpu_ctx.set_exhaustive(False)


def test_bounds_adversarial_content(gpu_ctx, oracle_lib):
    """Soundness of the branch-and-bound on content built to sit on its edges (content.adversarial_bound_blocks:
    rank-one scatter matrices, residuals of the order of the rounding allowance, two-colour and two-line blocks, +-1 LSB
    noise, every alpha variant): pruned == exhaustive on 2^20 blocks with the default options and on 2^17 blocks for slow
    indexing, the uniform metric, skewed weights and the single-colour flag; == the CPU oracle on the first 2^15."""
    import torch
    api = _api()
    rcp = oracle_lib.probe_rcp()
    gpu_ctx.set_rcp_table(rcp)
    blocks = content.adversarial_bound_blocks(20260929, 1 << 20)
    t = torch.from_numpy(blocks).cuda()
    plan = api.BC7EncodingPlan()
    try:
        cases = [(api.Options(), 1 << 20), (api.Options(flags=api.Flags.Better), 1 << 17), (api.Options(flags=api.Flags.Default | api.Flags.Uniform), 1 << 17),
                 (api.Options(redWeight=3.0, greenWeight=0.25, blueWeight=1.5, alphaWeight=0.1), 1 << 17), (api.Options(flags=api.Flags.Ultra), 1 << 16)]
        for i, (opt, n) in enumerate(cases):
            gpu_ctx.set_exhaustive(False)
            a = gpu_ctx.encode_bc7(t[:n], opt, plan).cpu().numpy()
            gpu_ctx.set_exhaustive(True)
            b = gpu_ctx.encode_bc7(t[:n], opt, plan).cpu().numpy()
            bad = _diff(a, b)
            assert bad.size == 0, "case %d: pruned != exhaustive at blocks %s" % (i, bad[:8])
            if i == 0:
                m = 1 << 15
                exp = oracle_lib.encode_bc7(blocks[:m], np.frombuffer(opt.tobytes(), np.uint8).copy(),
                                            np.frombuffer(plan.tobytes(), np.uint8).copy(), rcp, threads=16)
                bad = _diff(a[:m], exp)
                assert bad.size == 0, "pruned != oracle at blocks %s" % bad[:8]
    finally:
        gpu_ctx.set_exhaustive(False)


def test_device_tensor_path_and_ragged_sizes(gpu_ctx, oracle_lib):
    import torch
    api = _api()
    rcp = oracle_lib.probe_rcp()
    gpu_ctx.set_rcp_table(rcp)
    blocks = content.mixed_ldr_blocks(77, 5)  # 40 blocks: 2.5 waves
    exp = oracle_lib.encode_bc7(blocks, pyref.make_options(), np.frombuffer(api.BC7EncodingPlan().tobytes(), np.uint8).copy(), rcp, threads=8)
    for n in (8, 16, 24, 40):
        t = torch.from_numpy(blocks[:n].copy()).cuda()
        out = gpu_ctx.encode_bc7(t, api.Options(), api.BC7EncodingPlan())
        torch.cuda.synchronize()
        assert _diff(out.cpu().numpy(), exp[:n]).size == 0
    # empty input is a no-op, non-multiple of 8 is rejected
    assert gpu_ctx.encode_bc7(np.zeros((0, 16, 4), np.uint8)).shape[0] == 0
    with pytest.raises(api.CvttError):
        gpu_ctx.encode_bc7(np.zeros((4, 16, 4), np.uint8))


def test_respect_punchthrough_many_refine_rounds(gpu_ctx, oracle_lib):
    """BC7_RespectPunchThrough with more refine rounds than the LDS trial table holds (2, kMaxPTRefine): the reference clamps
    refineRoundsBC7 only from below (BC67.cpp:1044-1045), so 7, 9 and 12 rounds must work too -- the table moves to HBM.  (One
    launch here; test_respect_punchthrough_chunked_launches splits such a call.)"""
    api = _api()
    rcp = oracle_lib.probe_rcp()
    gpu_ctx.set_rcp_table(rcp)
    blocks = np.concatenate([content.mixed_ldr_blocks(4242, 40), content.mixed_ldr_blocks(9, 24)[::-1]])
    plan = api.BC7EncodingPlan()
    PTF = api.Flags.BC7_RespectPunchThrough
    for opt in (api.Options(flags=api.Flags.Default | PTF, refineRoundsBC7=7), api.Options(flags=api.Flags.Better | PTF, refineRoundsBC7=9),
                api.Options(flags=api.Flags.Default | PTF | api.Flags.Uniform, refineRoundsBC7=12)):
        exp = oracle_lib.encode_bc7(blocks, np.frombuffer(opt.tobytes(), np.uint8).copy(),
                                    np.frombuffer(plan.tobytes(), np.uint8).copy(), rcp, threads=8)
        out = gpu_ctx.encode_bc7(blocks, opt, plan)
        bad = _diff(out, exp)
        assert bad.size == 0, "flags %x refine %d blocks %s" % (opt.flags, opt.refineRoundsBC7, bad[:8])
    # back to a table that fits LDS on the same context
    opt = api.Options(flags=api.Flags.Default | PTF)
    exp = oracle_lib.encode_bc7(blocks, np.frombuffer(opt.tobytes(), np.uint8).copy(), np.frombuffer(plan.tobytes(), np.uint8).copy(), rcp, threads=8)
    assert _diff(gpu_ctx.encode_bc7(blocks, opt, plan), exp).size == 0


def test_respect_punchthrough_chunked_launches(oracle_lib, monkeypatch):
    """The same path when the HBM trial table does not hold the whole call: the shim sizes its launches for 256 MB of table;
    the developer knob CVTTMI_BC7_PT_TABLE_KB (read when a context is created) makes 504 blocks = 31.5 waves go in four
    launches of 10 + 10 + 10 + 1.5 waves -- per-launch input / output offsets, a ragged last chunk that is not a multiple of
    16 blocks, and the table reused from launch to launch."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    api = _api()
    monkeypatch.setenv("CVTTMI_BC7_PT_TABLE_KB", "140")  # 7 rounds: 14 KB per wave -> 10 waves per launch
    ctx = api.Context(0)
    monkeypatch.delenv("CVTTMI_BC7_PT_TABLE_KB")
    rcp = oracle_lib.probe_rcp()
    ctx.set_rcp_table(rcp)
    blocks = np.concatenate([content.mixed_ldr_blocks(4242, 40), content.mixed_ldr_blocks(9, 24)[::-1]])[:504]
    plan = api.BC7EncodingPlan()
    PTF = api.Flags.BC7_RespectPunchThrough
    for opt in (api.Options(flags=api.Flags.Default | PTF, refineRoundsBC7=7), api.Options(flags=api.Flags.Better | PTF, refineRoundsBC7=3)):
        exp = oracle_lib.encode_bc7(blocks, np.frombuffer(opt.tobytes(), np.uint8).copy(),
                                    np.frombuffer(plan.tobytes(), np.uint8).copy(), rcp, threads=8)
        out = ctx.encode_bc7(blocks, opt, plan)
        bad = _diff(out, exp)
        assert bad.size == 0, "flags %x refine %d blocks %s" % (opt.flags, opt.refineRoundsBC7, bad[:8])


def test_respect_punchthrough(gpu_ctx, oracle_lib):
    """BC7_RespectPunchThrough: the reference's commit rule couples the 8 blocks of a group per trial in modes 6/7
    (and commits NOT-better results of invalid lanes, ParallelMath.h:900-905); mixed content with binary alpha,
    opaque and translucent blocks in the same groups, fast / slow indexing, with the single-colour flag, 1-5 refine rounds"""
    api = _api()
    rcp = oracle_lib.probe_rcp()
    gpu_ctx.set_rcp_table(rcp)
    blocks = np.concatenate([content.mixed_ldr_blocks(31337, 36), content.mixed_ldr_blocks(6, 12)[::-1]])
    plan = api.BC7EncodingPlan()
    PTF = api.Flags.BC7_RespectPunchThrough
    for opt in (api.Options(flags=api.Flags.Default | PTF), api.Options(flags=api.Flags.Better | PTF),
                api.Options(flags=api.Flags.Ultra | PTF), api.Options(flags=api.Flags.Default | PTF | api.Flags.Uniform, refineRoundsBC7=1),
                api.Options(flags=api.Flags.Default | PTF, refineRoundsBC7=3), api.Options(flags=api.Flags.Default | PTF, refineRoundsBC7=5)):
        exp = oracle_lib.encode_bc7(blocks, np.frombuffer(opt.tobytes(), np.uint8).copy(),
                                    np.frombuffer(plan.tobytes(), np.uint8).copy(), rcp, threads=8)
        for exhaustive in (False, True):
            gpu_ctx.set_exhaustive(exhaustive)
            out = gpu_ctx.encode_bc7(blocks, opt, plan)
            gpu_ctx.set_exhaustive(False)
            bad = _diff(out, exp)
            assert bad.size == 0, "flags %x refine %d exhaustive %s blocks %s" % (opt.flags, opt.refineRoundsBC7, exhaustive, bad[:8])


def test_single_colour_flag_on_dark_content(gpu_ctx, oracle_lib):
    """BC7_TrySingleColor / Flags::Ultra (BASELINE config 5b) on content where the fixed
    (0,0,0[,255]) candidate is competitive: near-black noise, with and without alpha"""
    api = _api()
    rcp = oracle_lib.probe_rcp()
    gpu_ctx.set_rcp_table(rcp)
    rng = np.random.Generator(np.random.PCG64(5))
    blocks = np.zeros((256, 16, 4), np.uint8)
    for b in range(256):
        blocks[b, :, :3] = rng.integers(0, (1, 2, 3, 4, 6, 8, 12, 20)[b % 8] + 1, (16, 3))
        blocks[b, :, 3] = 255 if (b // 8) % 3 == 0 else (rng.integers(250, 256, 16) if (b // 8) % 3 == 1 else rng.choice(np.array([0, 255], np.uint8), 16))
        if b % 5 == 0:
            blocks[b, rng.integers(0, 16), :3] = rng.integers(0, 256, 3)
    plan = api.BC7EncodingPlan()
    for flags in (api.Flags.Default | api.Flags.BC7_TrySingleColor, api.Flags.Ultra, api.Flags.Ultra | api.Flags.Uniform):
        opt = api.Options(flags=flags)
        exp = oracle_lib.encode_bc7(blocks, np.frombuffer(opt.tobytes(), np.uint8).copy(),
                                    np.frombuffer(plan.tobytes(), np.uint8).copy(), rcp, threads=8)
        out = gpu_ctx.encode_bc7(blocks, opt, plan)
        bad = _diff(out, exp)
        assert bad.size == 0, "flags %x blocks %s" % (flags, bad[:8])


def test_second_launch_hand_over_is_invisible(gpu_ctx, oracle_lib, monkeypatch):
    """Blocks with many live mode-7 partitions are finished by a second launch (bc7_kernel.hip, HARD).  Off, with three
    slots only (most such blocks find none and search on themselves), with a slot for every block and with the defaults
    (off below half a million blocks; the full-size hash tests run with it on), the output is the oracle's."""
    api = _api()
    rcp = oracle_lib.probe_rcp()
    blocks = np.concatenate([content.mixed_ldr_blocks(777, 40), content.config_blocks(9, 256, 256)])
    opt, plan = api.Options(), api.BC7EncodingPlan()
    exp = oracle_lib.encode_bc7(blocks, np.frombuffer(opt.tobytes(), np.uint8).copy(),
                                np.frombuffer(plan.tobytes(), np.uint8).copy(), rcp, threads=8)
    for env in ({"CVTTMI_BC7_HARD_MIN": "0"}, {"CVTTMI_BC7_HARD_CAP": "3", "CVTTMI_BC7_HARD_MIN": "2"},
                {"CVTTMI_BC7_HARD_CAP": "4096", "CVTTMI_BC7_HARD_MIN": "2", "CVTTMI_BC7_HARD_DIV": "1000000"}, {}):
        for k in ("CVTTMI_BC7_HARD_MIN", "CVTTMI_BC7_HARD_CAP", "CVTTMI_BC7_HARD_DIV"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ctx = api.Context(0)  # the settings are read when the context is created
        ctx.set_rcp_table(rcp)
        bad = _diff(ctx.encode_bc7(blocks, opt, plan), exp)
        assert bad.size == 0, "%s: blocks %s differ" % (env, bad[:8])


def test_config2_full_size_hash(gpu_ctx):
    """BASELINE config 2: 4096x4096 random RGBA, seed 2 (1,048,576 blocks) -- SHA-256 of the
    whole output equals the reference's (generated with the recorded RCPPS table)."""
    import torch
    api = _api()
    h = json.load(open(os.path.join(GOLD, "config_hashes.json")))
    gpu_ctx.set_rcp_table(np.array(h["rcp_hex"], np.uint32).view(np.float32))
    for key, opaque in (("config2_bc7_4096_seed2", False), ("config2b_bc7_4096_seed2_opaque", True)):
        blocks = content.config_blocks(2, 4096, 4096, opaque=opaque)
        t = torch.from_numpy(blocks).cuda()
        out = gpu_ctx.encode_bc7(t, api.Options(), api.BC7EncodingPlan())
        torch.cuda.synchronize()
        out = out.cpu().numpy()
        head = np.load(os.path.join(GOLD, key + "_head.npy"))
        assert _diff(out[:512], head).size == 0
        assert hashlib.sha256(out.tobytes()).hexdigest() == h[key]


def test_config5_full_size_hash(gpu_ctx):
    """BASELINE config 5a: 16384x16384 random RGBA, seed 5 (16,777,216 blocks, 1 GiB in, 256 MiB out), through the
    device-side tiling of the linear image; a checksum per quarter of the image and the SHA-256 of the whole
    output equal the reference's."""
    import torch
    from convectionkernels_amd import synth
    api = _api()
    h = json.load(open(os.path.join(GOLD, "config_hashes.json")))
    if "config5_bc7_16384_seed5" not in h:
        pytest.skip("config 5 hash not generated")
    gpu_ctx.set_rcp_table(np.array(h["rcp_hex"], np.uint32).view(np.float32))
    img = torch.from_numpy(synth.image_rgba8(5, 16384, 16384)).cuda()
    out = gpu_ctx.encode_image("bc7", img, api.Options(), api.BC7EncodingPlan())
    torch.cuda.synchronize()
    del img
    out = out.cpu().numpy()
    assert out.shape == (16777216, 16)
    band = out.shape[0] // 4
    for i in range(4):
        assert hashlib.sha256(out[i * band:(i + 1) * band].tobytes()).hexdigest() == h["config5_band_hashes"][i], "band %d" % i
    assert hashlib.sha256(out.tobytes()).hexdigest() == h["config5_bc7_16384_seed5"]
    # BASELINE config 5b: the same image with Flags::Ultra (slow indexing + BC7_TrySingleColor, ConvectionKernels.h:68)
    if "config5b_bc7_16384_seed5_ultra" in h:
        img = torch.from_numpy(synth.image_rgba8(5, 16384, 16384)).cuda()
        out = gpu_ctx.encode_image("bc7", img, api.Options(flags=api.Flags.Ultra), api.BC7EncodingPlan())
        torch.cuda.synchronize()
        del img
        out = out.cpu().numpy()
        assert _diff(out[:512], np.load(os.path.join(GOLD, "config5b_bc7_16384_seed5_ultra_head.npy"))).size == 0
        for i in range(4):
            assert hashlib.sha256(out[i * band:(i + 1) * band].tobytes()).hexdigest() == h["config5b_band_hashes"][i], "5b band %d" % i
        assert hashlib.sha256(out.tobytes()).hexdigest() == h["config5b_bc7_16384_seed5_ultra"]


def test_arithmetic_contract(gpu_ctx):
    """the device's binary32 divide and square root are correctly rounded (DIVPS / SQRTPS of the reference's lanes):
    4M pseudo-random finite operand patterns against the host"""
    bad_div, bad_sqrt = gpu_ctx.selftest_arith(1 << 22, 20260929)
    assert (bad_div, bad_sqrt) == (0, 0)


def test_regression_sqrt_rounding_boundary(gpu_ctx, oracle_lib):
    """group 85482 of BASELINE config 5: the PCA axis length of one block sits where a 1-ulp square root flips a 5-bit
    endpoint (found by the 16384^2 hash); needs the exactly rounded sqrt"""
    api = _api()
    h = json.load(open(os.path.join(GOLD, "config_hashes.json")))
    rcp = np.array(h["rcp_hex"], np.uint32).view(np.float32)
    gpu_ctx.set_rcp_table(rcp)
    g = np.load(os.path.join(GOLD, "regress_bc7_group_683856.npy"))
    exp = oracle_lib.encode_bc7(g, pyref.make_options(), np.frombuffer(api.BC7EncodingPlan().tobytes(), np.uint8).copy(), rcp)
    assert exp[2].tobytes().hex() == "d0f2f4e15a3ea851efaa08c23109ead3"
    assert _diff(gpu_ctx.encode_bc7(g), exp).size == 0
