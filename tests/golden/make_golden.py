#!/usr/bin/env python3
"""Generate the committed golden vectors with the REAL reference (oracle/_ref, compiled from
/root/reference by oracle/Makefile with the canonical flags of SURVEY.md App. C).

Run in the build container only:   python tests/golden/make_golden.py
Outputs (data only -- inputs, expected outputs, the generating host's RCPPS table):
    tests/golden/bc7_mixed.npz      mixed-content groups x option/plan variants
    tests/golden/known_answers.npz  SURVEY.md App. H group (BC7/BC1/ETC2RGBA/BC6HU)
    tests/golden/config_hashes.json SHA-256 of whole-image outputs for the BASELINE configs
"""
import hashlib
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import content  # noqa: E402
from oracle import pyref  # noqa: E402


def bc7_variants(ref):
    P = pyref
    v = {
        "default": (P.make_options(), ref.default_plan()),
        "uniform": (P.make_options(flags=P.FLAGS_DEFAULT | P.FLAG_UNIFORM), ref.default_plan()),
        "punchthrough": (P.make_options(flags=P.FLAGS_DEFAULT | P.FLAG_BC7_RESPECT_PUNCHTHROUGH), ref.default_plan()),
        "better": (P.make_options(flags=P.FLAGS_BETTER), ref.default_plan()),
        # Flags::Ultra = BC7_TrySingleColor | S3TC_Paranoid | S3TC_Exhaustive | ETC_FakeBT709Accurate (slow indexing)
        "ultra": (P.make_options(flags=0x010 | 0x100 | 0x080 | 0x800), ref.default_plan()),
        "singlecolor": (P.make_options(flags=P.FLAGS_DEFAULT | P.FLAG_BC7_TRY_SINGLE_COLOR), ref.default_plan()),
        "refine1": (P.make_options(refine_bc7=1), ref.default_plan()),
        "refine3": (P.make_options(refine_bc7=3), ref.default_plan()),
        "weights": (P.make_options(weights=(0.5, 1.0, 0.25, 2.0)), ref.default_plan()),
    }
    for q in (1, 20, 60, 100):
        v["quality%d" % q] = (P.make_options(), ref.plan_from_quality(q))
    return v


def parallel_encode(fn, blocks, per_out, threads=8):
    n = blocks.shape[0]
    groups = n // 8
    cuts = [8 * (groups * i // threads) for i in range(threads + 1)]
    out = np.zeros((n, per_out), np.uint8)

    def work(i):
        a, b = cuts[i], cuts[i + 1]
        if b > a:
            out[a:b] = fn(blocks[a:b])

    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(work, range(threads)))
    return out


def main():
    ref = pyref.RefLib()
    fast = pyref.RefLib(fast=True)
    rcp = ref.probe_rcp()

    # ---- mixed content x variants ----
    blocks = content.mixed_ldr_blocks(20260929, 24)
    arrays = {"blocks": blocks, "rcp": rcp}
    names = []
    for name, (opt, plan) in bc7_variants(ref).items():
        out = ref.encode_bc7(blocks, opt, plan)
        assert (out == fast.encode_bc7(blocks, opt, plan)).all(), name  # BC7 is build-stable
        arrays["opt_" + name] = opt
        arrays["plan_" + name] = plan
        arrays["out_" + name] = out
        names.append(name)
    arrays["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "bc7_mixed.npz"), **arrays)

    # ---- BC7 plan configuration: every quality, and random fine-tuning parameter sets ----
    rngp = np.random.Generator(np.random.PCG64(4242))
    fts = rngp.integers(0, 5, (48, pyref.SIZEOF_BC7_FINETUNE)).astype(np.uint8)
    fts[::3][rngp.random((16, pyref.SIZEOF_BC7_FINETUNE)) < 0.7] = 0
    fts[0] = 0
    fts[1] = 4
    np.savez_compressed(os.path.join(HERE, "bc7_plans.npz"),
                        quality=np.stack([ref.plan_from_quality(q) for q in range(1, 101)]),
                        finetune=fts, finetune_plans=np.stack([ref.plan_from_finetune(f) for f in fts]))

    # ---- BC1: mixed content + config 1 image x option variants ----
    P = pyref
    bc1_blocks = np.concatenate([content.mixed_ldr_blocks(777, 24), content.config_blocks(1, 64, 64)])
    b1 = {"blocks": bc1_blocks, "rcp": rcp}
    b1names = []
    for name, o in {
        "default": P.make_options(),
        "plain": P.make_options(flags=P.FLAG_BC7_FAST_INDEXING),
        "uniform": P.make_options(flags=P.FLAGS_DEFAULT | P.FLAG_UNIFORM),
        "threshold09": P.make_options(threshold=0.9),
        "threshold0": P.make_options(threshold=0.0),
        "refine1_seeds2": P.make_options(refine_s3tc=1, seed_points=2),
        "refine3": P.make_options(refine_s3tc=3),
        "weights": P.make_options(weights=(0.5, 1.0, 0.25, 2.0)),
        # S3TC_Exhaustive: Flags::Better (with S3TC_Paranoid), and without the paranoid error metric
        "better": P.make_options(flags=P.FLAGS_BETTER),
        "exhaustive_plain": P.make_options(flags=P.FLAG_S3TC_EXHAUSTIVE | P.FLAG_UNIFORM),
        "exhaustive_weights": P.make_options(flags=P.FLAG_S3TC_EXHAUSTIVE, weights=(0.5, 1.0, 0.25, 2.0), threshold=0.3),
    }.items():
        b1["opt_" + name] = o
        b1["out_" + name] = ref.encode_bc1(bc1_blocks, o)
        b1names.append(name)
    b1["names"] = np.array(b1names)
    np.savez_compressed(os.path.join(HERE, "bc1_mixed.npz"), **b1)

    # ---- BC2 / BC3 / BC4 / BC5: mixed content + alpha structures x option variants ----
    sblocks = np.concatenate([content.mixed_ldr_blocks(2345, 24), content.alpha_structure_blocks(6, 256)])
    s3 = {"blocks": sblocks, "rcp": rcp}
    for name, o in {
        "default": P.make_options(),
        "uniform_seeds2_refine1": P.make_options(flags=P.FLAGS_DEFAULT | P.FLAG_UNIFORM, seed_points=2, refine_iic=1, refine_s3tc=1),
        "refine3_seeds3": P.make_options(refine_iic=3, seed_points=3),
        "better": P.make_options(flags=P.FLAGS_BETTER),
        "exhaustive_plain": P.make_options(flags=P.FLAG_S3TC_EXHAUSTIVE | P.FLAG_UNIFORM),
    }.items():
        s3["opt_" + name] = o
        for fmt, tag in ((2, "bc2"), (3, "bc3"), (4, "bc4u"), (5, "bc4s"), (6, "bc5u"), (7, "bc5s")):
            s3["out_%s_%s" % (tag, name)] = ref.encode_s3tc(sblocks, o, fmt)
    np.savez_compressed(os.path.join(HERE, "s3tc_mixed.npz"), **s3)

    # ---- BC6H: mixed HDR content x option variants (canonical -O1 build only: hazard H1) ----
    hdr = content.mixed_hdr_blocks(606, 16)
    hdrs = content.mixed_hdr_blocks(607, 8, signed=True)
    b6 = {"blocks": hdr, "blocks_signed": hdrs, "rcp": rcp}
    for name, o in {
        "default": P.make_options(),
        "fast": P.make_options(flags=P.FLAGS_DEFAULT | P.FLAG_BC6H_FAST_INDEXING),
        "uniform": P.make_options(flags=P.FLAGS_DEFAULT | P.FLAG_UNIFORM),
        "seeds2_refine2": P.make_options(seed_points=2, refine_bc6h=2),
        "weights": P.make_options(weights=(0.5, 1.0, 0.25, 2.0)),
    }.items():
        b6["opt_" + name] = o
        b6["out_" + name] = ref.encode_bc6h(hdr, o, False)
        b6["outs_" + name] = ref.encode_bc6h(hdrs, o, True)
    np.savez_compressed(os.path.join(HERE, "bc6h_mixed.npz"), **b6)

    # ---- ETC2 (canonical -O1 build): mixed content + config-4 image x option variants x modes ----
    etc_blocks = np.concatenate([content.mixed_ldr_blocks(404, 24), content.config_blocks(4, 64, 64)])
    e2 = {"blocks": etc_blocks}
    for name, o in {
        "default": P.make_options(),
        "uniform": P.make_options(flags=P.FLAGS_DEFAULT | P.FLAG_UNIFORM),
        "weights": P.make_options(weights=(0.5, 1.0, 0.25, 2.0)),
    }.items():
        e2["opt_" + name] = o
        for mode, tag in ((0, "rgb"), (1, "rgba"), (2, "alpha"), (3, "etc1")):
            e2["out_%s_%s" % (tag, name)] = ref.encode_etc2(etc_blocks, o, mode)
    # ETC_UseFakeBT709 (0x400), + ETC_FakeBT709Accurate (0x800), + Uniform (sector split only): colour formats
    fake_opts = {
        "fake709": P.make_options(flags=P.FLAGS_DEFAULT | 0x400),
        "fake709_accurate": P.make_options(flags=P.FLAGS_DEFAULT | 0x400 | 0x800),
        "fake709_uniform": P.make_options(flags=P.FLAGS_DEFAULT | 0x400 | P.FLAG_UNIFORM),
    }
    for name, o in fake_opts.items():
        e2["opt_" + name] = o
        for mode, tag in ((0, "rgb"), (1, "rgba"), (3, "etc1")):
            e2["out_%s_%s" % (tag, name)] = ref.encode_etc2(etc_blocks, o, mode)
    # punch-through alpha (EncodeETC2PunchthroughAlpha): cut-out structures x thresholds / metrics
    e2["pt_blocks"] = content.punchthrough_blocks(17, 2)
    for name, o in fake_opts.items():
        e2["pt_out_" + name] = ref.encode_etc2(e2["pt_blocks"], o, 4)
    for name, o in {
        "default": P.make_options(),
        "uniform_t025": P.make_options(flags=P.FLAGS_DEFAULT | P.FLAG_UNIFORM, threshold=0.25),
        "weights_t0": P.make_options(weights=(1.0, 0.3, 2.0, 1.0), threshold=0.0),
        "t1": P.make_options(threshold=1.0),
    }.items():
        e2["pt_opt_" + name] = o
        e2["pt_out_" + name] = ref.encode_etc2(e2["pt_blocks"], o, 4)
    # EAC R11 (EncodeETC2Alpha11), unsigned and signed, incl. out-of-range inputs (the reference clamps)
    e2["r11_blocks"] = content.mixed_r11_blocks(11, 64)
    e2["r11_unsigned"] = ref.encode_eac11(e2["r11_blocks"], P.make_options(), False)
    e2["r11_signed"] = ref.encode_eac11(e2["r11_blocks"], P.make_options(), True)
    np.savez_compressed(os.path.join(HERE, "etc2_mixed.npz"), **e2)

    # ---- known answers (App. H) ----
    ka = content.known_answer_group_ldr()
    opt = pyref.make_options()
    np.savez_compressed(os.path.join(HERE, "known_answers.npz"), blocks=ka, rcp=rcp,
                        bc7=ref.encode_bc7(ka, opt, ref.default_plan()),
                        bc1=ref.encode_bc1(ka, opt),
                        etc2rgba=ref.encode_etc2(ka, opt, 1),
                        hdr_blocks=content.known_answer_group_hdr(),
                        bc6hu=ref.encode_bc6h(content.known_answer_group_hdr(), opt, False))

    # ---- decoders: the reference's DecodeBC7 / DecodeBC6HU / DecodeBC6HS on encoder output and on random bytes
    # (every mode, reserved modes, arbitrary field values) ----
    rng = np.random.Generator(np.random.PCG64(777))
    g7 = np.load(os.path.join(HERE, "bc7_mixed.npz"))
    bc7_in = np.concatenate([g7["out_default"], g7["out_better"], g7["out_quality20"], rng.integers(0, 256, (2048, 16), dtype=np.uint8)])
    bc7_in[-8:, 0] = 0  # reserved mode
    g6 = np.load(os.path.join(HERE, "bc6h_mixed.npz"))
    rnd6 = rng.integers(0, 256, (2048, 16), dtype=np.uint8)
    bc6u_in = np.concatenate([g6["out_default"], g6["out_fast"], rnd6])
    bc6s_in = np.concatenate([g6["outs_default"], rnd6])
    np.savez_compressed(os.path.join(HERE, "decode.npz"), bc7_in=bc7_in, bc7_out=ref.decode_bc7(bc7_in),
                        bc6u_in=bc6u_in, bc6u_out=ref.decode_bc6h(bc6u_in, False),
                        bc6s_in=bc6s_in, bc6s_out=ref.decode_bc6h(bc6s_in, True))

    if "--skip-images" in sys.argv:
        return
    plan = ref.default_plan()
    if "--only-5b" in sys.argv:
        # config 5b (BC7 16384^2, seed 5, Flags::Ultra = slow indexing + BC7_TrySingleColor): ~15 CPU-minutes on 8 cores,
        # std::threads inside the shim; merged into the existing config_hashes.json
        path = os.path.join(HERE, "config_hashes.json")
        hashes = json.load(open(path))
        assert hashes["rcp_hex"] == [int(x) for x in rcp.view(np.uint32)]
        from convectionkernels_amd.api import Flags
        ultra = pyref.make_options(flags=Flags.Ultra)  # ConvectionKernels.h:68
        b = content.config_blocks(5, 16384, 16384)
        out, done, secs = fast.encode_mt("bc7", b, ultra, plan, threads=os.cpu_count() or 8, budget_s=1e9, chunk_blocks=512)
        assert done == b.shape[0]
        hashes["config5b_bc7_16384_seed5_ultra"] = hashlib.sha256(out.tobytes()).hexdigest()
        band = out.shape[0] // 4
        hashes["config5b_band_hashes"] = [hashlib.sha256(out[i * band:(i + 1) * band].tobytes()).hexdigest() for i in range(4)]
        np.save(os.path.join(HERE, "config5b_bc7_16384_seed5_ultra_head.npy"), out[:512])
        print("config5b", hashes["config5b_bc7_16384_seed5_ultra"], "%.0f s" % secs, flush=True)
        with open(path, "w") as f:
            json.dump(hashes, f, indent=1)
        return
    if "--only-big-images" in sys.argv:
        # configs 3, 4, 5 (BC6HU 4096^2, ETC2 RGBA 4096^2, BC7 16384^2): ~10 CPU-minutes on 8 cores;
        # merged into the existing config_hashes.json
        path = os.path.join(HERE, "config_hashes.json")
        hashes = json.load(open(path))
        assert hashes["rcp_hex"] == [int(x) for x in rcp.view(np.uint32)]
        b = content.config_blocks_hdr(3, 4096, 4096)
        out = parallel_encode(lambda x: ref.encode_bc6h(x, opt, False), b, 16)
        hashes["config3_bc6hu_4096_seed3"] = hashlib.sha256(out.tobytes()).hexdigest()
        print("config3", hashes["config3_bc6hu_4096_seed3"], flush=True)
        b = content.config_blocks(4, 4096, 4096)
        out = parallel_encode(lambda x: ref.encode_etc2(x, opt, 1), b, 16)
        hashes["config4_etc2rgba_4096_seed4"] = hashlib.sha256(out.tobytes()).hexdigest()
        print("config4", hashes["config4_etc2rgba_4096_seed4"], flush=True)
        b = content.config_blocks(5, 16384, 16384)
        out = parallel_encode(lambda x: fast.encode_bc7(x, opt, plan), b, 16)
        hashes["config5_bc7_16384_seed5"] = hashlib.sha256(out.tobytes()).hexdigest()
        # checksum of checksums: one SHA-256 per 4096-block-row band (4 bands of 4096 rows of pixels each)
        band = out.shape[0] // 4
        hashes["config5_band_hashes"] = [hashlib.sha256(out[i * band:(i + 1) * band].tobytes()).hexdigest() for i in range(4)]
        print("config5", hashes["config5_bc7_16384_seed5"], flush=True)
        with open(path, "w") as f:
            json.dump(hashes, f, indent=1)
        return
    # ---- whole-image hashes for the BASELINE.json configs ----
    hashes = {"rcp_hex": [int(x) for x in rcp.view(np.uint32)]}
    b = content.config_blocks(1, 256, 256)
    hashes["config1_bc1_256_seed1"] = hashlib.sha256(ref.encode_bc1(b, opt).tobytes()).hexdigest()
    for key, seed, opaque in (("config2_bc7_4096_seed2", 2, False), ("config2b_bc7_4096_seed2_opaque", 2, True)):
        b = content.config_blocks(seed, 4096, 4096, opaque=opaque)
        out = parallel_encode(lambda x: fast.encode_bc7(x, opt, plan), b, 16)
        hashes[key] = hashlib.sha256(out.tobytes()).hexdigest()
        # first 64 groups kept verbatim so a failing hash can be localised
        np.save(os.path.join(HERE, key + "_head.npy"), out[:512])
        print(key, hashes[key])
    with open(os.path.join(HERE, "config_hashes.json"), "w") as f:
        json.dump(hashes, f, indent=1)


if __name__ == "__main__":
    main()
