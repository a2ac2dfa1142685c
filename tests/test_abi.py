"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/cvtt_mi355x.h declares, and its PODs have the reference's layout.  No compute calls."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "cvtt_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cvttmi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from convectionkernels_amd import api
    lib = api.load_library()
    names = _declared_symbols()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), n
    assert set(api.exported_symbols()) <= set(names)


def test_pod_layouts():
    from convectionkernels_amd import api
    lib = api.load_library()
    assert ctypes.sizeof(api.Options) == 44
    assert ctypes.sizeof(api.BC7EncodingPlan) == 808
    o = api.Options()
    o2 = api.Options()
    ctypes.memset(ctypes.addressof(o2), 0xAA, 44)
    lib.cvttmi_default_options(ctypes.byref(o2))
    assert bytes(o) == bytes(o2)
    p = api.BC7EncodingPlan()
    p2 = api.BC7EncodingPlan()
    lib.cvttmi_default_bc7_plan(ctypes.byref(p2))
    assert bytes(p) == bytes(p2)
    assert api.BC7EncodingPlan.mode7RGBAPartitionEnabled.offset == 32
    assert api.BC7EncodingPlan.seedPointsForShapeRGB.offset == 61
    assert api.BC7EncodingPlan.rgbShapeList.offset == 563


def test_no_device_fails_loudly():
    import torch
    from convectionkernels_amd import api
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(api.CvttError):
        api.Context(0)
    with pytest.raises(api.CvttError):
        api.EncodeBC7(np.zeros((8, 16, 4), np.uint8))


def test_product_never_touches_the_oracle():
    """the shipped package must not import / link / load anything under oracle/"""
    import re
    bad = re.compile(r"(from|import)\s+oracle|oracle/|cvtt_oracle|libcvtt_ref|pyref")
    pkg = os.path.join(ROOT, "convectionkernels_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not bad.search(text), os.path.join(dirpath, f)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "include")):
        for hdr in files:
            assert not bad.search(open(os.path.join(dirpath, hdr)).read()), hdr


CXX_CLIENT = r"""
// a client written against the reference's C++ header, built against ours
#include "cvtt/ConvectionKernels.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>
static void *allocate(void *, size_t n) { return malloc(n); }
static void release(void *, void *p, size_t) { free(p); }
// `threads` caller threads, `kinds` different Options (kind = thread % kinds: per-texture weights), each thread with its own
// input, hammer EncodeBC1 / EncodeBC7 together; every output must equal what the same call gives alone afterwards.  With more
// threads than a coalesced launch takes (256 groups; CVTTMI_DROPIN_MAX_GROUPS) a leader must still carry its own request.
static int many(int threads, int kinds)
{
    const int REPS = 6;
    std::vector<cvtt::PixelBlockU8> tin(threads * 8);
    std::vector<uint8_t> tout(threads * 128, 0xEE), want(threads * 128, 0);
    std::vector<cvtt::Options> opt(threads);
    cvtt::BC7EncodingPlan plan;
    cvtt::Kernels::ConfigureBC7EncodingPlanFromQuality(plan, 10);
    for (int t = 0; t < threads; t++)
    {
        for (int b = 0; b < 8; b++)
            for (int p = 0; p < 16; p++)
                for (int c = 0; c < 4; c++)
                    tin[t * 8 + b].m_pixels[p][c] = (uint8_t)((t * 37 + b * 29 + p * 13 + c * 71 + (p * p + t) * (c + 1)) & 0xFF);
        opt[t].redWeight = 0.25f + 0.125f * (float)(t % kinds);
    }
    auto one = [&](int t, uint8_t *dst) {
        if ((t % kinds) & 1)
            cvtt::Kernels::EncodeBC7(dst, &tin[t * 8], opt[t], plan);
        else
            cvtt::Kernels::EncodeBC1(dst, &tin[t * 8], opt[t]);
    };
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++)
        th.push_back(std::thread([&, t] { for (int rep = 0; rep < REPS; rep++) one(t, &tout[t * 128]); }));
    for (int t = 0; t < threads; t++)
        th[t].join();
    int bad = 0;
    for (int t = 0; t < threads; t++)
    {
        one(t, &want[t * 128]);
        bad += memcmp(&tout[t * 128], &want[t * 128], ((t % kinds) & 1) ? 128 : 64) != 0;
    }
    printf("many %d\n", bad);
    return 0;
}
int main(int argc, char **argv)
{
    cvtt::Options options;
    cvtt::BC7EncodingPlan plan;
    if (sizeof(options) != 44 || sizeof(plan) != 808 || options.refineRoundsIIC != 8 || !plan.mode6Enabled)
        return 2;
    // plan configuration is host-side: works without a device
    cvtt::BC7FineTuningParams ft;
    cvtt::BC7EncodingPlan qplan, fplan;
    cvtt::Kernels::ConfigureBC7EncodingPlanFromQuality(qplan, 20);
    if (sizeof(ft) != 285 || ft.mode6SP != 4 || !cvtt::Kernels::ConfigureBC7EncodingPlanFromFineTuningParams(fplan, ft))
        return 3;
    if (fplan.rgbNumShapesToEvaluate != 242 || qplan.rgbaNumShapesToEvaluate == 0 || qplan.rgbaNumShapesToEvaluate >= 129)
        return 4;
    if (argc < 2)
        return 0; // layout check only
    if (argc >= 4 && !strcmp(argv[1], "many"))
        return many(atoi(argv[2]), atoi(argv[3]));
    cvtt::PixelBlockU8 in[cvtt::NumParallelBlocks];
    cvtt::PixelBlockF16 hdr[cvtt::NumParallelBlocks];
    for (unsigned b = 0; b < 8; b++)
        for (int p = 0; p < 16; p++)
            for (int c = 0; c < 4; c++)
            {
                int v = (29 * b + 13 * p + 71 * c + (p * p + 3 * b) * (c + 1)) & 0xFF;
                if (b >= 4 && c == 3) v = 255;
                in[b].m_pixels[p][c] = (uint8_t)v;
                hdr[b].m_pixels[p][c] = (int16_t)(c == 3 ? 0x3C00 : (((8 + (b + p + c) % 12) << 10) | ((131 * b + 61 * p + 17 * c + 7 * p * p) & 0x3FF)));
            }
    uint8_t out[11][128];
    memset(out, 0, sizeof(out));
    cvtt::Kernels::EncodeBC7(out[0], in, options, plan);
    cvtt::Kernels::EncodeBC1(out[1], in, options);
    cvtt::Kernels::EncodeBC6HU(out[2], hdr, options);
    cvtt::Kernels::EncodeBC6HS(out[3], hdr, options);
    cvtt::ETC2CompressionData *data = cvtt::Kernels::AllocETC2Data(allocate, NULL, options);
    cvtt::Kernels::EncodeETC2(out[4], in, options, data);
    cvtt::Kernels::EncodeETC2RGBA(out[5], in, options, data);
    cvtt::Kernels::ReleaseETC2Data(data, release);
    cvtt::Kernels::EncodeETC2Alpha(out[6], in, options);
    data = cvtt::Kernels::AllocETC2Data(allocate, NULL, options);
    cvtt::Kernels::EncodeETC2PunchthroughAlpha(out[7], in, options, data);
    cvtt::Kernels::ReleaseETC2Data(data, release);
    cvtt::ETC1CompressionData *data1 = cvtt::Kernels::AllocETC1Data(allocate, NULL);
    cvtt::Kernels::EncodeETC1(out[8], in, options, data1);
    cvtt::Kernels::ReleaseETC1Data(data1, release);
    cvtt::Kernels::EncodeBC7(out[9], in, options, qplan);
    // scratch allocated with other Options than the Encode call's: the chroma axes are the allocation's (ETC.cpp:3117-3145)
    cvtt::Options other;
    other.redWeight = 0.9f; other.greenWeight = 0.3f; other.blueWeight = 0.6f;
    data = cvtt::Kernels::AllocETC2Data(allocate, NULL, other);
    cvtt::Kernels::EncodeETC2RGBA(out[10], in, options, data);
    cvtt::Kernels::ReleaseETC2Data(data, release);
    for (int k = 0; k < 11; k++)
    {
        for (int i = 0; i < 128; i++)
            printf("%02x", out[k][i]);
        printf("\n");
    }
    // the reference's callers run one worker thread per group (etc2packer.cpp:215-281).  Concurrent one-group calls are
    // coalesced into shared launches (cxx_api.cpp): sixteen threads, each with its OWN input (the blocks rotated by the thread
    // number) and one of four kinds of call -- two formats, two plans, two Options -- hammer the library together; every
    // result must equal what the same call gives on its own afterwards.
    int bad = 0;
    {
        const int T = 16, REPS = 40;
        static cvtt::PixelBlockU8 tin[T][cvtt::NumParallelBlocks];
        static uint8_t tout[T][128], want[T][128];
        cvtt::Options uni;
        uni.flags |= cvtt::Flags::Uniform;
        for (int i = 0; i < T; i++)
            for (unsigned b = 0; b < 8; b++)
                tin[i][b] = in[(b + i) % 8];
        auto one = [&](int i, uint8_t *dst) {
            switch (i % 4)
            {
            case 0: cvtt::Kernels::EncodeBC7(dst, tin[i], options, plan); break;
            case 1: cvtt::Kernels::EncodeBC7(dst, tin[i], options, qplan); break;
            case 2: cvtt::Kernels::EncodeBC7(dst, tin[i], uni, plan); break;
            default: cvtt::Kernels::EncodeETC2RGBA(dst, tin[i], options, NULL); break;
            }
        };
        std::thread th[T];
        for (int i = 0; i < T; i++)
            th[i] = std::thread([&, i] { for (int rep = 0; rep < REPS; rep++) one(i, tout[i]); });
        for (int i = 0; i < T; i++)
            th[i].join();
        for (int i = 0; i < T; i++)
        {
            one(i, want[i]);
            bad += memcmp(tout[i], want[i], 128) != 0;
        }
        bad += memcmp(want[0], out[0], 128) != 0; // thread 0 has the unrotated blocks and the default call
    }
    printf("threads %d\n", bad);
    return 0;
}
"""


def _build_cxx_client(tmp_path):
    import subprocess
    from convectionkernels_amd import api
    src = tmp_path / "client.cpp"
    src.write_text(CXX_CLIENT)
    exe = tmp_path / "client"
    libdir = os.path.dirname(os.path.abspath(os.environ.get("CVTTMI_LIB", api._LIB_PATH)))
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-pthread", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-lcvtt_mi355x", "-Wl,-rpath," + libdir])
    return exe


def test_cxx_header_is_source_compatible(tmp_path):
    """a client of the reference's cvtt::Kernels API compiles and links against our header + .so"""
    import subprocess
    exe = _build_cxx_client(tmp_path)
    assert subprocess.call([str(exe)]) == 0


@pytest.mark.gpu
def test_cxx_api_matches_oracle(tmp_path, oracle_lib, gpu_ctx):
    import subprocess
    from tests import content
    from oracle import pyref
    exe = _build_cxx_client(tmp_path)
    lines = subprocess.check_output([str(exe), "run"]).decode().split()
    ldr = content.known_answer_group_ldr()
    hdr = content.known_answer_group_hdr()
    o = pyref.make_options()
    from convectionkernels_amd import api
    plan = np.frombuffer(bytes(api.BC7EncodingPlan()), np.uint8).copy()
    rcp = api.Context(0).get_rcp_table()  # a fresh context's built-in table, as the C++ client's
    want = [oracle_lib.encode_bc7(ldr, o, plan, rcp), oracle_lib.encode_bc1(ldr, o, rcp),
            oracle_lib.encode_bc6h(hdr, o, signed=False, rcp=rcp), oracle_lib.encode_bc6h(hdr, o, signed=True, rcp=rcp),
            oracle_lib.encode_etc2(ldr, o, mode=0), oracle_lib.encode_etc2(ldr, o, mode=1), oracle_lib.encode_etc2(ldr, o, mode=2),
            oracle_lib.encode_etc2(ldr, o, mode=4), oracle_lib.encode_etc2(ldr, o, mode=3)]
    qplan = api.BC7EncodingPlan()
    api.ConfigureBC7EncodingPlanFromQuality(qplan, 20)
    want.append(oracle_lib.encode_bc7(ldr, o, np.frombuffer(bytes(qplan), np.uint8).copy(), rcp))
    want.append(oracle_lib.encode_etc2(ldr, o, mode=1, alloc_options=pyref.make_options(weights=(0.9, 0.3, 0.6, 1.0))))
    for k, w in enumerate(want):
        got = bytes.fromhex(lines[k])[:w.size]
        assert got == w.tobytes(), k
    assert lines[11:13] == ["threads", "0"]  # sixteen caller threads, four kinds of call, coalesced launches


@pytest.mark.gpu
@pytest.mark.parametrize("threads,kinds,max_groups", [(300, 1, None), (48, 1, 4), (40, 2, 3), (40, 13, None)])
def test_cxx_api_many_callers(tmp_path, gpu_ctx, threads, kinds, max_groups):
    """more caller threads of one kind than a coalesced launch takes (300 > 256; 48 > 4 with the developer knob), and a pool
    whose kinds outnumber the coalescer's slots: every one-group call returns the bytes it gives alone (VERDICT r4 weak 5,
    ADVICE r4: the leader must carry its own request; different kinds must not serialise behind one flag)"""
    import subprocess
    exe = _build_cxx_client(tmp_path)
    env = dict(os.environ)
    if max_groups:
        env["CVTTMI_DROPIN_MAX_GROUPS"] = str(max_groups)
    out = subprocess.check_output([str(exe), "many", str(threads), str(kinds)], env=env, timeout=600).decode().split()
    assert out == ["many", "0"], out


def test_headline_kernel_needs_no_scratch():
    """the BC7 fast-indexing kernel (BASELINE configs[1]) is built for 4 waves per SIMD (128 VGPRs, <= 10 240 B of LDS) and
    must not spill: 64 B of scratch per lane were 2.8x the algorithmic HBM traffic in round 2 (tools/kernel_resources.py
    reads the code objects of the shipped library)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "tools", "kernel_resources.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    if not os.path.exists(m.READELF):
        pytest.skip("llvm-readelf not installed")
    k = m.kernels()
    head = [v for name, v in k.items() if name.startswith("cvttmi_bc7_kernel<true, false, false>")]
    assert len(head) == 1, sorted(k)
    assert head[0]["scratch_bytes_per_lane"] == 0 and head[0]["vgpr"] <= 128 and head[0]["lds_bytes"] <= 10240, head[0]


@pytest.mark.gpu
def test_library_before_torch_in_one_process():
    """api.load_library() first, `import torch` afterwards, then a context: one HIP runtime per process (round 5: in this order
    the library's own runtime found no device any more; load_library now imports torch first when it is installed)"""
    import subprocess
    import sys
    code = ("from convectionkernels_amd import api; api.load_library(); import torch; "
            "assert torch.cuda.is_available(); c = api.Context(0); import numpy as np; "
            "print(c.encode_bc1(np.zeros((8, 16, 4), np.uint8)).shape)")
    out = subprocess.check_output([sys.executable, "-c", code], cwd=ROOT, timeout=600).decode()
    assert "(8, 8)" in out, out


def test_loading_the_library_does_not_import_torch():
    """api.load_library() preloads the HIP runtime PyTorch ships (when PyTorch is installed) instead of importing torch: a
    numpy-only caller pays no multi-second import and no GPU initialisation, and a later `import torch` finds the one runtime
    of the process."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from convectionkernels_amd import api\n"
            "api.load_library()\n"
            "assert 'torch' not in sys.modules, 'load_library imported torch'\n"
            "maps = open('/proc/self/maps').read()\n"
            "n = len({l.split()[-1] for l in maps.splitlines() if 'libamdhip64' in l})\n"
            "print('runtimes', n)\n" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert p.stdout.split()[-1] == "1"  # exactly one libamdhip64 mapped


@pytest.mark.gpu
def test_numpy_only_process_then_torch_share_one_runtime():
    """a process that loads the library WITHOUT torch, encodes, and only then imports torch: one HIP runtime, both work"""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from convectionkernels_amd import api\n"
            "ctx = api.Context(0)\n"
            "assert 'torch' not in sys.modules\n"
            "b = np.arange(8 * 64, dtype=np.uint8).reshape(8, 16, 4)\n"
            "first = ctx.encode_bc1(b)\n"
            "import torch\n"
            "assert torch.cuda.is_available()\n"
            "t = torch.from_numpy(b).cuda()\n"
            "again = ctx.encode_bc1(t).cpu().numpy()\n"
            "assert (first == again).all() and int(t.sum().item()) == int(b.sum())\n"
            "maps = open('/proc/self/maps').read()\n"
            "print('runtimes', len({l.split()[-1] for l in maps.splitlines() if 'libamdhip64' in l}))\n" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    assert p.stdout.split()[-1] == "1"
