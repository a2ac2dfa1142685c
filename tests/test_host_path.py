"""GPU tests of the boundary's host-pointer entry points (the ones the reference's callers bind, ConvectionKernels.h:242-256
take host pointers): the chunked, double-buffered pipeline returns exactly what the device path returns -- for pageable and
page-locked buffers, for sizes around the chunk boundaries -- and a context shared between streams / threads stays correct."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CHUNK = 1 << 17  # shim.cpp hostPipeline


def _blocks(n, seed=7):
    from convectionkernels_amd import synth
    side = 4 * int(np.ceil(np.sqrt(n)))
    side = (side + 31) // 32 * 32
    b = synth.tile_blocks(synth.image_rgba8(seed, side, side))
    assert b.shape[0] >= n
    return np.ascontiguousarray(b[:n])


@pytest.mark.parametrize("n", [8, CHUNK - 8, CHUNK, CHUNK + 8, 2 * CHUNK + 1024, 3 * CHUNK + 64])
def test_bc1_host_pipeline_equals_device_path(gpu_ctx, n):
    import torch
    b = _blocks(n)
    dev = gpu_ctx.encode_bc1(torch.from_numpy(b).cuda()).cpu().numpy()
    assert (gpu_ctx.encode_bc1(b) == dev).all()
    pin_in = gpu_ctx.host_empty(b.shape)
    pin_in[...] = b
    pin_out = gpu_ctx.host_empty((n, 8))
    pin_out[...] = 0
    gpu_ctx.encode_bc1(pin_in, out=pin_out)
    assert (pin_out == dev).all()
    # mixed: pinned in, pageable out and the reverse
    assert (gpu_ctx.encode_bc1(pin_in) == dev).all()
    gpu_ctx.encode_bc1(b, out=pin_out)
    assert (pin_out == dev).all()


def test_bc7_host_pipeline_equals_device_path(gpu_ctx):
    import torch
    from convectionkernels_amd import api
    n = 2 * CHUNK + 4096  # three chunks; above the 2^19 blocks from which the BC7 hand-over list is used
    b = _blocks(n, seed=2)
    dev = gpu_ctx.encode_bc7(torch.from_numpy(b).cuda()).cpu().numpy()
    assert (gpu_ctx.encode_bc7(b) == dev).all()
    reg = b.copy()
    gpu_ctx.host_register(reg)
    try:
        out = gpu_ctx.host_empty((n, 16))
        gpu_ctx.encode_bc7(reg, api.Options(), api.BC7EncodingPlan(), out=out)
        assert (out == dev).all()
    finally:
        gpu_ctx.host_unregister(reg)


def test_other_formats_host_equals_device(gpu_ctx):
    import torch
    from convectionkernels_amd import synth
    b = _blocks(CHUNK + 512, seed=4)
    t = torch.from_numpy(b).cuda()
    for host, dev in ((gpu_ctx.encode_etc2_rgba, gpu_ctx.encode_etc2_rgba), (gpu_ctx.encode_bc3, gpu_ctx.encode_bc3), (gpu_ctx.encode_etc2_alpha, gpu_ctx.encode_etc2_alpha)):
        assert (host(b) == dev(t).cpu().numpy()).all()
    h = synth.tile_blocks(synth.image_f16bits(3, 64, 64))
    assert (gpu_ctx.encode_bc6h(h) == gpu_ctx.encode_bc6h(torch.from_numpy(h).cuda()).cpu().numpy()).all()
    packed = gpu_ctx.encode_bc7(b[:4096])
    assert (gpu_ctx.decode_bc7(packed) == gpu_ctx.decode_bc7(torch.from_numpy(packed).cuda()).cpu().numpy()).all()


def test_one_context_on_two_streams_and_two_threads(gpu_ctx):
    """the hand-over list and the BC6H scratch exist once per context: launches on different streams are ordered by the
    library (cvtt_mi355x.h, "Streams"), host-pointer calls from different threads are serialised"""
    import torch
    from convectionkernels_amd import synth
    n = 1 << 19  # hand-over list in use
    b1, b2 = _blocks(n, seed=2), _blocks(n, seed=3)
    t1, t2 = torch.from_numpy(b1).cuda(), torch.from_numpy(b2).cuda()
    e1 = gpu_ctx.encode_bc7(t1).cpu().numpy()
    e2 = gpu_ctx.encode_bc7(t2).cpu().numpy()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for i in range(6):
        with torch.cuda.stream(s1 if i % 2 == 0 else s2):
            outs.append(gpu_ctx.encode_bc7(t1 if i % 2 == 0 else t2))
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        assert (o.cpu().numpy() == (e1 if i % 2 == 0 else e2)).all(), i
    h = synth.tile_blocks(synth.image_f16bits(3, 256, 256))
    th = torch.from_numpy(h).cuda()
    eh = gpu_ctx.encode_bc6h(th).cpu().numpy()
    with torch.cuda.stream(s1):
        a = gpu_ctx.encode_bc6h(th)
    with torch.cuda.stream(s2):
        c = gpu_ctx.encode_bc6h(th)
    torch.cuda.synchronize()
    assert (a.cpu().numpy() == eh).all() and (c.cpu().numpy() == eh).all()
    # host-pointer calls from two threads on the same context
    res = {}

    def work(key, blocks):
        res[key] = gpu_ctx.encode_bc7(blocks)
    ths = [threading.Thread(target=work, args=(k, v)) for k, v in (("a", b1), ("b", b2), ("c", b1))]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert (res["a"] == e1).all() and (res["b"] == e2).all() and (res["c"] == e1).all()


def test_two_contexts_on_two_streams_overlap_and_agree(gpu_ctx):
    """the pipelined caller of bench.py's `two_streams` leg: two contexts, each on its own stream, encode different images at the
    same time (nothing is shared between contexts: each has its hand-over list, counters and BC6H work space); every output equals
    the one-context result.  Mixed formats, so that kernels of different shapes are resident together."""
    import torch
    from convectionkernels_amd import api, synth
    n = 1 << 18
    b1, b2 = _blocks(n, seed=12), _blocks(n, seed=13)
    t1, t2 = torch.from_numpy(b1).cuda(), torch.from_numpy(b2).cuda()
    h = torch.from_numpy(synth.tile_blocks(synth.image_f16bits(3, 256, 256))).cuda()
    want = {"bc7a": gpu_ctx.encode_bc7(t1).cpu().numpy(), "bc7b": gpu_ctx.encode_bc7(t2).cpu().numpy(),
            "etc": gpu_ctx.encode_etc2_rgba(t2[:32768]).cpu().numpy(), "hdr": gpu_ctx.encode_bc6h(h).cpu().numpy(),
            "bc1": gpu_ctx.encode_bc1(t1).cpu().numpy()}
    other = api.Context(0)
    other.set_rcp_table(gpu_ctx.get_rcp_table())
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    got = []
    for i in range(4):
        got.append(("bc7a", gpu_ctx.encode_bc7(t1, stream=sa.cuda_stream)))
        got.append(("bc7b", other.encode_bc7(t2, stream=sb.cuda_stream)))
        got.append(("hdr", gpu_ctx.encode_bc6h(h, stream=sa.cuda_stream)))
        got.append(("etc", other.encode_etc2_rgba(t2[:32768], stream=sb.cuda_stream)))
        got.append(("bc1", other.encode_bc1(t1, stream=sb.cuda_stream)))
    torch.cuda.synchronize()
    for k, o in got:
        assert (o.cpu().numpy() == want[k]).all(), k


def test_wrong_buffers_are_refused(gpu_ctx):
    import torch
    from convectionkernels_amd import api
    b = _blocks(64)
    t = torch.from_numpy(b).cuda()
    with pytest.raises(api.CvttError):
        gpu_ctx.encode_bc7(b, out=np.empty((63, 16), np.uint8))            # too small
    with pytest.raises(api.CvttError):
        gpu_ctx.encode_bc7(b, out=np.empty((64, 32), np.uint8)[:, :16])    # strided
    with pytest.raises(api.CvttError):
        gpu_ctx.encode_bc7(t, out=torch.empty((64, 16), dtype=torch.uint8))  # CPU tensor as the result of a device call
    with pytest.raises(api.CvttError):
        gpu_ctx.encode_bc7(t, out=torch.empty((32, 16), dtype=torch.uint8, device="cuda"))
    with pytest.raises(api.CvttError):
        gpu_ctx.encode_bc1(torch.from_numpy(b))                             # CPU tensor as input
    with pytest.raises(api.CvttError):
        gpu_ctx.decode_bc7(torch.zeros((8, 16), dtype=torch.uint8))          # CPU tensor into the device decoder
    with pytest.raises(api.CvttError):
        gpu_ctx.encode_bc1(t, out=torch.empty((64, 8), dtype=torch.uint8, device="cuda").t())  # non-contiguous
