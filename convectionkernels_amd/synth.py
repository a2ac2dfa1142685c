"""Synthetic inputs of SURVEY.md §8(d): SplitMix64 images, tiled into PixelBlocks.

Block (bx,by) = pixels x in [4bx,4bx+4), y in [4by,4by+4), pixel index 4*suby+subx; blocks
are stored row-major; a group is 8 consecutive blocks of one block row (the tiling of the
reference's example caller, etc2packer/etc2packer.cpp:215-248).
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed, n, start=0):
    """Draws start .. start+n-1 of SplitMix64 from state `seed` (uint64 array); the generator is counter based, so a
    shard of a stream is generated without the draws before it."""
    with np.errstate(over="ignore"):
        s = np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * np.arange(start + 1, start + n + 1, dtype=np.uint64)
        z = s
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def random_bytes(seed, nbytes):
    draws = splitmix64(seed, (nbytes + 7) // 8)
    return draws.view(np.uint8)[:nbytes].copy()  # little-endian bytes of each draw


def image_rgba8(seed, width, height, opaque=False):
    img = random_bytes(seed, width * height * 4).reshape(height, width, 4)
    if opaque:
        img[..., 3] = 255
    return img


def image_rgba8_rows(seed, width, height, row0, row1, opaque=False):
    """Pixel rows [row0, row1) of image_rgba8(seed, width, height): what one rank of a block-row sharded job needs."""
    assert 0 <= row0 <= row1 <= height and (width * 4) % 8 == 0
    first = row0 * width * 4 // 8
    count = (row1 - row0) * width * 4 // 8
    img = splitmix64(seed, count, start=first).view(np.uint8).reshape(row1 - row0, width, 4).copy()
    if opaque:
        img[..., 3] = 255
    return img


def tile_blocks(img):
    """(H,W,C) image -> (H/4*W/4, 16, C) PixelBlock array, row-major blocks."""
    h, w, c = img.shape
    assert h % 4 == 0 and w % 4 == 0
    t = img.reshape(h // 4, 4, w // 4, 4, c).transpose(0, 2, 1, 3, 4)
    return np.ascontiguousarray(t.reshape((h // 4) * (w // 4), 16, c))


def image_f16bits(seed, width, height):
    """Config 3: finite positive normal halfs, alpha = 1.0 (0x3C00); int16 bit patterns."""
    r = splitmix64(seed, width * height * 3).reshape(height, width, 3)
    half = (((np.uint64(1) + (r >> np.uint64(10)) % np.uint64(29)) << np.uint64(10)) | (r & np.uint64(0x3FF))).astype(np.uint16)
    out = np.empty((height, width, 4), np.uint16)
    out[..., :3] = half
    out[..., 3] = 0x3C00
    return out.view(np.int16)


def content_families(n, seed=2026):
    """Eight synthetic content families of n PixelBlockU8 each, (n, 16, 4) uint8 -- the throughput of the BC7 search
    depends on content because its pruning does: uniform noise (BASELINE configs), opaque noise, smooth
    gradients with and without alpha, photo-like low-variance blocks, two-colour blocks, punch-through alpha and
    near-opaque alpha.  Used by bench.py's `content_families` key, tools/family_bench.py and tools/stress_parity.py."""
    rng = np.random.Generator(np.random.PCG64(seed))
    yy, xx = np.divmod(np.arange(16), 4)
    f = {}
    f["noise"] = rng.integers(0, 256, (n, 16, 4), dtype=np.uint8)
    o = rng.integers(0, 256, (n, 16, 4), dtype=np.uint8)
    o[:, :, 3] = 255
    f["opaque noise"] = o
    c0 = rng.integers(0, 256, (n, 1, 4)).astype(np.float32)
    dx = rng.normal(0, 10, (n, 1, 4)).astype(np.float32)
    dy = rng.normal(0, 10, (n, 1, 4)).astype(np.float32)
    g = np.clip(np.rint(c0 + xx[None, :, None] * dx + yy[None, :, None] * dy), 0, 255).astype(np.uint8)
    f["gradient rgba"] = g.copy()
    g2 = g.copy()
    g2[:, :, 3] = 255
    f["gradient opaque"] = g2
    ph = np.clip(np.rint(c0 + rng.normal(0, 5, (n, 16, 4))), 0, 255).astype(np.uint8)
    ph[n // 2:, :, 3] = 255
    f["photo-like"] = ph
    ca = rng.integers(0, 256, (n, 1, 4), dtype=np.uint8)
    cb = rng.integers(0, 256, (n, 1, 4), dtype=np.uint8)
    m = rng.integers(0, 2, (n, 16, 1)).astype(bool)
    tc = np.where(m, ca, cb).astype(np.uint8)
    tc[::2, :, 3] = 255
    f["two colours"] = tc
    pt = rng.integers(0, 256, (n, 16, 4), dtype=np.uint8)
    pt[:, :, 3] = np.where(rng.integers(0, 2, (n, 16)) > 0, 255, 0)
    f["punch-through alpha"] = pt
    hi = rng.integers(0, 256, (n, 16, 4), dtype=np.uint8)
    hi[:, :, 3] = rng.integers(248, 256, (n, 16))
    f["alpha 248..255"] = hi
    return f


def hdr_content_families(n, seed=2027):
    """Three kinds of PixelBlockF16 content, (n, 16, 4) int16 half bit patterns, alpha = 1.0: the BC6H search skips what the
    delta coding of the end points rules out, so its rate depends on whether a block's end points are close
    together.  `noise`: BASELINE config 3 (every channel an independent positive normal half); `smooth ramps`: a linear
    gradient per block in linear light; `narrow range`: bright colours within +-40 half codes of a base colour."""
    rng = np.random.Generator(np.random.PCG64(seed))
    yy, xx = np.divmod(np.arange(16), 4)
    f = {}
    r = rng.integers(0, 1 << 62, (n, 16, 3), dtype=np.int64)
    noise = np.empty((n, 16, 4), np.uint16)
    noise[..., :3] = (((1 + (r >> 10) % 29) << 10) | (r & 0x3FF)).astype(np.uint16)
    noise[..., 3] = 0x3C00
    f["noise"] = noise.view(np.int16)
    c0 = rng.uniform(0.05, 4.0, (n, 1, 3))
    dx = rng.normal(0, 0.2, (n, 1, 3))
    dy = rng.normal(0, 0.2, (n, 1, 3))
    lin = np.clip(c0 + xx[None, :, None] * dx + yy[None, :, None] * dy, 0, 60000)
    ramp = np.empty((n, 16, 4), np.uint16)
    ramp[..., :3] = lin.astype(np.float16).view(np.uint16)
    ramp[..., 3] = 0x3C00
    f["smooth ramps"] = ramp.view(np.int16)
    base = rng.integers(0x3000, 0x7000, (n, 1, 3))
    nar = np.empty((n, 16, 4), np.uint16)
    nar[..., :3] = (base + rng.integers(-40, 41, (n, 16, 3))).astype(np.uint16)
    nar[..., 3] = 0x3C00
    f["narrow range"] = nar.view(np.int16)
    return f
