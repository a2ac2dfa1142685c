"""Synthetic inputs of SURVEY.md §8(d): SplitMix64 images, tiled into PixelBlocks.

Block (bx,by) = pixels x in [4bx,4bx+4), y in [4by,4by+4), pixel index 4*suby+subx; blocks
are stored row-major; a group is 8 consecutive blocks of one block row (the tiling of the
reference's example caller, etc2packer/etc2packer.cpp:215-248).
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed, n):
    """n draws of SplitMix64 starting from state `seed` (uint64 array)."""
    with np.errstate(over="ignore"):
        s = np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * np.arange(1, n + 1, dtype=np.uint64)
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
