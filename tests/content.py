"""Deterministic test content: groups of 8 PixelBlockU8 covering the cases SURVEY.md §8(c)
lists (random, smooth gradients, solid colours, two-colour, opaque, binary alpha, min-alpha
exactly 250 / 251 -- the BC67.cpp:1072 threshold -- and groups mixing opaque and alpha blocks).
"""
import numpy as np

from convectionkernels_amd import synth


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def mixed_ldr_blocks(seed, groups):
    """(groups*8, 16, 4) uint8."""
    rng = _rng(seed)
    out = np.zeros((groups * 8, 16, 4), np.uint8)
    yy, xx = np.divmod(np.arange(16), 4)
    for g in range(groups):
        kind = g % 12
        for b in range(8):
            blk = out[g * 8 + b]
            k = kind
            if kind == 11:  # mix of everything inside one group
                k = int(rng.integers(0, 11))
            if k == 0:  # uniform random RGBA
                blk[:] = rng.integers(0, 256, (16, 4))
            elif k == 1:  # random opaque
                blk[:] = rng.integers(0, 256, (16, 4))
                blk[:, 3] = 255
            elif k == 2:  # smooth gradient, opaque
                c0 = rng.integers(0, 256, 3).astype(np.float64)
                dx = rng.normal(0, 12, 3)
                dy = rng.normal(0, 12, 3)
                v = c0[None, :] + xx[:, None] * dx[None, :] + yy[:, None] * dy[None, :]
                blk[:, :3] = np.clip(np.rint(v), 0, 255)
                blk[:, 3] = 255
            elif k == 3:  # smooth gradient with smooth alpha
                c0 = rng.integers(0, 256, 4).astype(np.float64)
                dx = rng.normal(0, 10, 4)
                dy = rng.normal(0, 10, 4)
                v = c0[None, :] + xx[:, None] * dx[None, :] + yy[:, None] * dy[None, :]
                blk[:] = np.clip(np.rint(v), 0, 255)
            elif k == 4:  # solid colour (opaque or not)
                blk[:] = rng.integers(0, 256, 4)[None, :]
                if b & 1:
                    blk[:, 3] = 255
            elif k == 5:  # two colours along a random partition
                ca = rng.integers(0, 256, 4)
                cb = rng.integers(0, 256, 4)
                m = rng.integers(0, 2, 16).astype(bool)
                blk[:] = np.where(m[:, None], ca[None, :], cb[None, :])
                if b & 2:
                    blk[:, 3] = 255
            elif k == 6:  # binary (punch-through) alpha over noise
                blk[:] = rng.integers(0, 256, (16, 4))
                blk[:, 3] = np.where(rng.integers(0, 2, 16) > 0, 255, 0)
                if b == 3:
                    blk[:, 3] = 0
                if b == 5:
                    blk[:, 3] = 255
            elif k == 7:  # min alpha exactly 250 / 251 / 254 / 255
                blk[:] = rng.integers(0, 256, (16, 4))
                lo = (250, 251, 254, 255)[b & 3]
                blk[:, 3] = rng.integers(lo, 256, 16)
                blk[int(rng.integers(0, 16)), 3] = lo
            elif k == 8:  # opaque and alpha blocks in the same group
                blk[:] = rng.integers(0, 256, (16, 4))
                if b % 3 == 0:
                    blk[:, 3] = 255
            elif k == 9:  # low-variance noise around a colour (typical photo block)
                c0 = rng.integers(16, 240, 4).astype(np.float64)
                v = c0[None, :] + rng.normal(0, 6, (16, 4))
                blk[:] = np.clip(np.rint(v), 0, 255)
                if b & 1:
                    blk[:, 3] = 255
            else:  # k == 10: extremes / saturated values
                blk[:] = rng.choice(np.array([0, 1, 127, 128, 254, 255], np.uint8), (16, 4))
                if b & 4:
                    blk[:, 3] = 255
    return out


def known_answer_group_ldr():
    """SURVEY.md App. H input group (no RNG)."""
    blk = np.zeros((8, 16, 4), np.uint8)
    for b in range(8):
        for p in range(16):
            for c in range(4):
                v = (29 * b + 13 * p + 71 * c + (p * p + 3 * b) * (c + 1)) & 0xFF
                if b >= 4 and c == 3:
                    v = 255
                blk[b, p, c] = v
    return blk


def config_blocks(seed, width, height, opaque=False):
    return synth.tile_blocks(synth.image_rgba8(seed, width, height, opaque=opaque))


def known_answer_group_hdr():
    """SURVEY.md App. H HDR group: (8,16,4) int16 half bit patterns."""
    hdr = np.zeros((8, 16, 4), np.int16)
    for b in range(8):
        for p in range(16):
            for c in range(4):
                hdr[b, p, c] = ((8 + (b + p + c) % 12) << 10) | ((131 * b + 61 * p + 17 * c + 7 * p * p) & 0x3FF)
            hdr[b, p, 3] = 0x3C00
    return hdr


def mixed_hdr_blocks(seed, groups, signed=False):
    """(groups*8, 16, 4) int16 half bit patterns: random normals, smooth ramps, solid, two-colour,
    tiny/denormal values, huge values, and (signed=True) negative values."""
    rng = _rng(seed)
    out = np.zeros((groups * 8, 16, 4), np.uint16)
    yy, xx = np.divmod(np.arange(16), 4)
    for g in range(groups):
        kind = g % 8
        for b in range(8):
            k = kind if kind != 7 else int(rng.integers(0, 7))
            if k == 0:  # config-3 style: finite positive normals
                r = rng.integers(0, 1 << 62, (16, 3), dtype=np.int64)
                v = (((1 + (r >> 10) % 29) << 10) | (r & 0x3FF))
            elif k == 1:  # smooth ramp in linear space
                c0 = rng.uniform(0.05, 4.0, 3)
                dx = rng.normal(0, 0.2, 3)
                dy = rng.normal(0, 0.2, 3)
                f = np.clip(c0[None] + xx[:, None] * dx[None] + yy[:, None] * dy[None], 0, 60000)
                v = f.astype(np.float16).view(np.uint16).astype(np.int64)
            elif k == 2:  # solid
                v = np.repeat(rng.integers(0, 0x7BFF, (1, 3)), 16, axis=0)
            elif k == 3:  # two colours
                ca = rng.integers(0, 0x7BFF, 3)
                cb = rng.integers(0, 0x7BFF, 3)
                m = rng.integers(0, 2, 16).astype(bool)
                v = np.where(m[:, None], ca[None], cb[None])
            elif k == 4:  # tiny values incl. denormals and zero
                v = rng.integers(0, 0x0800, (16, 3))
            elif k == 5:  # anything, including inf/nan patterns and sign bits
                v = rng.integers(0, 0x10000, (16, 3))
            else:  # narrow range around a bright colour
                base = rng.integers(0x3000, 0x7000, 3)
                v = base[None] + rng.integers(-40, 41, (16, 3))
            v = np.asarray(v, np.int64) & 0xFFFF
            if signed and (b & 1):
                v = v | (rng.integers(0, 2, (16, 3)) << 15)
            out[g * 8 + b, :, :3] = v
            out[g * 8 + b, :, 3] = 0x3C00
    return out.view(np.int16)


def config_blocks_hdr(seed, width, height):
    return synth.tile_blocks(synth.image_f16bits(seed, width, height))


def tile_clamped(img):
    """numpy statement of the reference caller's tiling (etc2packer/etc2packer.cpp:215-247): groups of
    eight horizontally adjacent 4x4 blocks, reads clamped to the last column / row.
    (H,W,C) -> (ceil(H/4) * ceil(ceil(W/4)/8)*8, 16, C)"""
    h, w, c = img.shape
    rows = (h + 3) // 4
    per_row = ((w + 3) // 4 + 7) // 8 * 8
    ys = np.minimum(np.arange(rows * 4), h - 1)
    xs = np.minimum(np.arange(per_row * 4), w - 1)
    big = img[ys][:, xs]
    t = big.reshape(rows, 4, per_row, 4, c).transpose(0, 2, 1, 3, 4)
    return np.ascontiguousarray(t.reshape(rows * per_row, 16, c))


def compact_rows(packed, w, h):
    rows = (h + 3) // 4
    real = (w + 3) // 4
    per_row = (real + 7) // 8 * 8
    return np.ascontiguousarray(packed.reshape(rows, per_row, -1)[:, :real].reshape(rows * real, -1))


def mixed_r11_blocks(seed, groups):
    """(groups*8, 16) int16 PixelBlockScalarS16 content for EAC R11: in-range noise, narrow ranges, flat, ramps,
    boundary values and arbitrary int16 (the reference clamps to 0..2047 / -1023..1023)"""
    rng = _rng(seed)
    out = np.zeros((groups * 8, 16), np.int16)
    for b in range(groups * 8):
        k = b % 8
        if k == 0:
            out[b] = rng.integers(-1500, 2600, 16)
        elif k == 1:
            out[b] = rng.integers(0, 2048, 16)
        elif k == 2:
            out[b] = rng.integers(-1024, 1024, 16)
        elif k == 3:
            out[b] = rng.integers(0, 2048) + rng.integers(-20, 21, 16)
        elif k == 4:
            out[b] = rng.integers(-1000, 1000)
        elif k == 5:
            out[b] = np.linspace(rng.integers(-1024, 2048), rng.integers(-1024, 2048), 16)
        elif k == 6:
            out[b] = rng.choice(np.array([-32768, -1025, -1024, -1, 0, 1, 1023, 1024, 2047, 2048, 32767], np.int16), 16)
        else:
            out[b] = rng.integers(-32768, 32768, 16)
    return out


def alpha_structure_blocks(seed, n):
    """(n,16,4) uint8: per-channel structures that exercise the interpolated-alpha heuristics of BC3/4/5 -- narrow
    ranges, only the terminals, values next to the terminals, ranges with a few terminal outliers, flat, ramps"""
    rng = _rng(seed)
    out = np.zeros((n, 16, 4), np.uint8)
    for b in range(n):
        k = b % 8
        base = rng.integers(0, 256, 4)
        if k == 0:
            v = np.clip(base[None] + rng.integers(-3, 4, (16, 4)), 0, 255)
        elif k == 1:
            v = rng.choice(np.array([0, 255]), (16, 4))
        elif k == 2:
            v = rng.choice(np.array([0, 1, 2, 253, 254, 255]), (16, 4))
        elif k == 3:
            v = np.where(rng.integers(0, 4, (16, 4)) == 0, rng.choice(np.array([0, 255]), (16, 4)),
                         np.clip(base[None] + rng.integers(-20, 21, (16, 4)), 0, 255))
        elif k == 4:
            v = np.repeat(base[None], 16, 0)
        elif k == 5:
            v = np.clip(np.linspace(base, rng.integers(0, 256, 4), 16), 0, 255)
        elif k == 6:
            v = rng.integers(100, 156, (16, 4))
        else:
            v = rng.integers(0, 256, (16, 4))
        out[b] = v
    return out


def punchthrough_blocks(seed, groups_per_kind=4):
    """(N,16,4) uint8 for the punch-through (RGB8A1) encoder: mixed colour content whose alpha channel walks through
    the cases the reference distinguishes -- sparse / dense random cut-outs, fully transparent groups, one fully
    transparent block or a single transparent pixel in an otherwise opaque group (which still sends all eight blocks
    through the punch-through modes), transparent halves (the "ignorable" half-block of either flip), soft alpha
    around the threshold, and all-opaque groups."""
    rng = _rng(seed)
    parts = []

    def colour(i):
        return mixed_ldr_blocks(seed * 131 + i, groups_per_kind)[: groups_per_kind * 8].copy()

    for i, density in enumerate((0.05, 0.3, 0.7, 0.95)):
        b = colour(i)
        b[..., 3] = np.where(rng.random(b.shape[:2]) < density, 0, 255)
        parts.append(b)
    b = colour(10); b[..., 3] = 0; parts.append(b)
    b = colour(11); b[..., 3] = 255; b[::8, :, 3] = 0; parts.append(b)
    b = colour(12); b[..., 3] = 255; b[3::8, 5, 3] = 0; parts.append(b)
    b = colour(13); b[..., 3] = 255; b[:, :8, 3] = 0; parts.append(b)
    b = colour(14); b[..., 3] = 255; b[:, [0, 1, 4, 5, 8, 9, 12, 13], 3] = 0; parts.append(b)
    b = colour(15); b[..., 3] = 255; b[:, 8:, 3] = 0; parts.append(b)
    b = colour(16); b[..., 3] = 255; b[:, [2, 3, 6, 7, 10, 11, 14, 15], 3] = 0; parts.append(b)
    b = colour(17); b[..., 3] = rng.integers(120, 136, b.shape[:2]); parts.append(b)
    b = colour(18); b[..., 3] = 255; parts.append(b)
    b = colour(19); parts.append(b)  # whatever alpha the mixed content carries
    return np.concatenate(parts)


def dark_blocks(seed, n):
    """(4n,16,4) uint8: dark content on which the ETC2 T mode's "zero slot" (hazard H2: the candidate just past a block's
    unique line colours reads as black when another block of the group has more of them) can be the winner: very dark
    noise, dark with bright outliers, black + random pixels with binary alpha, flat colours with black pixels."""
    rng = _rng(seed)
    fam = []
    b = rng.integers(0, 24, (n, 16, 4)).astype(np.uint8); b[..., 3] = 255; fam.append(b)
    b = rng.integers(0, 8, (n, 16, 4)).astype(np.uint8)
    m = rng.random((n, 16)) < 0.15
    b[m] = rng.integers(100, 256, (int(m.sum()), 4)); b[..., 3] = 255; fam.append(b)
    b = np.zeros((n, 16, 4), np.uint8)
    m = rng.random((n, 16)) < 0.3
    b[m] = rng.integers(0, 256, (int(m.sum()), 4)); b[..., 3] = rng.integers(0, 2, (n, 16)) * 255; fam.append(b)
    b = rng.integers(0, 256, (n, 1, 4)).astype(np.uint8).repeat(16, 1); b[::3, ::2, :3] = 0; b[..., 3] = 255; fam.append(b)
    return np.concatenate(fam)


def adversarial_bound_blocks(seed, n):
    """(n,16,4) uint8, n a multiple of 64: content built to sit ON the edges of the BC7 branch-and-bound (DESIGN.md 4.1,
    "soundness of the bounds"): subsets whose scatter matrix has rank one (trace == lambda_max: the bound's subtraction
    cancels), the same with +-1 LSB noise (residual of the order of the rounding allowance n*delta^2, where the bound
    switches on), with perpendicular noise of growing amplitude (the bound crosses the best error somewhere on the way),
    two-colour blocks (every subset collinear), two lines split by a random mask, flat blocks with one outlier, and
    their opaque / constant-alpha / binary-alpha / alpha-on-the-line variants.  Families change every 8 blocks and one
    group in eight mixes them, so the group-wide booleans (BC67.cpp:1069, 1072) see every combination."""
    assert n % 64 == 0
    rng = _rng(seed)
    k = np.arange(16)

    def line(m, amp_perp=0, lsb=0.0):
        c0 = rng.integers(0, 256, (m, 1, 4)).astype(np.float64)
        d = rng.integers(-12, 13, (m, 1, 4)).astype(np.float64)
        t = rng.permuted(np.tile(np.arange(16), (m, 1)), axis=1)[:, :, None] - 7.5
        v = c0 + d * t
        if amp_perp:
            v = v + rng.integers(-amp_perp, amp_perp + 1, (m, 16, 4))
        if lsb:
            v = v + (rng.random((m, 16, 4)) < lsb) * rng.choice(np.array([-1, 1]), (m, 16, 4))
        return np.clip(np.rint(v), 0, 255).astype(np.uint8)

    def two_colour(m, lsb=0.0):
        a = rng.integers(0, 256, (m, 1, 4))
        b = rng.integers(0, 256, (m, 1, 4))
        mask = rng.integers(0, 2, (m, 16, 1)).astype(bool)
        v = np.where(mask, a, b).astype(np.float64)
        if lsb:
            v = v + (rng.random((m, 16, 4)) < lsb) * rng.choice(np.array([-1, 1]), (m, 16, 4))
        return np.clip(v, 0, 255).astype(np.uint8)

    def two_lines(m, amp=0):
        a, b = line(m, amp), line(m, amp)
        mask = rng.integers(0, 2, (m, 16, 1)).astype(bool)
        return np.where(mask, a, b)

    def flat_outlier(m):
        v = rng.integers(0, 256, (m, 1, 4)).repeat(16, 1)
        px = rng.integers(0, 16, m)
        v[np.arange(m), px] = rng.integers(0, 256, (m, 4))
        return v.astype(np.uint8)

    makers = [lambda m: line(m), lambda m: line(m, lsb=0.1), lambda m: line(m, lsb=0.5), lambda m: line(m, 1), lambda m: line(m, 2),
              lambda m: line(m, 3), lambda m: line(m, 4), lambda m: line(m, 6), lambda m: line(m, 8), lambda m: line(m, 12),
              lambda m: two_colour(m), lambda m: two_colour(m, 0.1), lambda m: two_colour(m, 0.5), lambda m: two_lines(m),
              lambda m: two_lines(m, 1), lambda m: two_lines(m, 3), lambda m: flat_outlier(m)]
    groups = n // 8
    per = -(-groups // len(makers))
    fam = np.concatenate([mk(per * 8).reshape(per, 8, 16, 4) for mk in makers])   # (len*per, 8, 16, 4), family-major
    order = rng.permutation(fam.shape[0])[:groups]
    out = fam[order].copy()
    # one group in eight: every block from a different family
    mixed = np.arange(groups) % 8 == 7
    flatf = fam.reshape(-1, 16, 4)
    out[mixed] = flatf[rng.integers(0, flatf.shape[0], (int(mixed.sum()), 8))]
    out = out.reshape(groups * 8, 16, 4)
    # alpha variants per group: as generated / opaque / constant / binary / 250-255 (the allowRGBModes threshold)
    av = rng.integers(0, 5, groups).repeat(8)
    out[av == 1, :, 3] = 255
    c = rng.integers(0, 255, groups * 8)
    sel = av == 2
    out[sel, :, 3] = c[sel, None]
    sel = av == 3
    out[sel, :, 3] = rng.integers(0, 2, (int(sel.sum()), 16)) * 255
    sel = av == 4
    out[sel, :, 3] = rng.integers(250, 256, (int(sel.sum()), 16))
    # inside an opaque group, one block in 16 gets a single translucent pixel (a block can be opaque in an alpha group)
    one = (av == 1) & (rng.integers(0, 16, groups * 8) == 0)
    out[one, 5, 3] = 200
    return np.ascontiguousarray(out.astype(np.uint8))
