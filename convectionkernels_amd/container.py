"""Texture containers for the packed blocks ("container writer").

The encoders return blocks row-major, ceil(W/4) per row -- exactly the payload order of a KTX 1.1 mip level and of a
DDS surface, so a container is a header in front of the device output.  KTX mirrors what the reference's example
packer writes (etc2packer.cpp:116-197: one level, one face, no key/value data, little endian); DDS (DX10 header) is
added for the BC formats.  `read_ktx` / `read_dds` exist for the tests and for tools that want the blocks back."""
import struct

import numpy as np

# name -> (bytes per block, GL internal format, GL base internal format, DXGI format or None)
_GL_RGB, _GL_RGBA, _GL_RED, _GL_RG = 0x1907, 0x1908, 0x1903, 0x8227
FORMATS = {
    "etc1": (8, 0x8D64, _GL_RGB, None),             # GL_ETC1_RGB8_OES
    "etc2": (8, 0x9274, _GL_RGB, None),             # GL_COMPRESSED_RGB8_ETC2
    "etc2rgba": (16, 0x9278, _GL_RGBA, None),       # GL_COMPRESSED_RGBA8_ETC2_EAC
    "etc2punchthrough": (8, 0x9276, _GL_RGBA, None),  # GL_COMPRESSED_RGB8_PUNCHTHROUGH_ALPHA1_ETC2
    "r11u": (8, 0x9270, _GL_RED, None),             # GL_COMPRESSED_R11_EAC
    "r11s": (8, 0x9271, _GL_RED, None),             # GL_COMPRESSED_SIGNED_R11_EAC
    "bc1": (8, 0x83F1, _GL_RGBA, 71),               # GL_COMPRESSED_RGBA_S3TC_DXT1_EXT / DXGI_FORMAT_BC1_UNORM
    "bc2": (16, 0x83F2, _GL_RGBA, 74),
    "bc3": (16, 0x83F3, _GL_RGBA, 77),
    "bc4u": (8, 0x8DBB, _GL_RED, 80),               # GL_COMPRESSED_RED_RGTC1
    "bc4s": (8, 0x8DBC, _GL_RED, 81),
    "bc5u": (16, 0x8DBD, _GL_RG, 83),
    "bc5s": (16, 0x8DBE, _GL_RG, 84),
    "bc6hu": (16, 0x8E8F, _GL_RGB, 95),             # GL_COMPRESSED_RGB_BPTC_UNSIGNED_FLOAT
    "bc6hs": (16, 0x8E8E, _GL_RGB, 96),
    "bc7": (16, 0x8E8C, _GL_RGBA, 98),              # GL_COMPRESSED_RGBA_BPTC_UNORM
}
# the names the reference's packer uses on its command line (etc2packer.cpp:34-42)
ALIASES = {"etc2rgb": "etc2"}

_KTX_ID = bytes([0xAB, 0x4B, 0x54, 0x58, 0x20, 0x31, 0x31, 0xBB, 0x0D, 0x0A, 0x1A, 0x0A])
_DDS_MAGIC = b"DDS "


def canonical(fmt):
    fmt = ALIASES.get(fmt, fmt)
    if fmt not in FORMATS:
        raise ValueError("unknown texture format %r" % (fmt,))
    return fmt


def _payload(fmt, width, height, packed):
    per = FORMATS[fmt][0]
    bw, bh = (width + 3) // 4, (height + 3) // 4
    if hasattr(packed, "detach"):  # torch tensor (device or host)
        packed = packed.detach().cpu().numpy()
    data = np.ascontiguousarray(packed, dtype=np.uint8).reshape(-1)
    if data.size != bw * bh * per:
        raise ValueError("%s %dx%d needs %d bytes of blocks, got %d" % (fmt, width, height, bw * bh * per, data.size))
    return data


def ktx_bytes(fmt, width, height, packed):
    """KTX 1.1 file image: the 64-byte header of etc2packer.cpp:123-147, imageSize, blocks."""
    fmt = canonical(fmt)
    data = _payload(fmt, width, height, packed)
    _, internal, base, _ = FORMATS[fmt]
    header = _KTX_ID + struct.pack("<13I", 0x04030201, 0, 1, 0, internal, base, width, height, 0, 0, 1, 1, 0)
    return header + struct.pack("<I", data.size) + data.tobytes()


def write_ktx(path, fmt, width, height, packed):
    with open(path, "wb") as f:
        f.write(ktx_bytes(fmt, width, height, packed))


def read_ktx(path_or_bytes):
    """-> (format name, width, height, blocks (N, bytesPerBlock) uint8)"""
    raw = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
    if raw[:12] != _KTX_ID:
        raise ValueError("not a KTX 1.1 file")
    f = struct.unpack_from("<13I", raw, 12)
    if f[0] != 0x04030201:
        raise ValueError("big-endian KTX is not supported")
    internal, width, height, kv = f[4], f[6], f[7], f[12]
    names = [n for n, v in FORMATS.items() if v[1] == internal]
    if not names:
        raise ValueError("unsupported glInternalFormat 0x%x" % internal)
    off = 64 + kv
    size, = struct.unpack_from("<I", raw, off)
    per = FORMATS[names[0]][0]
    blocks = np.frombuffer(raw, np.uint8, size, off + 4).reshape(-1, per).copy()
    return names[0], width, height, blocks


def dds_bytes(fmt, width, height, packed):
    """DDS with the DX10 extension header (BC formats only): one 2-D surface, one mip level."""
    fmt = canonical(fmt)
    per, _, _, dxgi = FORMATS[fmt]
    if dxgi is None:
        raise ValueError("%s has no DXGI format; use KTX" % fmt)
    data = _payload(fmt, width, height, packed)
    DDSD_CAPS, DDSD_HEIGHT, DDSD_WIDTH, DDSD_PIXELFORMAT, DDSD_LINEARSIZE = 0x1, 0x2, 0x4, 0x1000, 0x80000
    flags = DDSD_CAPS | DDSD_HEIGHT | DDSD_WIDTH | DDSD_PIXELFORMAT | DDSD_LINEARSIZE
    pixel_format = struct.pack("<2I4s5I", 32, 0x4, b"DX10", 0, 0, 0, 0, 0)  # DDPF_FOURCC
    header = struct.pack("<7I", 124, flags, height, width, data.size, 0, 1) + bytes(44) + pixel_format + \
        struct.pack("<5I", 0x1000, 0, 0, 0, 0)  # DDSCAPS_TEXTURE
    assert len(header) == 124
    dx10 = struct.pack("<5I", dxgi, 3, 0, 1, 0)  # D3D10_RESOURCE_DIMENSION_TEXTURE2D, array size 1
    return _DDS_MAGIC + header + dx10 + data.tobytes()


def write_dds(path, fmt, width, height, packed):
    with open(path, "wb") as f:
        f.write(dds_bytes(fmt, width, height, packed))


def read_dds(path_or_bytes):
    raw = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
    if raw[:4] != _DDS_MAGIC or raw[84:88] != b"DX10":
        raise ValueError("not a DX10 DDS file")
    height, width = struct.unpack_from("<2I", raw, 12)
    dxgi, = struct.unpack_from("<I", raw, 128)
    names = [n for n, v in FORMATS.items() if v[3] == dxgi]
    if not names:
        raise ValueError("unsupported DXGI format %d" % dxgi)
    per = FORMATS[names[0]][0]
    n = ((width + 3) // 4) * ((height + 3) // 4)
    blocks = np.frombuffer(raw, np.uint8, n * per, 148).reshape(n, per).copy()
    return names[0], width, height, blocks
