"""TEST INFRASTRUCTURE ONLY: ctypes drivers for the two CPU checkers.

* ``RefLib``    -- oracle/_ref/libcvtt_ref*.so, the unmodified reference compiled by
                   oracle/Makefile (present only where /root/reference was available
                   at build time, or where the prebuilt .so travelled).
* ``OracleLib`` -- oracle/libcvtt_oracle.so, our plain-C restatement.

Nothing in the product package imports this module.
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

SIZEOF_OPTIONS = 44
SIZEOF_BC7_PLAN = 808
SIZEOF_BC7_FINETUNE = 285

FLAG_BC7_FAST_INDEXING = 0x008
FLAG_BC7_TRY_SINGLE_COLOR = 0x010
FLAG_BC7_RESPECT_PUNCHTHROUGH = 0x020
FLAG_BC6H_FAST_INDEXING = 0x040
FLAG_S3TC_EXHAUSTIVE = 0x080
FLAG_S3TC_PARANOID = 0x100
FLAG_UNIFORM = 0x200
FLAGS_DEFAULT = FLAG_BC7_FAST_INDEXING | FLAG_S3TC_PARANOID
FLAGS_BETTER = FLAG_S3TC_PARANOID | FLAG_S3TC_EXHAUSTIVE


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


class RefLib:
    def __init__(self, fast=False):
        name = "libcvtt_ref_fast.so" if fast else "libcvtt_ref.so"
        path = os.path.join(HERE, "_ref", name)
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = ctypes.CDLL(path)
        L = self.lib
        for n in ("ref_sizeof_options", "ref_sizeof_bc7_plan", "ref_sizeof_bc7_finetune"):
            getattr(L, n).restype = ctypes.c_size_t
        assert L.ref_sizeof_options() == SIZEOF_OPTIONS
        assert L.ref_sizeof_bc7_plan() == SIZEOF_BC7_PLAN
        assert L.ref_sizeof_bc7_finetune() == SIZEOF_BC7_FINETUNE
        L.ref_encode_etc2.restype = ctypes.c_int

    @staticmethod
    def available(fast=False):
        name = "libcvtt_ref_fast.so" if fast else "libcvtt_ref.so"
        return os.path.exists(os.path.join(HERE, "_ref", name))

    def default_options(self):
        buf = np.zeros(SIZEOF_OPTIONS, np.uint8)
        self.lib.ref_default_options(buf.ctypes.data_as(ctypes.c_void_p))
        return buf

    def default_plan(self):
        buf = np.zeros(SIZEOF_BC7_PLAN, np.uint8)
        self.lib.ref_default_bc7_plan(buf.ctypes.data_as(ctypes.c_void_p))
        return buf

    def plan_from_quality(self, q):
        buf = np.zeros(SIZEOF_BC7_PLAN, np.uint8)
        self.lib.ref_bc7_plan_from_quality(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(q))
        return buf

    def plan_from_finetune(self, params):
        params = np.ascontiguousarray(params, np.uint8)
        assert params.size == SIZEOF_BC7_FINETUNE
        buf = np.zeros(SIZEOF_BC7_PLAN, np.uint8)
        self.lib.ref_bc7_plan_from_finetune(buf.ctypes.data_as(ctypes.c_void_p), params.ctypes.data_as(ctypes.c_void_p))
        return buf

    def probe_rcp(self):
        out = np.zeros(17, np.float32)
        self.lib.ref_probe_rcp(out.ctypes.data_as(ctypes.c_void_p))
        return out

    def encode_bc7(self, blocks, options, plan):
        blocks, pb = _u8(blocks)
        n = blocks.size // 64
        assert n % 8 == 0
        out = np.zeros(n * 16, np.uint8)
        self.lib.ref_encode_bc7(out.ctypes.data_as(ctypes.c_void_p), pb, ctypes.c_size_t(n),
                                options.ctypes.data_as(ctypes.c_void_p), plan.ctypes.data_as(ctypes.c_void_p))
        return out.reshape(n, 16)

    def encode_bc1(self, blocks, options):
        blocks, pb = _u8(blocks)
        n = blocks.size // 64
        assert n % 8 == 0
        out = np.zeros(n * 8, np.uint8)
        self.lib.ref_encode_bc1(out.ctypes.data_as(ctypes.c_void_p), pb, ctypes.c_size_t(n),
                                options.ctypes.data_as(ctypes.c_void_p))
        return out.reshape(n, 8)

    def encode_bc6h(self, blocks_f16bits, options, signed=False):
        b = np.ascontiguousarray(blocks_f16bits, dtype=np.int16)
        n = b.size // 64
        assert n % 8 == 0
        out = np.zeros(n * 16, np.uint8)
        self.lib.ref_encode_bc6h(out.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p),
                                 ctypes.c_size_t(n), options.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(int(signed)))
        return out.reshape(n, 16)

    def encode_etc2(self, blocks, options, mode, alloc_options=None):
        """mode 0: ETC2 RGB (8 B), 1: ETC2 RGBA (16 B), 2: EAC alpha (8 B), 3: ETC1 (8 B), 4: ETC2 punch-through alpha (8 B).
        alloc_options: the Options handed to AllocETC2Data (default: the encode options)."""
        blocks, pb = _u8(blocks)
        n = blocks.size // 64
        assert n % 8 == 0
        per = 16 if mode == 1 else 8
        out = np.zeros(n * per, np.uint8)
        ao = options if alloc_options is None else alloc_options
        self.lib.ref_encode_etc2_alloc.restype = ctypes.c_int
        rc = self.lib.ref_encode_etc2_alloc(out.ctypes.data_as(ctypes.c_void_p), pb, ctypes.c_size_t(n),
                                            options.ctypes.data_as(ctypes.c_void_p), ao.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(mode))
        assert rc == 0
        return out.reshape(n, per)

    def encode_s3tc(self, blocks, options, fmt):
        """fmt 2..7 = BC2, BC3, BC4U, BC4S, BC5U, BC5S; (N,16,4) uint8 (int8 bit patterns for the signed formats)"""
        blocks, pb = _u8(blocks)
        n = blocks.size // 64
        assert n % 8 == 0
        per = 8 if fmt in (4, 5) else 16
        out = np.zeros(n * per, np.uint8)
        self.lib.ref_encode_s3tc(out.ctypes.data_as(ctypes.c_void_p), pb, ctypes.c_size_t(n),
                                 options.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(fmt))
        return out.reshape(n, per)

    def encode_eac11(self, blocks_s16, options, signed=False):
        """EncodeETC2Alpha11: (N,16) int16 -> (N,8) uint8"""
        b = np.ascontiguousarray(blocks_s16, dtype=np.int16)
        n = b.size // 16
        assert n % 8 == 0
        out = np.zeros(n * 8, np.uint8)
        self.lib.ref_encode_eac11(out.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n),
                                  options.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(int(signed)))
        return out.reshape(n, 8)

    MT_FORMATS = {"bc7": (0, 64, 16), "bc1": (1, 64, 8), "bc6hu": (2, 128, 16), "bc6hs": (3, 128, 16), "etc2": (4, 64, 8), "etc2rgba": (5, 64, 16)}

    def encode_mt(self, fmt, blocks, options, plan=None, threads=1, budget_s=5.0, chunk_blocks=64):
        """Time-bounded multi-threaded run (std::thread inside the shim, no Python in the loop): returns
        (out[:done], done_blocks, seconds).  Only the prefix that was finished within the budget is returned."""
        code, in_bytes, out_bytes = self.MT_FORMATS[fmt]
        b = np.ascontiguousarray(blocks)
        n = b.nbytes // in_bytes
        out = np.zeros(n * out_bytes, np.uint8)
        done = ctypes.c_uint64(0)
        secs = ctypes.c_double(0.0)
        self.lib.ref_encode_mt.restype = ctypes.c_int
        rc = self.lib.ref_encode_mt(ctypes.c_int(code), out.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n),
                                    options.ctypes.data_as(ctypes.c_void_p), plan.ctypes.data_as(ctypes.c_void_p) if plan is not None else None,
                                    ctypes.c_int(threads), ctypes.c_double(budget_s), ctypes.c_size_t(chunk_blocks),
                                    ctypes.byref(done), ctypes.byref(secs))
        if rc != 0:
            raise RuntimeError("ref_encode_mt rc=%d" % rc)
        d = int(done.value)
        return out.reshape(n, out_bytes)[:d], d, float(secs.value)

    def decode_bc6h(self, bc, signed=False):
        bc, pb = _u8(bc)
        n = bc.size // 16
        assert n % 8 == 0
        out = np.zeros(n * 64, np.int16)
        self.lib.ref_decode_bc6h(out.ctypes.data_as(ctypes.c_void_p), pb, ctypes.c_size_t(n), ctypes.c_int(int(signed)))
        return out.reshape(n, 16, 4)

    def decode_bc7(self, bc):
        bc, pb = _u8(bc)
        n = bc.size // 16
        assert n % 8 == 0
        out = np.zeros(n * 64, np.uint8)
        self.lib.ref_decode_bc7(out.ctypes.data_as(ctypes.c_void_p), pb, ctypes.c_size_t(n))
        return out.reshape(n, 16, 4)


def make_options(flags=FLAGS_DEFAULT, threshold=0.5, weights=None, refine_bc7=2, refine_bc6h=3,
                 refine_iic=8, refine_s3tc=2, seed_points=4):
    """Byte image of cvtt::Options (ConvectionKernels.h:73-103), 44 bytes."""
    if weights is None:
        weights = (np.float32(0.2125) / np.float32(0.7154), np.float32(1.0),
                   np.float32(0.0721) / np.float32(0.7154), np.float32(1.0))
    buf = np.zeros(SIZEOF_OPTIONS, np.uint8)
    buf[0:4] = np.frombuffer(np.uint32(flags).tobytes(), np.uint8)
    f = np.array([threshold, weights[0], weights[1], weights[2], weights[3]], np.float32)
    buf[4:24] = np.frombuffer(f.tobytes(), np.uint8)
    i = np.array([refine_bc7, refine_bc6h, refine_iic, refine_s3tc, seed_points], np.int32)
    buf[24:44] = np.frombuffer(i.tobytes(), np.uint8)
    return buf


class OracleLib:
    """oracle/libcvtt_oracle.so -- the plain-C restatement (cvtt_oracle.c)."""

    def __init__(self):
        path = os.path.join(HERE, "libcvtt_oracle.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path + " (run `make -C oracle oracle`)")
        self.lib = ctypes.CDLL(path)
        L = self.lib
        L.orc_sizeof_options.restype = ctypes.c_size_t
        L.orc_sizeof_bc7_plan.restype = ctypes.c_size_t
        assert L.orc_sizeof_options() == SIZEOF_OPTIONS
        assert L.orc_sizeof_bc7_plan() == SIZEOF_BC7_PLAN
        L.orc_encode_bc7.restype = ctypes.c_int
        L.orc_encode_bc1.restype = ctypes.c_int
        L.orc_encode_bc6h.restype = ctypes.c_int
        L.orc_encode_etc2.restype = ctypes.c_int

    def probe_rcp(self):
        out = np.zeros(17, np.float32)
        self.lib.orc_probe_rcp(out.ctypes.data_as(ctypes.c_void_p))
        return out

    def encode_bc7(self, blocks, options, plan, rcp=None, threads=1):
        blocks, pb = _u8(blocks)
        n = blocks.size // 64
        assert n % 8 == 0
        out = np.zeros(n * 16, np.uint8)
        rcp_p = None
        if rcp is not None:
            rcp = np.ascontiguousarray(rcp, np.float32)
            assert rcp.size == 17
            rcp_p = rcp.ctypes.data_as(ctypes.c_void_p)
        rc = self.lib.orc_encode_bc7(out.ctypes.data_as(ctypes.c_void_p), pb, ctypes.c_size_t(n),
                                     options.ctypes.data_as(ctypes.c_void_p), plan.ctypes.data_as(ctypes.c_void_p),
                                     rcp_p, ctypes.c_int(threads))
        if rc != 0:
            raise RuntimeError("orc_encode_bc7 rc=%d" % rc)
        return out.reshape(n, 16)

    def encode_bc6h(self, blocks_f16bits, options, signed=False, rcp=None, threads=1):
        b = np.ascontiguousarray(blocks_f16bits, dtype=np.int16)
        n = b.size // 64
        assert n % 8 == 0
        out = np.zeros(n * 16, np.uint8)
        rcp_p = None
        if rcp is not None:
            rcp = np.ascontiguousarray(rcp, np.float32)
            rcp_p = rcp.ctypes.data_as(ctypes.c_void_p)
        rc = self.lib.orc_encode_bc6h(out.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n),
                                      options.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(int(signed)), rcp_p, ctypes.c_int(threads))
        if rc != 0:
            raise RuntimeError("orc_encode_bc6h rc=%d" % rc)
        return out.reshape(n, 16)

    def encode_etc2(self, blocks, options, mode, threads=1, alloc_options=None):
        """mode 0: ETC2 RGB (8 B), 1: ETC2 RGBA (16 B), 2: EAC alpha (8 B), 3: ETC1 (8 B), 4: ETC2 punch-through alpha (8 B).
        alloc_options: the Options handed to AllocETC2Data (default: the encode options)."""
        blocks, pb = _u8(blocks)
        n = blocks.size // 64
        assert n % 8 == 0
        per = 16 if mode == 1 else 8
        out = np.zeros(n * per, np.uint8)
        ao = options if alloc_options is None else alloc_options
        self.lib.orc_encode_etc2_alloc.restype = ctypes.c_int
        rc = self.lib.orc_encode_etc2_alloc(out.ctypes.data_as(ctypes.c_void_p), pb, ctypes.c_size_t(n),
                                            options.ctypes.data_as(ctypes.c_void_p), ao.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(mode), ctypes.c_int(threads))
        if rc != 0:
            raise RuntimeError("orc_encode_etc2 rc=%d" % rc)
        return out.reshape(n, per)

    def encode_s3tc(self, blocks, options, fmt, rcp=None, threads=1):
        blocks, pb = _u8(blocks)
        n = blocks.size // 64
        assert n % 8 == 0
        per = 8 if fmt in (4, 5) else 16
        out = np.zeros(n * per, np.uint8)
        rcp_p = None
        if rcp is not None:
            rcp = np.ascontiguousarray(rcp, np.float32)
            rcp_p = rcp.ctypes.data_as(ctypes.c_void_p)
        rc = self.lib.orc_encode_s3tc(out.ctypes.data_as(ctypes.c_void_p), pb, ctypes.c_size_t(n),
                                      options.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(fmt), rcp_p, ctypes.c_int(threads))
        if rc != 0:
            raise RuntimeError("orc_encode_s3tc rc=%d" % rc)
        return out.reshape(n, per)

    def encode_eac11(self, blocks_s16, signed=False):
        b = np.ascontiguousarray(blocks_s16, dtype=np.int16)
        n = b.size // 16
        assert n % 8 == 0
        out = np.zeros(n * 8, np.uint8)
        rc = self.lib.orc_encode_eac11(out.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n),
                                       ctypes.c_int(int(signed)))
        if rc != 0:
            raise RuntimeError("orc_encode_eac11 rc=%d" % rc)
        return out.reshape(n, 8)

    def encode_bc1(self, blocks, options, rcp=None, threads=1):
        blocks, pb = _u8(blocks)
        n = blocks.size // 64
        assert n % 8 == 0
        out = np.zeros(n * 8, np.uint8)
        rcp_p = None
        if rcp is not None:
            rcp = np.ascontiguousarray(rcp, np.float32)
            rcp_p = rcp.ctypes.data_as(ctypes.c_void_p)
        rc = self.lib.orc_encode_bc1(out.ctypes.data_as(ctypes.c_void_p), pb, ctypes.c_size_t(n),
                                     options.ctypes.data_as(ctypes.c_void_p), rcp_p, ctypes.c_int(threads))
        if rc != 0:
            raise RuntimeError("orc_encode_bc1 rc=%d" % rc)
        return out.reshape(n, 8)
