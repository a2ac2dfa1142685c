"""Python host-side mirror of the reference's public interface for the hot path
(reference ConvectionKernels.h:73-103 ``Options``, 142-199 ``BC7EncodingPlan``, 236-277
``cvtt::Kernels``), on top of the C ABI in ``include/cvtt_mi355x.h``.

PyTorch is plumbing only (device memory / streams); numpy arrays go through the library's
own pinned staging.  There is no CPU fallback: if the HIP library cannot be loaded or no
gfx950 device is present every encode raises ``CvttError``.
"""
import ctypes
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "lib", "libcvtt_mi355x.so")

NumParallelBlocks = 8  # reference ConvectionKernels.h:71


class Flags:
    """cvtt::Flags, reference ConvectionKernels.h:33-69."""
    BC7_FastIndexing = 0x008
    BC7_TrySingleColor = 0x010
    BC7_RespectPunchThrough = 0x020
    BC6H_FastIndexing = 0x040
    S3TC_Exhaustive = 0x080
    S3TC_Paranoid = 0x100
    Uniform = 0x200
    ETC_UseFakeBT709 = 0x400
    ETC_FakeBT709Accurate = 0x800
    Fastest = BC6H_FastIndexing | BC7_FastIndexing | S3TC_Paranoid
    Faster = Fastest
    Fast = BC7_FastIndexing | S3TC_Paranoid
    Default = BC7_FastIndexing | S3TC_Paranoid
    Better = S3TC_Paranoid | S3TC_Exhaustive
    Ultra = BC7_TrySingleColor | S3TC_Paranoid | S3TC_Exhaustive | ETC_FakeBT709Accurate


class Options(ctypes.Structure):
    """cvtt::Options (44 bytes)."""
    _fields_ = [
        ("flags", ctypes.c_uint32),
        ("threshold", ctypes.c_float),
        ("redWeight", ctypes.c_float),
        ("greenWeight", ctypes.c_float),
        ("blueWeight", ctypes.c_float),
        ("alphaWeight", ctypes.c_float),
        ("refineRoundsBC7", ctypes.c_int32),
        ("refineRoundsBC6H", ctypes.c_int32),
        ("refineRoundsIIC", ctypes.c_int32),
        ("refineRoundsS3TC", ctypes.c_int32),
        ("seedPoints", ctypes.c_int32),
    ]

    def __init__(self, **kw):
        super().__init__()
        f32 = np.float32
        self.flags = Flags.Default
        self.threshold = 0.5
        self.redWeight = float(f32(0.2125) / f32(0.7154))
        self.greenWeight = 1.0
        self.blueWeight = float(f32(0.0721) / f32(0.7154))
        self.alphaWeight = 1.0
        self.refineRoundsBC7 = 2
        self.refineRoundsBC6H = 3
        self.refineRoundsIIC = 8
        self.refineRoundsS3TC = 2
        self.seedPoints = 4
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError(k)
            setattr(self, k, v)

    def tobytes(self):
        return bytes(self)

    @classmethod
    def frombytes(cls, b):
        o = cls()
        ctypes.memmove(ctypes.addressof(o), bytes(b), ctypes.sizeof(cls))
        return o


class BC7EncodingPlan(ctypes.Structure):
    """cvtt::BC7EncodingPlan (808 bytes); default = every shape and partition, 4 seed points."""
    kNumRGBAShapes = 129
    kNumRGBShapes = 243
    _fields_ = [
        ("mode1PartitionEnabled", ctypes.c_uint64),
        ("mode2PartitionEnabled", ctypes.c_uint64),
        ("mode3PartitionEnabled", ctypes.c_uint64),
        ("mode0PartitionEnabled", ctypes.c_uint16),
        ("mode7RGBAPartitionEnabled", ctypes.c_uint64),
        ("mode7RGBPartitionEnabled", ctypes.c_uint64),
        ("mode4SP", (ctypes.c_uint8 * 2) * 4),
        ("mode5SP", ctypes.c_uint8 * 4),
        ("mode6Enabled", ctypes.c_uint8),
        ("seedPointsForShapeRGB", ctypes.c_uint8 * 243),
        ("seedPointsForShapeRGBA", ctypes.c_uint8 * 129),
        ("rgbaShapeList", ctypes.c_uint8 * 129),
        ("rgbaNumShapesToEvaluate", ctypes.c_uint8),
        ("rgbShapeList", ctypes.c_uint8 * 243),
        ("rgbNumShapesToEvaluate", ctypes.c_uint8),
    ]

    def __init__(self):
        super().__init__()
        full = 0xFFFFFFFFFFFFFFFF
        self.mode0PartitionEnabled = 0xFFFF
        self.mode1PartitionEnabled = full
        self.mode2PartitionEnabled = full
        self.mode3PartitionEnabled = full
        self.mode7RGBAPartitionEnabled = full
        self.mode7RGBPartitionEnabled = full
        self.mode6Enabled = 1
        for i in range(4):
            self.mode4SP[i][0] = 4
            self.mode4SP[i][1] = 4
            self.mode5SP[i] = 4
        for i in range(243):
            self.rgbShapeList[i] = i
            self.seedPointsForShapeRGB[i] = 4
        for i in range(129):
            self.rgbaShapeList[i] = i
            self.seedPointsForShapeRGBA[i] = 4
        self.rgbNumShapesToEvaluate = 243
        self.rgbaNumShapesToEvaluate = 129

    def tobytes(self):
        return bytes(self)

    @classmethod
    def frombytes(cls, b):
        o = cls()
        ctypes.memmove(ctypes.addressof(o), bytes(b), ctypes.sizeof(cls))
        return o


class BC7FineTuningParams(ctypes.Structure):
    """cvtt::BC7FineTuningParams (285 bytes): seed points (0 = off) per mode and partition / rotation / index selector;
    default = 4 everywhere."""
    _fields_ = [
        ("mode0SP", ctypes.c_uint8 * 16),
        ("mode1SP", ctypes.c_uint8 * 64),
        ("mode2SP", ctypes.c_uint8 * 64),
        ("mode3SP", ctypes.c_uint8 * 64),
        ("mode4SP", (ctypes.c_uint8 * 2) * 4),
        ("mode5SP", ctypes.c_uint8 * 4),
        ("mode6SP", ctypes.c_uint8),
        ("mode7SP", ctypes.c_uint8 * 64),
    ]

    def __init__(self):
        super().__init__()
        ctypes.memset(ctypes.addressof(self), 4, ctypes.sizeof(self))

    def tobytes(self):
        return bytes(self)

    @classmethod
    def frombytes(cls, b):
        o = cls()
        ctypes.memmove(ctypes.addressof(o), bytes(b), ctypes.sizeof(cls))
        return o


assert ctypes.sizeof(Options) == 44
assert ctypes.sizeof(BC7EncodingPlan) == 808
assert ctypes.sizeof(BC7FineTuningParams) == 285


class CvttError(RuntimeError):
    pass


_EXPORTS = (
    "cvttmi_source_sha256", "cvttmi_default_options", "cvttmi_default_bc7_plan", "cvttmi_create", "cvttmi_destroy",
    "cvttmi_last_error", "cvttmi_set_rcp_table", "cvttmi_get_rcp_table",
    "cvttmi_encode_bc7_device", "cvttmi_encode_bc7", "cvttmi_timing_enable", "cvttmi_timing_read",
    "cvttmi_set_exhaustive", "cvttmi_encode_bc1_device", "cvttmi_encode_bc1",
    "cvttmi_encode_bc6h_device", "cvttmi_encode_bc6h",
    "cvttmi_encode_etc2_device", "cvttmi_encode_etc2_rgba_device", "cvttmi_encode_etc2_alpha_device",
    "cvttmi_encode_etc2", "cvttmi_encode_etc2_rgba", "cvttmi_encode_etc2_alpha",
    "cvttmi_tiled_block_count", "cvttmi_tile_image_device", "cvttmi_compact_rows_device",
    "cvttmi_selftest_arith",
    "cvttmi_encode_etc2_alpha11_device", "cvttmi_encode_etc2_alpha11",
    "cvttmi_encode_bc2_device", "cvttmi_encode_bc3_device", "cvttmi_encode_bc4_device", "cvttmi_encode_bc5_device",
    "cvttmi_encode_bc2", "cvttmi_encode_bc3", "cvttmi_encode_bc4", "cvttmi_encode_bc5",
    "cvttmi_decode_bc7_device", "cvttmi_decode_bc7", "cvttmi_decode_bc6h_device", "cvttmi_decode_bc6h",
    "cvttmi_encode_etc1_device", "cvttmi_encode_etc1",
    "cvttmi_encode_etc2_punchthrough_alpha_device", "cvttmi_encode_etc2_punchthrough_alpha",
    "cvttmi_encode_etc2_with_data_device", "cvttmi_encode_etc2_with_data",
    "cvttmi_default_bc7_fine_tuning", "cvttmi_bc7_plan_from_quality", "cvttmi_bc7_plan_from_fine_tuning",
    "cvttmi_host_alloc", "cvttmi_host_free", "cvttmi_host_register", "cvttmi_host_unregister",
    "cvttmi_shard_block_rows", "cvttmi_multi_create", "cvttmi_multi_destroy", "cvttmi_multi_last_error", "cvttmi_multi_num_devices",
    "cvttmi_multi_context", "cvttmi_multi_last_shard", "cvttmi_multi_set_rcp_table", "cvttmi_multi_set_exhaustive", "cvttmi_multi_encode", "cvttmi_multi_encode_device",
    "cvttmi_encode_bc7_multi", "cvttmi_encode_bc1_multi", "cvttmi_encode_bc6h_multi", "cvttmi_encode_etc2_rgba_multi",
    "cvttmi_dropin_set_devices",
)

_lib = None


def _preload_hip_runtime():
    want = os.environ.get("CVTTMI_PRELOAD_HIP", "")
    if want == "0" or "torch" in sys.modules:
        return None
    cand = None
    if want:
        cand = want
    else:
        import importlib.util
        try:
            spec = importlib.util.find_spec("torch")  # locates the package without importing it
        except (ImportError, ValueError):
            spec = None
        if spec is not None and spec.submodule_search_locations:
            for d in spec.submodule_search_locations:
                c = os.path.join(d, "lib", "libamdhip64.so")
                if os.path.exists(c):
                    cand = c
                    break
    if not cand:
        return None
    try:
        return ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
    except OSError:
        return None


def load_library():
    """dlopen the HIP library (loudly fails when it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("CVTTMI_LIB", _LIB_PATH)  # override: A/B builds of the same library
    if not os.path.exists(path):
        raise CvttError("%s is missing: build it with `make -C convectionkernels_amd/csrc` "
                        "(or __graft_entry__.build())" % path)
    # One HIP runtime per process.  The library links against libamdhip64 by soname; PyTorch-ROCm ships a libamdhip64 of its own.
    # Loaded in the order library -> torch, the process ends up with two runtimes and the library's finds no device any more
    # (cvttmi_create = CVTTMI_E_NO_DEVICE; observed on the MI355X box, round 5).  The dynamic loader binds a soname once per
    # process, so it is enough that torch's runtime is mapped (RTLD_GLOBAL) BEFORE this library: then both use that one.
    #   * torch already imported: its runtime is there, nothing to do;
    #   * torch installed but not imported (numpy-only callers): preload just its libamdhip64.so -- no `import torch`, no
    #     multi-second start-up, no GPU initialisation -- so that a later `import torch` meets the runtime it expects;
    #   * no torch: the system runtime, as for any C / C++ caller (INTEGRATION.md "One HIP runtime per process").
    # CVTTMI_PRELOAD_HIP=<path> names the runtime to preload explicitly; CVTTMI_PRELOAD_HIP=0 turns the preload off.
    _preload_hip_runtime()
    lib = ctypes.CDLL(path)
    variant = "CVTTMI_LIB" in os.environ  # a developer's A/B library (tools/ab_*.sh) may predate the newest entry points
    for name in _EXPORTS:
        if not hasattr(lib, name) and not (variant and name == "cvttmi_source_sha256"):
            raise CvttError("symbol %s missing from %s" % (name, path))
    if hasattr(lib, "cvttmi_source_sha256"):
        lib.cvttmi_source_sha256.restype = ctypes.c_char_p
        lib.cvttmi_source_sha256.argtypes = []
    lib.cvttmi_last_error.restype = ctypes.c_char_p
    lib.cvttmi_last_error.argtypes = [ctypes.c_void_p]
    lib.cvttmi_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]
    lib.cvttmi_destroy.argtypes = [ctypes.c_void_p]
    lib.cvttmi_set_rcp_table.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.cvttmi_get_rcp_table.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.cvttmi_encode_bc7_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.cvttmi_encode_bc7.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                      ctypes.c_void_p, ctypes.c_void_p]
    lib.cvttmi_encode_bc1_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                             ctypes.c_void_p, ctypes.c_void_p]
    lib.cvttmi_encode_bc1.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.cvttmi_encode_bc6h_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                              ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    lib.cvttmi_encode_bc6h.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                       ctypes.c_void_p, ctypes.c_int]
    for n in ("cvttmi_encode_etc2", "cvttmi_encode_etc2_rgba", "cvttmi_encode_etc2_alpha", "cvttmi_encode_etc1",
              "cvttmi_encode_etc2_punchthrough_alpha"):
        getattr(lib, n).argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        getattr(lib, n + "_device").argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                                 ctypes.c_void_p, ctypes.c_void_p]
    lib.cvttmi_encode_etc2_with_data.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    lib.cvttmi_encode_etc2_with_data_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p,
                                                        ctypes.c_int, ctypes.c_void_p]
    lib.cvttmi_tiled_block_count.restype = ctypes.c_size_t
    lib.cvttmi_tiled_block_count.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
    lib.cvttmi_tile_image_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32,
                                             ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    lib.cvttmi_compact_rows_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32,
                                               ctypes.c_uint32, ctypes.c_void_p]
    for n in ("cvttmi_encode_bc2", "cvttmi_encode_bc3"):
        getattr(lib, n).argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        getattr(lib, n + "_device").argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
    for n in ("cvttmi_encode_bc4", "cvttmi_encode_bc5"):
        getattr(lib, n).argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int]
        getattr(lib, n + "_device").argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int,
                                                 ctypes.c_void_p]
    lib.cvttmi_encode_etc2_alpha11_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                                      ctypes.c_void_p, ctypes.c_void_p]
    lib.cvttmi_encode_etc2_alpha11.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    lib.cvttmi_decode_bc7_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.cvttmi_decode_bc7.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    lib.cvttmi_decode_bc6h_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    lib.cvttmi_decode_bc6h.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    lib.cvttmi_selftest_arith.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64),
                                          ctypes.POINTER(ctypes.c_uint64)]
    lib.cvttmi_bc7_plan_from_quality.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.cvttmi_bc7_plan_from_fine_tuning.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.cvttmi_default_bc7_fine_tuning.argtypes = [ctypes.c_void_p]
    lib.cvttmi_host_alloc.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    lib.cvttmi_host_free.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.cvttmi_host_register.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    lib.cvttmi_host_unregister.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.cvttmi_timing_enable.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.cvttmi_set_exhaustive.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.cvttmi_timing_read.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64)]
    if hasattr(lib, "cvttmi_multi_create"):
        lib.cvttmi_shard_block_rows.argtypes = [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                                ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
        lib.cvttmi_multi_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_int]
        lib.cvttmi_multi_destroy.argtypes = [ctypes.c_void_p]
        lib.cvttmi_multi_last_error.restype = ctypes.c_char_p
        lib.cvttmi_multi_last_error.argtypes = [ctypes.c_void_p]
        lib.cvttmi_multi_num_devices.argtypes = [ctypes.c_void_p]
        lib.cvttmi_multi_last_shard.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
        lib.cvttmi_multi_set_rcp_table.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.cvttmi_multi_set_exhaustive.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.cvttmi_multi_encode_device.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t,
                                                   ctypes.c_void_p, ctypes.c_void_p]
        lib.cvttmi_multi_encode.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t,
                                            ctypes.c_void_p, ctypes.c_void_p]
        lib.cvttmi_encode_bc7_multi.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t,
                                                ctypes.c_void_p, ctypes.c_void_p]
        for n in ("cvttmi_encode_bc1_multi", "cvttmi_encode_etc2_rgba_multi"):
            getattr(lib, n).argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p]
        lib.cvttmi_encode_bc6h_multi.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t,
                                                 ctypes.c_void_p, ctypes.c_int]
        lib.cvttmi_dropin_set_devices.argtypes = [ctypes.c_void_p, ctypes.c_int]
    _lib = lib
    return lib


def library_source_sha256():
    """Build identity compiled into the loaded library (csrc/Makefile: SHA-256 over the kernel and shim sources, the public
    headers and the compiler flags).  The same for every build of the same tree, whatever the build directory; profiles/
    summaries carry it and bench.py quotes their counters only when it matches."""
    lib = load_library()
    return lib.cvttmi_source_sha256().decode() if hasattr(lib, "cvttmi_source_sha256") else ""


def library_fatbin_sha256(path=None):
    """SHA-256 of the device code (.hip_fatbin section) of the library that load_library() uses: identifies the kernel
    objects a profile was taken with (bench.py only quotes profiles/rNN/summary.json counters when it matches)."""
    import hashlib
    import struct
    path = path or os.environ.get("CVTTMI_LIB", _LIB_PATH)
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"\x7fELF" or data[4] != 2:
        return hashlib.sha256(data).hexdigest()
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
    secs = [struct.unpack_from("<IIQQQQIIQQ", data, shoff + i * shentsize) for i in range(shnum)]
    stroff = secs[shstrndx][4]
    for name_off, _type, _flags, _addr, off, size, *_ in secs:
        end = data.index(b"\0", stroff + name_off)
        if data[stroff + name_off:end] == b".hip_fatbin":
            return hashlib.sha256(data[off:off + size]).hexdigest()
    return hashlib.sha256(data).hexdigest()


def exported_symbols():
    return _EXPORTS


class Context:
    """One encoder context on one HIP device (tables in HBM, staging and work buffers, rcp table).

    Concurrency: a context owns ONE set of device work buffers (the BC7 hand-over list and punch-through table, the host-path
    staging ring).  The calls that use them are ordered by the library itself when they arrive on
    different streams (cvtt_mi355x.h "Streams"; calls that use no shared work space are simply queued on their stream), the
    host side of every call on a context is serialised by a mutex in the library, and the host-pointer calls additionally by
    a lock here, so sharing a context between threads or streams is safe but not concurrent -- use one Context per worker
    thread (they are cheap: ~10 KB of tables + the work buffers) to overlap independent jobs."""

    def __init__(self, device=0):
        import threading
        self._host_lock = threading.Lock()
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        rc = self._lib.cvttmi_create(ctypes.byref(self._h), int(device))
        if rc != 0:
            raise CvttError("cvttmi_create(device=%d) failed with %d (no gfx950 device?)" % (device, rc))
        self.device = int(device)

    def close(self):
        if self._h:
            self._lib.cvttmi_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            msg = self._lib.cvttmi_last_error(self._h)
            raise CvttError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))

    # -- argument checks shared by every entry point: a wrong buffer must raise here, never reach a kernel or memcpy --
    @staticmethod
    def _host_out(out, n, out_bytes, dtype=np.uint8, shape=None):
        """the caller's numpy result buffer, or a fresh one"""
        if out is None:
            return np.empty(shape if shape is not None else (n, out_bytes), dtype)
        need = n * out_bytes
        if not (isinstance(out, np.ndarray) and out.flags["C_CONTIGUOUS"] and out.flags["WRITEABLE"] and out.nbytes == need
                and out.dtype.itemsize == np.dtype(dtype).itemsize):
            raise CvttError("out must be a writable C-contiguous numpy array of exactly %d bytes (%d blocks x %d), %d-byte elements"
                            % (need, n, out_bytes, np.dtype(dtype).itemsize))
        return out

    def _device_in(self, t, what="blocks"):
        import torch
        if not (isinstance(t, torch.Tensor) and t.is_cuda):
            raise CvttError("%s must be a numpy array or a CUDA tensor (got %s)" % (what, type(t).__name__ if not isinstance(t, torch.Tensor) else "a CPU tensor"))
        if t.device.index != self.device:
            raise CvttError("%s lives on cuda:%d but this context was created for cuda:%d" % (what, t.device.index, self.device))
        return t.contiguous()

    def _device_out(self, out, n, out_bytes, like, dtype=None, shape=None):
        """the caller's CUDA result tensor, or a fresh one on the input's device"""
        import torch
        dtype = dtype or torch.uint8
        if out is None:
            return torch.empty(shape if shape is not None else (n, out_bytes), dtype=dtype, device=like.device)
        need = n * out_bytes
        if not (isinstance(out, torch.Tensor) and out.is_cuda and out.device == like.device and out.is_contiguous()
                and out.numel() * out.element_size() == need and out.element_size() == torch.empty((), dtype=dtype).element_size()):
            raise CvttError("out must be a contiguous CUDA tensor on %s of exactly %d bytes (%d blocks x %d)" % (like.device, need, n, out_bytes))
        return out

    def _stream(self, stream, device):
        import torch
        return torch.cuda.current_stream(device).cuda_stream if stream is None else stream

    # -- reciprocal table (host RCPPS probe; see include/cvtt_mi355x.h) --
    def get_rcp_table(self):
        out = np.zeros(17, np.float32)
        self._check(self._lib.cvttmi_get_rcp_table(self._h, out.ctypes.data), "get_rcp_table")
        return out

    def set_rcp_table(self, lut):
        lut = np.ascontiguousarray(lut, np.float32)
        assert lut.size == 17
        self._check(self._lib.cvttmi_set_rcp_table(self._h, lut.ctypes.data), "set_rcp_table")

    # -- page-locked host memory for the host-pointer (numpy) entry points --
    def host_empty(self, shape, dtype=np.uint8):
        """numpy array in page-locked memory (cvttmi_host_alloc): the numpy entry points move it over PCIe in place,
        pipelined with the search, instead of staging it through a copy.  Freed when the array and its views are gone."""
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        p = ctypes.c_void_p()
        self._check(self._lib.cvttmi_host_alloc(self._h, ctypes.byref(p), nbytes), "host_alloc")
        lib, addr = self._lib, p.value

        class _Owner:
            def __del__(self_inner):
                # without the context: the array may outlive it (cvttmi_host_free accepts NULL)
                try:
                    lib.cvttmi_host_free(None, ctypes.c_void_p(addr))
                except Exception:  # noqa
                    pass
        buf = (ctypes.c_uint8 * max(1, nbytes)).from_address(addr)
        buf._owner = _Owner()
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def host_register(self, array):
        """page-lock an existing C-contiguous numpy array in place (undo with host_unregister before freeing it)"""
        a = np.ascontiguousarray(array)
        if a.ctypes.data != array.ctypes.data:
            raise CvttError("array must be C-contiguous")
        self._check(self._lib.cvttmi_host_register(self._h, a.ctypes.data, a.nbytes), "host_register")

    def host_unregister(self, array):
        self._check(self._lib.cvttmi_host_unregister(self._h, array.ctypes.data), "host_unregister")

    def set_exhaustive(self, on=True):
        """search every candidate like the reference (default: exact branch-and-bound pruning)"""
        self._check(self._lib.cvttmi_set_exhaustive(self._h, 1 if on else 0), "set_exhaustive")

    # -- timing of the kernels launched by *_device calls (HIP events on the launch stream) --
    def timing_enable(self, on=True):
        self._check(self._lib.cvttmi_timing_enable(self._h, 1 if on else 0), "timing_enable")

    def timing_read(self):
        ms = ctypes.c_double()
        n = ctypes.c_uint64()
        self._check(self._lib.cvttmi_timing_read(self._h, ctypes.byref(ms), ctypes.byref(n)), "timing_read")
        return ms.value, n.value

    # -- BC7 --
    def encode_bc7(self, blocks, options=None, plan=None, out=None, stream=None):
        """Batched cvtt::Kernels::EncodeBC7.  ``blocks``: (N,16,4) uint8 numpy array (host
        path) or a torch uint8 tensor on this context's GPU (device path, asynchronous on
        ``stream`` / the current torch stream).  N must be a multiple of 8."""
        options = options if options is not None else Options()
        plan = plan if plan is not None else BC7EncodingPlan()
        if isinstance(blocks, np.ndarray):
            b = np.ascontiguousarray(blocks, np.uint8)
            n = b.size // 64
            if b.size % 64 or n % NumParallelBlocks:
                raise CvttError("blocks must hold a multiple of 8 PixelBlockU8")
            res = self._host_out(out, n, 16)
            with self._host_lock:
                self._check(self._lib.cvttmi_encode_bc7(self._h, res.ctypes.data, b.ctypes.data, n,
                                                        ctypes.addressof(options), ctypes.addressof(plan)), "encode_bc7")
            return res
        import torch
        b = self._device_in(blocks)
        if b.dtype != torch.uint8:
            raise CvttError("blocks must be a numpy uint8 array or a CUDA uint8 tensor")
        n = b.numel() // 64
        if b.numel() % 64 or n % NumParallelBlocks:
            raise CvttError("blocks must hold a multiple of 8 PixelBlockU8")
        res = self._device_out(out, n, 16, b)
        stream = self._stream(stream, b.device)
        self._check(self._lib.cvttmi_encode_bc7_device(self._h, res.data_ptr(), b.data_ptr(), n,
                                                       ctypes.addressof(options), ctypes.addressof(plan),
                                                       ctypes.c_void_p(stream)), "encode_bc7_device")
        return res


    # -- generic helper for the formats without a plan argument --
    def _encode_simple(self, host_fn, dev_fn, what, blocks, options, out, stream, in_bytes, out_bytes):
        options = options if options is not None else Options()
        if isinstance(blocks, np.ndarray):
            b = np.ascontiguousarray(blocks)
            n = b.nbytes // in_bytes
            if b.nbytes % in_bytes or n % NumParallelBlocks:
                raise CvttError("blocks must hold a multiple of 8 pixel blocks")
            res = self._host_out(out, n, out_bytes)
            with self._host_lock:
                self._check(host_fn(self._h, res.ctypes.data, b.ctypes.data, n, ctypes.addressof(options)), what)
            return res
        b = self._device_in(blocks)
        nbytes = b.numel() * b.element_size()
        n = nbytes // in_bytes
        if nbytes % in_bytes or n % NumParallelBlocks:
            raise CvttError("blocks must hold a multiple of 8 pixel blocks")
        res = self._device_out(out, n, out_bytes, b)
        stream = self._stream(stream, b.device)
        self._check(dev_fn(self._h, res.data_ptr(), b.data_ptr(), n, ctypes.addressof(options), ctypes.c_void_p(stream)), what)
        return res

    # -- BC1 --
    def encode_bc1(self, blocks, options=None, out=None, stream=None):
        """Batched cvtt::Kernels::EncodeBC1: (N,16,4) uint8 -> (N,8) uint8."""
        return self._encode_simple(self._lib.cvttmi_encode_bc1, self._lib.cvttmi_encode_bc1_device, "encode_bc1",
                                   blocks, options, out, stream, 64, 8)


    # -- BC2 / BC3 / BC4 / BC5 --
    def encode_bc2(self, blocks, options=None, out=None, stream=None):
        """Batched cvtt::Kernels::EncodeBC2: (N,16,4) uint8 -> (N,16) uint8 [explicit alpha | colour]."""
        return self._encode_simple(self._lib.cvttmi_encode_bc2, self._lib.cvttmi_encode_bc2_device, "encode_bc2", blocks, options, out, stream, 64, 16)

    def encode_bc3(self, blocks, options=None, out=None, stream=None):
        """Batched cvtt::Kernels::EncodeBC3: (N,16,4) uint8 -> (N,16) uint8 [interpolated alpha | colour]."""
        return self._encode_simple(self._lib.cvttmi_encode_bc3, self._lib.cvttmi_encode_bc3_device, "encode_bc3", blocks, options, out, stream, 64, 16)

    def encode_bc4(self, blocks, options=None, signed=False, out=None, stream=None):
        """Batched cvtt::Kernels::EncodeBC4U / EncodeBC4S (red channel; signed: int8 PixelBlockS8): -> (N,8) uint8."""
        sg = 1 if signed else 0
        host = lambda h, o, b, n, opt: self._lib.cvttmi_encode_bc4(h, o, b, n, opt, sg)
        dev = lambda h, o, b, n, opt, st: self._lib.cvttmi_encode_bc4_device(h, o, b, n, opt, sg, st)
        if isinstance(blocks, np.ndarray) and blocks.dtype == np.int8:
            blocks = blocks.view(np.uint8)
        return self._encode_simple(host, dev, "encode_bc4", blocks, options, out, stream, 64, 8)

    def encode_bc5(self, blocks, options=None, signed=False, out=None, stream=None):
        """Batched cvtt::Kernels::EncodeBC5U / EncodeBC5S (red, green): -> (N,16) uint8."""
        sg = 1 if signed else 0
        host = lambda h, o, b, n, opt: self._lib.cvttmi_encode_bc5(h, o, b, n, opt, sg)
        dev = lambda h, o, b, n, opt, st: self._lib.cvttmi_encode_bc5_device(h, o, b, n, opt, sg, st)
        if isinstance(blocks, np.ndarray) and blocks.dtype == np.int8:
            blocks = blocks.view(np.uint8)
        return self._encode_simple(host, dev, "encode_bc5", blocks, options, out, stream, 64, 16)

    # -- BC6H --
    def encode_bc6h(self, blocks, options=None, signed=False, out=None, stream=None):
        """Batched cvtt::Kernels::EncodeBC6HU / EncodeBC6HS: (N,16,4) half bit patterns (int16 /
        uint16 / float16 numpy array, or a CUDA tensor of 2-byte elements) -> (N,16) uint8."""
        sg = 1 if signed else 0
        host = lambda h, o, b, n, opt: self._lib.cvttmi_encode_bc6h(h, o, b, n, opt, sg)
        dev = lambda h, o, b, n, opt, st: self._lib.cvttmi_encode_bc6h_device(h, o, b, n, opt, sg, st)
        return self._encode_simple(host, dev, "encode_bc6h", blocks, options, out, stream, 128, 16)


    # -- ETC1 / ETC2 --
    def encode_etc1(self, blocks, options=None, out=None, stream=None):
        """Batched cvtt::Kernels::EncodeETC1: (N,16,4) uint8 -> (N,8) uint8."""
        return self._encode_simple(self._lib.cvttmi_encode_etc1, self._lib.cvttmi_encode_etc1_device, "encode_etc1",
                                   blocks, options, out, stream, 64, 8)

    def _encode_etc2_kind(self, kind, what, blocks, options, out, stream, out_bytes, compression_data):
        """The three calls that take the reference's ETC2CompressionData: the chroma axes of the sector split belong to the
        Options AllocETC2Data was called with (reference ConvectionKernels_ETC.cpp:3117-3145), the rest to `options`."""
        ao = None
        if compression_data is not None:
            ao = compression_data.alloc_options if isinstance(compression_data, ETC2CompressionData) else compression_data
        aop = ctypes.byref(ao) if ao is not None else None
        host = lambda h, o, b, n, opt: self._lib.cvttmi_encode_etc2_with_data(h, o, b, n, opt, aop, kind)
        dev = lambda h, o, b, n, opt, st: self._lib.cvttmi_encode_etc2_with_data_device(h, o, b, n, opt, aop, kind, st)
        return self._encode_simple(host, dev, what, blocks, options, out, stream, 64, out_bytes)

    def encode_etc2(self, blocks, options=None, out=None, stream=None, compression_data=None):
        """Batched cvtt::Kernels::EncodeETC2 (RGB): (N,16,4) uint8 -> (N,8) uint8.  compression_data: what AllocETC2Data
        returned (or the Options it was given); None = allocated with `options`."""
        return self._encode_etc2_kind(0, "encode_etc2", blocks, options, out, stream, 8, compression_data)

    def encode_etc2_punchthrough_alpha(self, blocks, options=None, out=None, stream=None, compression_data=None):
        """Batched cvtt::Kernels::EncodeETC2PunchthroughAlpha: (N,16,4) uint8 -> (N,8) uint8 (RGB8A1 blocks; a pixel is
        transparent when its alpha is below floor(clamp(options.threshold, 0, 1) * 255 + 1))."""
        return self._encode_etc2_kind(4, "encode_etc2_punchthrough_alpha", blocks, options, out, stream, 8, compression_data)

    def encode_etc2_rgba(self, blocks, options=None, out=None, stream=None, compression_data=None):
        """Batched cvtt::Kernels::EncodeETC2RGBA: (N,16,4) uint8 -> (N,16) uint8 = [EAC alpha | colour]."""
        return self._encode_etc2_kind(1, "encode_etc2_rgba", blocks, options, out, stream, 16, compression_data)

    def encode_etc2_alpha(self, blocks, options=None, out=None, stream=None):
        """Batched cvtt::Kernels::EncodeETC2Alpha (EAC 8-bit): (N,16,4) uint8 -> (N,8) uint8."""
        return self._encode_simple(self._lib.cvttmi_encode_etc2_alpha, self._lib.cvttmi_encode_etc2_alpha_device,
                                   "encode_etc2_alpha", blocks, options, out, stream, 64, 8)


    def encode_etc2_alpha11(self, blocks, signed=False, options=None, out=None, stream=None):
        """Batched cvtt::Kernels::EncodeETC2Alpha11 (EAC R11): (N,16) int16 PixelBlockScalarS16 -> (N,8) uint8."""
        sg = 1 if signed else 0
        host = lambda h, o, b, n, opt: self._lib.cvttmi_encode_etc2_alpha11(h, o, b, n, sg, opt)
        dev = lambda h, o, b, n, opt, st: self._lib.cvttmi_encode_etc2_alpha11_device(h, o, b, n, sg, opt, st)
        if isinstance(blocks, np.ndarray):
            blocks = np.ascontiguousarray(blocks, np.int16)
        return self._encode_simple(host, dev, "encode_etc2_alpha11", blocks, options, out, stream, 32, 8)

    # -- decoders (cvtt::Kernels::DecodeBC7 / DecodeBC6HU / DecodeBC6HS) --
    def _decode(self, packed, fmt, stream):
        hdr = fmt != "bc7"
        sg = 1 if fmt == "bc6hs" else 0
        if isinstance(packed, np.ndarray):
            b = np.ascontiguousarray(packed, np.uint8)
            n = b.size // 16
            if b.size % 16 or n % NumParallelBlocks:
                raise CvttError("packed must hold a multiple of 8 16-byte blocks")
            res = np.empty((n, 16, 4), np.int16 if hdr else np.uint8)
            with self._host_lock:
                rc = (self._lib.cvttmi_decode_bc6h(self._h, res.ctypes.data, b.ctypes.data, n, sg) if hdr
                      else self._lib.cvttmi_decode_bc7(self._h, res.ctypes.data, b.ctypes.data, n))
            self._check(rc, "decode")
            return res
        import torch
        b = self._device_in(packed, "packed")
        nbytes = b.numel() * b.element_size()
        n = nbytes // 16
        if nbytes % 16 or n % NumParallelBlocks:
            raise CvttError("packed must hold a multiple of 8 16-byte blocks")
        res = torch.empty((n, 16, 4), dtype=torch.int16 if hdr else torch.uint8, device=b.device)
        stream = self._stream(stream, b.device)
        rc = (self._lib.cvttmi_decode_bc6h_device(self._h, res.data_ptr(), b.data_ptr(), n, sg, ctypes.c_void_p(stream)) if hdr
              else self._lib.cvttmi_decode_bc7_device(self._h, res.data_ptr(), b.data_ptr(), n, ctypes.c_void_p(stream)))
        self._check(rc, "decode")
        return res

    def decode_bc7(self, packed, stream=None):
        """(N,16) uint8 packed BC7 blocks (numpy or CUDA tensor) -> (N,16,4) uint8 PixelBlockU8"""
        return self._decode(packed, "bc7", stream)

    def decode_bc6h(self, packed, signed=False, stream=None):
        """(N,16) uint8 packed BC6H blocks -> (N,16,4) int16 half bit patterns (alpha = 0x3C00)"""
        return self._decode(packed, "bc6hs" if signed else "bc6hu", stream)

    def psnr_bc7(self, blocks, packed):
        """PSNR (dB, over the four channels) of the decoded `packed` blocks against the source PixelBlockU8 tensor, on the device"""
        import torch
        d = self.decode_bc7(packed).to(torch.float32) - blocks.reshape(-1, 16, 4).to(torch.float32)
        mse = float((d * d).mean().item())
        return float("inf") if mse == 0.0 else 10.0 * float(np.log10(255.0 * 255.0 / mse))

    def selftest_arith(self, count=1 << 22, seed=1):
        """(divide mismatches, sqrt mismatches) of the device against the host's IEEE results"""
        d, q = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._check(self._lib.cvttmi_selftest_arith(self._h, count, seed, ctypes.byref(d), ctypes.byref(q)), "selftest_arith")
        return int(d.value), int(q.value)

    # -- image -> PixelBlock tiling on the device (reference etc2packer.cpp:215-247, 275-281) --
    def tile_image(self, image, stream=None):
        """(H,W,4) CUDA tensor, uint8 (RGBA8) or a 2-byte dtype (RGBA16F bit patterns), rows may be
        strided -> (ceil(H/4) * ceil(ceil(W/4)/8)*8, 16, 4) PixelBlock tensor: groups of eight
        horizontally adjacent blocks, reads clamped at the right / bottom edge."""
        import torch
        if not (isinstance(image, torch.Tensor) and image.is_cuda and image.dim() == 3 and image.shape[2] == 4):
            raise CvttError("image must be a CUDA tensor of shape (H, W, 4)")
        if image.device.index != self.device:
            raise CvttError("image lives on cuda:%d but this context was created for cuda:%d" % (image.device.index, self.device))
        if image.stride(2) != 1 or image.stride(1) != 4:
            image = image.contiguous()
        h, w = int(image.shape[0]), int(image.shape[1])
        esz = image.element_size()
        if esz not in (1, 2):
            raise CvttError("image must be RGBA8 or RGBA16F")
        n = self._lib.cvttmi_tiled_block_count(w, h)
        blocks = torch.empty((n, 16, 4), dtype=image.dtype, device=image.device)
        if stream is None:
            stream = torch.cuda.current_stream(image.device).cuda_stream
        self._check(self._lib.cvttmi_tile_image_device(self._h, blocks.data_ptr(), image.data_ptr(), w, h,
                                                       image.stride(0) * esz, 0 if esz == 1 else 1, ctypes.c_void_p(stream)), "tile_image")
        return blocks

    def compact_rows(self, packed, width, height, stream=None):
        """drop the blocks that only pad the last group of every block row: (padded N, B) -> (ceil(H/4)*ceil(W/4), B)"""
        import torch
        packed = self._device_in(packed, "packed")
        bpb = int(packed.shape[1])
        out = torch.empty((((height + 3) // 4) * ((width + 3) // 4), bpb), dtype=torch.uint8, device=packed.device)
        if stream is None:
            stream = torch.cuda.current_stream(packed.device).cuda_stream
        self._check(self._lib.cvttmi_compact_rows_device(self._h, out.data_ptr(), packed.contiguous().data_ptr(), width, height, bpb,
                                                         ctypes.c_void_p(stream)), "compact_rows")
        return out

    def encode_image(self, fmt, image, options=None, plan=None, stream=None):
        """linear image in HBM -> packed blocks of format `fmt`, row-major, ceil(W/4) blocks per row: tiling, encode and
        row compaction all on the device.  fmt: "bc7", "bc1", "bc2", "bc3", "bc4u", "bc4s", "bc5u", "bc5s", "etc1",
        "etc2" (= "etc2rgb"), "etc2rgba", "etc2punchthrough", "eac" (8-bit EAC alpha block alone) from an (H,W,4) uint8 image;
        "bc6hu", "bc6hs" from an (H,W,4) half-float image."""
        h, w = int(image.shape[0]), int(image.shape[1])
        blocks = self.tile_image(image, stream)
        simple = {"bc1": self.encode_bc1, "bc2": self.encode_bc2, "bc3": self.encode_bc3, "etc1": self.encode_etc1,
                  "etc2": self.encode_etc2, "etc2rgb": self.encode_etc2, "etc2rgba": self.encode_etc2_rgba, "eac": self.encode_etc2_alpha,
                  "etc2punchthrough": self.encode_etc2_punchthrough_alpha}
        if fmt == "bc7":
            packed = self.encode_bc7(blocks, options, plan, stream=stream)
        elif fmt in simple:
            packed = simple[fmt](blocks, options, stream=stream)
        elif fmt in ("bc4u", "bc4s"):
            packed = self.encode_bc4(blocks, options, signed=(fmt == "bc4s"), stream=stream)
        elif fmt in ("bc5u", "bc5s"):
            packed = self.encode_bc5(blocks, options, signed=(fmt == "bc5s"), stream=stream)
        elif fmt in ("bc6hu", "bc6hs"):
            packed = self.encode_bc6h(blocks, options, signed=(fmt == "bc6hs"), stream=stream)
        else:
            raise CvttError("unknown format %r" % (fmt,))
        return packed if w % 32 == 0 else self.compact_rows(packed, w, h, stream)


_default_ctx = {}


def shard_block_rows(block_rows, blocks_per_row, rank, world):
    """cvttmi_shard_block_rows: [first, last) of `rank`'s shard -- whole block rows, group aligned (the C statement of
    sharding.shard_block_rows; needs no device)"""
    lib = load_library()
    lo, hi = ctypes.c_size_t(), ctypes.c_size_t()
    rc = lib.cvttmi_shard_block_rows(block_rows, blocks_per_row, rank, world, ctypes.byref(lo), ctypes.byref(hi))
    if rc != 0:
        raise CvttError("cvttmi_shard_block_rows failed (%d)" % rc)
    return lo.value, hi.value


class MultiContext:
    """One job on several devices behind the C ABI (cvttmi_multi_*, csrc/multi.cpp): one context per entry of `devices` (a
    device may appear more than once), block-row shards, packed output straight into the caller's buffer.  Host arrays only."""
    FORMATS = {"bc7": (0, 64, 16), "bc1": (1, 64, 8), "bc6hu": (2, 128, 16), "bc6hs": (3, 128, 16), "etc2": (4, 64, 8), "etc2rgba": (5, 64, 16)}

    def __init__(self, devices):
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        self.devices = [int(d) for d in devices]
        arr = (ctypes.c_int * len(self.devices))(*self.devices)
        rc = self._lib.cvttmi_multi_create(ctypes.byref(self._h), arr, len(self.devices))
        if rc != 0:
            raise CvttError("cvttmi_multi_create(%s) failed with %d" % (self.devices, rc))

    def close(self):
        if self._h:
            self._lib.cvttmi_multi_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_rcp_table(self, lut):
        lut = np.ascontiguousarray(lut, np.float32)
        assert lut.size == 17
        rc = self._lib.cvttmi_multi_set_rcp_table(self._h, lut.ctypes.data)
        if rc != 0:
            raise CvttError("cvttmi_multi_set_rcp_table failed (%d)" % rc)

    def set_exhaustive(self, on=True):
        self._lib.cvttmi_multi_set_exhaustive(self._h, 1 if on else 0)

    def last_shards(self):
        out = []
        for i in range(len(self.devices)):
            lo, hi = ctypes.c_size_t(), ctypes.c_size_t()
            self._lib.cvttmi_multi_last_shard(self._h, i, ctypes.byref(lo), ctypes.byref(hi))
            out.append((lo.value, hi.value))
        return out

    def encode(self, fmt, blocks, options=None, plan=None, blocks_per_row=0, out=None):
        code, in_bytes, out_bytes = self.FORMATS[fmt]
        options = options if options is not None else Options()
        if fmt == "bc7" and plan is None:
            plan = BC7EncodingPlan()
        b = np.ascontiguousarray(blocks)
        n = b.nbytes // in_bytes
        if b.nbytes % in_bytes or n % NumParallelBlocks:
            raise CvttError("blocks must hold a multiple of 8 PixelBlocks")
        res = out if out is not None else np.empty((n, out_bytes), np.uint8)
        assert res.nbytes == n * out_bytes and res.flags["C_CONTIGUOUS"]
        rc = self._lib.cvttmi_multi_encode(self._h, code, res.ctypes.data, b.ctypes.data, n, blocks_per_row,
                                           ctypes.addressof(options), ctypes.addressof(plan) if plan is not None else None)
        if rc != 0:
            raise CvttError("cvttmi_multi_encode(%s) failed (%d): %s" % (fmt, rc, self._lib.cvttmi_multi_last_error(self._h).decode()))
        return res


    def encode_device(self, fmt, shards, out, blocks_per_row=0, options=None, plan=None):
        """cvttmi_multi_encode_device: `shards[r]` = a device tensor on devices[r] holding shard r's PixelBlocks (None for an empty
        shard), `out` = a device tensor on devices[0] for all the packed blocks.  The shard table is that of
        shard_block_rows(rows, blocks_per_row, r, len(devices)); returns `out` when it is complete."""
        code, in_bytes, out_bytes = self.FORMATS[fmt]
        options = options if options is not None else Options()
        if fmt == "bc7" and plan is None:
            plan = BC7EncodingPlan()
        n = out.numel() * out.element_size() // out_bytes
        if len(shards) != len(self.devices):
            raise CvttError("one shard (or None) per device")
        ptrs = (ctypes.c_void_p * len(shards))(*[None if t is None else t.data_ptr() for t in shards])
        import torch
        for t in shards:
            if t is not None:
                torch.cuda.current_stream(t.device).synchronize()  # the library launches on streams of its own
        rc = self._lib.cvttmi_multi_encode_device(self._h, code, out.data_ptr(), ptrs, n, blocks_per_row,
                                                  ctypes.addressof(options), ctypes.addressof(plan) if plan is not None else None)
        if rc != 0:
            raise CvttError("cvttmi_multi_encode_device(%s) failed (%d): %s" % (fmt, rc, self._lib.cvttmi_multi_last_error(self._h).decode()))
        return out


def default_context(device=0):
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


# ---- cvtt::Kernels-style free functions (8-block call convention of the reference) ----
def ConfigureBC7EncodingPlanFromQuality(encodingPlan, quality):
    """cvtt::Kernels::ConfigureBC7EncodingPlanFromQuality (BC67.cpp:3291-3352): fills `encodingPlan` in place for quality
    1..100 (clamped).  Host-side; needs no device."""
    if load_library().cvttmi_bc7_plan_from_quality(ctypes.byref(encodingPlan), int(quality)) != 0:
        raise CvttError("cvttmi_bc7_plan_from_quality failed")
    return encodingPlan


def ConfigureBC7EncodingPlanFromFineTuningParams(encodingPlan, params):
    """cvtt::Kernels::ConfigureBC7EncodingPlanFromFineTuningParams (BC67.cpp:3355-3483); returns True like the reference."""
    if load_library().cvttmi_bc7_plan_from_fine_tuning(ctypes.byref(encodingPlan), ctypes.byref(params)) != 0:
        raise CvttError("cvttmi_bc7_plan_from_fine_tuning failed")
    return True


def EncodeBC7(pBlocks, options=None, encodingPlan=None, device=0):
    """cvtt::Kernels::EncodeBC7 (reference ConvectionKernels_API.cpp:41-54): any multiple of
    NumParallelBlocks blocks; returns the packed 16-byte blocks."""
    return default_context(device).encode_bc7(pBlocks, options, encodingPlan)


def EncodeBC2(pBlocks, options=None, device=0):
    """cvtt::Kernels::EncodeBC2 (reference ConvectionKernels_API.cpp:101-115)."""
    return default_context(device).encode_bc2(pBlocks, options)


def EncodeBC3(pBlocks, options=None, device=0):
    return default_context(device).encode_bc3(pBlocks, options)


def EncodeBC4U(pBlocks, options=None, device=0):
    return default_context(device).encode_bc4(pBlocks, options, signed=False)


def EncodeBC4S(pBlocks, options=None, device=0):
    return default_context(device).encode_bc4(pBlocks, options, signed=True)


def EncodeBC5U(pBlocks, options=None, device=0):
    return default_context(device).encode_bc5(pBlocks, options, signed=False)


def EncodeBC5S(pBlocks, options=None, device=0):
    return default_context(device).encode_bc5(pBlocks, options, signed=True)


def EncodeETC2Alpha11(pBlocks, isSigned=False, options=None, device=0):
    """cvtt::Kernels::EncodeETC2Alpha11 (reference ConvectionKernels_API.cpp:258-268)."""
    return default_context(device).encode_etc2_alpha11(pBlocks, isSigned, options)


def DecodeBC7(pBC, device=0):
    """cvtt::Kernels::DecodeBC7 (reference ConvectionKernels_API.cpp:305-310), batched"""
    return default_context(device).decode_bc7(pBC)


def DecodeBC6HU(pBC, device=0):
    return default_context(device).decode_bc6h(pBC, signed=False)


def DecodeBC6HS(pBC, device=0):
    return default_context(device).decode_bc6h(pBC, signed=True)


def EncodeBC1(pBlocks, options=None, device=0):
    """cvtt::Kernels::EncodeBC1 (reference ConvectionKernels_API.cpp:86-99)."""
    return default_context(device).encode_bc1(pBlocks, options)


def EncodeBC6HU(pBlocks, options=None, device=0):
    """cvtt::Kernels::EncodeBC6HU (reference ConvectionKernels_API.cpp:56-69)."""
    return default_context(device).encode_bc6h(pBlocks, options, signed=False)


def EncodeBC6HS(pBlocks, options=None, device=0):
    """cvtt::Kernels::EncodeBC6HS (reference ConvectionKernels_API.cpp:71-84)."""
    return default_context(device).encode_bc6h(pBlocks, options, signed=True)


def EncodeETC1(pBlocks, options=None, compressionData=None, device=0):
    """cvtt::Kernels::EncodeETC1 (reference ConvectionKernels_API.cpp:201-214); the ETC1CompressionData scratch
    argument is accepted and ignored."""
    return default_context(device).encode_etc1(pBlocks, options)


class ETC2CompressionData:
    """What cvtt::Kernels::AllocETC2Data returns (reference ConvectionKernels_ETC.cpp:3100-3145).  The 136 KB of scratch have
    no counterpart (the kernels' scratch is LDS); what the reference fixes at allocation time and the encoder needs later are
    the two chroma axes, derived from the colour weights of the Options given here."""

    def __init__(self, options=None):
        self.alloc_options = Options.frombytes(bytes(options if options is not None else Options()))


def AllocETC2Data(options=None):
    """cvtt::Kernels::AllocETC2Data (no allocator callbacks: nothing is allocated in caller memory)."""
    return ETC2CompressionData(options)


def ReleaseETC2Data(compressionData):
    """cvtt::Kernels::ReleaseETC2Data: nothing to free."""
    return None


def EncodeETC2(pBlocks, options=None, compressionData=None, device=0):
    """cvtt::Kernels::EncodeETC2 (reference ConvectionKernels_API.cpp:216-229).  compressionData: from AllocETC2Data; its
    Options fix the chroma axes, as in the reference (scratch itself lives in LDS)."""
    return default_context(device).encode_etc2(pBlocks, options, compression_data=compressionData)


def EncodeETC2PunchthroughAlpha(pBlocks, options=None, compressionData=None, device=0):
    """cvtt::Kernels::EncodeETC2PunchthroughAlpha (reference ConvectionKernels_API.cpp:231-244)."""
    return default_context(device).encode_etc2_punchthrough_alpha(pBlocks, options, compression_data=compressionData)


def EncodeETC2RGBA(pBlocks, options=None, compressionData=None, device=0):
    """cvtt::Kernels::EncodeETC2RGBA (reference ConvectionKernels_API.cpp:270-286)."""
    return default_context(device).encode_etc2_rgba(pBlocks, options, compression_data=compressionData)


def EncodeETC2Alpha(pBlocks, options=None, device=0):
    """cvtt::Kernels::EncodeETC2Alpha (reference ConvectionKernels_API.cpp:246-256)."""
    return default_context(device).encode_etc2_alpha(pBlocks, options)
