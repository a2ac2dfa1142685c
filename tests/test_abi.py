"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/cvtt_mi355x.h declares, and its PODs have the reference's layout.  No compute calls."""
import ctypes
import os
import re

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
    for hdr in os.listdir(os.path.join(ROOT, "include")):
        assert not bad.search(open(os.path.join(ROOT, "include", hdr)).read()), hdr
