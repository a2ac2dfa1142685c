import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    """A plain `pytest` (no -m) on a machine without a GPU skips the gpu tests; `-m gpu` there still fails loudly (the
    gpu_ctx fixture), so a GPU run can never pass on a silent skip."""
    if config.getoption("-m"):
        return
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device visible (select with -m gpu to make this an error)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_lib():
    """The plain-C restatement (test infrastructure); built on demand with gcc."""
    so = os.path.join(ROOT, "oracle", "libcvtt_oracle.so")
    src = os.path.join(ROOT, "oracle", "cvtt_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    from oracle import pyref
    return pyref.OracleLib()


@pytest.fixture(scope="session")
def ref_lib():
    """The real reference compiled into oracle/_ref (only where it was built)."""
    from oracle import pyref
    if not pyref.RefLib.available():
        pytest.skip("oracle/_ref/libcvtt_ref.so not present")
    return pyref.RefLib()


@pytest.fixture(scope="session")
def gpu_ctx():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    from convectionkernels_amd import api
    return api.Context(0)
