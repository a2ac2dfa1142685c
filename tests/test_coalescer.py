"""csrc/coalescer.h, the queue behind the reference-style one-group calls of the C++ face, on the CPU with a stand-in
backend (tests/coalescer_harness.cpp), under AddressSanitizer and ThreadSanitizer:
* more caller threads of one kind than a launch takes (300 against 256, and 48 against 4): round 4's leader could launch
  without its own request and return with nothing written, leaving a dead stack object queued;
* a pool of mixed kinds: one slot per kind, launches of different kinds side by side, slots re-keyed when kinds change,
  more kinds than slots (the call then runs on the caller's own context)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = {
    "300_callers_one_kind": ["300", "12", "256", "1"],
    "48_callers_cap_4": ["48", "60", "4", "1"],
    "lone_caller": ["1", "200", "256", "1", "0"],
    "mixed_5_kinds": ["40", "40", "8", "5"],
    "more_kinds_than_slots": ["40", "40", "8", "13"],
}


@pytest.fixture(scope="module", params=["address,undefined", "thread"])
def harness(request, tmp_path_factory):
    exe = tmp_path_factory.mktemp("coalescer") / ("harness_" + request.param.split(",")[0])
    cmd = ["g++", "-std=c++11", "-O1", "-g", "-pthread", "-fsanitize=" + request.param, "-fno-omit-frame-pointer",
           os.path.join(ROOT, "tests", "coalescer_harness.cpp"), "-o", str(exe)]
    try:
        subprocess.check_call(cmd)
    except (OSError, subprocess.CalledProcessError):
        pytest.skip("g++ -fsanitize=%s not usable here" % request.param)
    return str(exe)


@pytest.mark.parametrize("case", sorted(CASES))
def test_coalescer_under_sanitizers(harness, case):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", TSAN_OPTIONS="halt_on_error=1")
    p = subprocess.run([harness] + CASES[case], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    out = p.stdout.decode() + p.stderr.decode()
    assert p.returncode == 0 and out.startswith("ok "), out[-3000:]
    if case == "lone_caller":
        assert "launches=200 largest=1 " in out, out  # a lone caller never waits and never shares a launch
    if case == "48_callers_cap_4":
        assert "largest=4 " in out, out
