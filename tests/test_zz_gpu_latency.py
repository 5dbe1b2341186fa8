"""Latency of the drop-in call through the C ABI from C++ (tests/native/latency_check.cpp): upload / solve / download of 1, 8
and 64 windows of the benchmark shape.  The reference calls this path with ONE window at 10 Hz; the oracle (CPU restatement
of the reference path, one solve thread + 4 marginalisation threads) needs ~12 ms for the same window.

(The file sorts last on purpose: its bars are wall-clock bars on whatever host runs them -- the 64-window call took 2.9 ms on
the build's box and 29.9 ms on the round-3 driver box -- and a run with `-x` should report every parity test before it can
stop here.)"""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_and_run(tmpdir):
    exe = os.path.join(str(tmpdir), "latency_check")
    libdir = os.path.join(ROOT, "vplines-slam_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "native", "latency_check.cpp"),
                           "-L", libdir, "-lvplines_hip", "-Wl,-rpath," + libdir, "-o", exe])
    return json.loads(subprocess.check_output([exe], text=True, timeout=300))


def test_single_window_latency_through_the_c_abi(tmp_path):
    d = build_and_run(tmp_path)
    for k in ("nW1", "nW8", "nW64"):
        assert d[k]["iterations"] == 5 and d[k]["prior_n"] == 45
        assert abs(d[k]["upload_ms"] + d[k]["solve_ms"] + d[k]["download_ms"] - d[k]["total_ms"]) < 0.5
    print("C-ABI latency [ms]:", d)
    assert d["nW1"]["total_ms"] <= 3.0
    # the 64-window call (measured 2.9 ms on the build's box, 29.9 ms on the round-3 driver box with
    # dozens of pageable copies per call; now one pinned H2D + one pinned D2H)
    assert d["nW64"]["total_ms"] <= 6.0
    assert d["nW64"]["upload_ms"] + d["nW64"]["download_ms"] <= 2.0
    # the wall clock of a leg stays near its device time: nothing waits on the host side
    for k in ("nW1", "nW8", "nW64"):
        assert d[k]["device_solve_ms"] > 0 and d[k]["solve_ms"] <= d[k]["device_solve_ms"] + 1.0, d[k]
