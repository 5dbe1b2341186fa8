"""The C++ host adapter (reference-shaped classes over the C ABI) against the oracle.  Compiles a
small C++ program against libvplines_hip.so and parses its output.  GPU only."""
import os
import subprocess

import numpy as np
import pytest

import oracle_api as o
import vplines_slam_amd as v

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_adapter_classes_match_oracle(tmp_path):
    exe = str(tmp_path / "host_adapter_check")
    libdir = os.path.join(ROOT, "vplines-slam_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "native", "host_adapter_check.cpp"),
                           "-L", libdir, "-lvplines_hip", "-Wl,-rpath," + libdir, "-o", exe])
    out = subprocess.check_output([exe], text=True)
    vals = {ln.split()[0]: np.array([float(x) for x in ln.split()[1:]]) for ln in out.strip().splitlines()}
    Pi = np.array([0.1, -0.2, 0.3, 0.01, 0.02, -0.03, 0.9993]); Pj = np.array([0.3, -0.1, 0.35, 0.03, 0.01, -0.02, 0.9993])
    ex = np.array([-0.02, -0.06, 0.01, 0.0, 0.0, 0.7071067811865476, 0.7071067811865476])
    pts = np.array([0.12, -0.08, 1.0, 0.05, -0.11, 1.0])
    r, j = o.projection_factor(np.concatenate([Pi, Pj, ex, [0.25]])[None], pts[None])
    expect = np.concatenate([r[0], j[0, 0:14], j[0, 28:42], j[0, 42:44]])
    assert np.abs(vals["proj"] - expect).max() < 1e-10 * max(1, np.abs(expect).max())
    assert np.abs(vals["proj_nojac"] - r[0]).max() < 1e-10 * max(1, np.abs(r).max())
    orth = np.array([0.3, -0.2, 1.1, 0.2])
    lp = np.concatenate([Pi, ex, orth])[None]
    r, j = o.line_factor(lp, np.array([[0.1, 0.2, -0.15, 0.22]]))
    assert np.abs(vals["line"] - np.concatenate([r[0], j[0, 28:36]])).max() < 1e-9 * max(1, np.abs(j).max())
    r, j = o.vp_factor(lp, np.array([[0.3, -0.2, 0.9]]))
    assert np.abs(vals["vp"] - np.concatenate([r[0], j[0, 28:36]])).max() < 1e-9 * max(1, np.abs(j).max())
    assert np.abs(vals["pose_plus"] - o.pose_plus(Pi[None], np.array([[0.01, -0.02, 0.03, 0.004, -0.005, 0.006]]))[0]).max() < 1e-14
    assert np.abs(vals["orth_plus"] - o.line_orth_plus(orth[None], np.array([[0.01, -0.02, 0.03, 0.004]]))[0]).max() < 1e-13
    assert list(vals["sizes"]) == [7, 6, 4, 4]


def test_frontend_adapter_classes_match_oracle(tmp_path):
    """EDLineDetector::EDline + LineMatching::Matching through the reference-shaped C++ classes"""
    exe = str(tmp_path / "frontend_adapter_check")
    libdir = os.path.join(ROOT, "vplines-slam_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "native", "frontend_adapter_check.cpp"),
                           "-L", libdir, "-lvplines_hip", "-Wl,-rpath," + libdir, "-o", exe])
    imgs = [np.load(os.path.join(ROOT, "tests", "golden", "mh04_%d.npy" % i)) for i in (1, 2)]
    paths = []
    for i, im in enumerate(imgs):
        p = str(tmp_path / ("f%d.raw" % i))
        im.tofile(p)
        paths.append(p)
    out = subprocess.check_output([exe, paths[0], paths[1], "752", "480"], text=True)
    vals = {ln.split()[0]: np.array([float(x) for x in ln.split()[1:]]) for ln in out.strip().splitlines()}
    L = [o.edlines(im) for im in imgs]
    key = lambda a: a[np.lexsort((np.round(a[:, 3], 2), np.round(a[:, 2], 2), np.round(a[:, 1], 2), np.round(a[:, 0], 2)))]
    got = []
    for k in range(2):
        g = vals["lines%d" % k].reshape(-1, 10)
        got.append(g)
        assert len(g) == len(L[k])
        assert np.abs(key(g)[:, :4] - key(L[k])[:, :4]).max() < 1e-3
        assert np.abs(key(g)[:, 4:7] - key(L[k])[:, 4:7]).max() < 1e-9
    # the matcher is checked on the device's own line order (chain order == the oracle's order)
    ok, r2c, _ = o.line_match(imgs[0], imgs[1], got[0], got[1])
    assert vals["match"][0] == 1 and ok
    assert np.array_equal(vals["match"][1:].astype(int), r2c)
    assert list(vals["empty"]) == [0, 55, 55]          # Matching() == false leaves the output vector alone
