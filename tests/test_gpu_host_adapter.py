"""The C++ host adapter (reference-shaped classes over the C ABI) against the oracle.  Compiles a
small C++ program against libvplines_hip.so and parses its output.  GPU only."""
import ctypes as C
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
    # EDline with the reference's default smoothed = false (Gaussian pre-blur) + LineFilter, as its demo programs call them;
    # 258 lines is what the reference's own edline_result.png shows for this frame (tests/test_edline_reference_picture.py)
    ld = o.edlines(imgs[0], min_len=25, smoothed=False)
    lf = o.line_filter(ld, 3.0)
    assert list(vals["demo"]) == [258, len(lf)] and len(ld) == 258
    assert np.abs(vals["demolines"].reshape(-1, 4) - lf[:, :4]).max() < 1e-3


def _quat_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def test_marginalization_info_mirror_matches_oracle(tmp_path, gpu_ctx):
    """vplhost::MarginalizationInfo / ResidualBlockInfo / IMUFactor / IntegrationBase / MarginalizationFactor driven as
    estimator.cpp:1229-1378 drives the reference's classes (tests/native/marg_info_check.cpp), against the oracle's
    marginalisation of the same window: m, n, kept blocks and their shifted addresses, x0, J0^T J0, J0^T r0; IMUFactor and
    IntegrationBase against the oracle's; the new prior evaluated at its linearisation point returns r0."""
    exe = str(tmp_path / "marg_info_check")
    libdir = os.path.join(ROOT, "vplines-slam_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "native", "marg_info_check.cpp"),
                           "-L", libdir, "-lvplines_hip", "-Wl,-rpath," + libdir, "-o", exe])
    opt = v.default_options()
    cfg = v.workload.config(40, 18, True)
    B, keep = v.workload.primed_batch(gpu_ctx, [7], cfg, opt, 3)
    w = B[0]
    assert w.prior is not None and w.prior.n > 0
    # para_LineFeature of every line: world orth from the start-camera-frame Pluecker (getLineOrthVector)
    Ric, tic = _quat_R(w.ex_pose[3:]), w.ex_pose[:3]
    orth = []
    for l in range(len(w.line_start)):
        s = w.line_start[l]
        Rwc = _quat_R(w.pose[s, 3:]) @ Ric
        twc = _quat_R(w.pose[s, 3:]) @ tic + w.pose[s, :3]
        vw = Rwc @ w.line_plk[l, 3:]
        nw = Rwc @ w.line_plk[l, :3] + np.cross(twc, vw)
        orth.append(o.plk_to_orth(np.concatenate([nw, vw])))
    fmt = lambda a: " ".join("%.17g" % x for x in np.asarray(a, np.float64).reshape(-1))
    lines = [fmt(w.pose), fmt(w.speed_bias), fmt(w.ex_pose)]
    imu = w.extra["imu_samples"][1]
    lines += [str(len(imu)), fmt(w.extra["imu_acc0"][1]), fmt(w.extra["imu_gyr0"][1]), fmt(w.speed_bias[1, 3:6]),
              fmt(w.speed_bias[1, 6:9]), fmt(imu)]
    poff = np.concatenate([[0], np.cumsum(w.point_nobs)])
    p0 = [i for i in range(len(w.point_start)) if w.point_start[i] == 0]
    lines.append(str(len(p0)))
    for i in p0:
        lines += [str(int(w.point_nobs[i])), fmt(w.point_obs[poff[i]:poff[i + 1]]), fmt(w.inv_depth[i])]
    loff = np.concatenate([[0], np.cumsum(w.line_nobs)])
    l0 = [i for i in range(len(w.line_start)) if w.line_start[i] == 0]
    lines.append(str(len(l0)))
    for i in l0:
        lines += [str(int(w.line_nobs[i])), fmt(w.line_obs[loff[i]:loff[i + 1]]), fmt(orth[i])]
    pr = w.prior
    lines += ["1", "%d %d" % (pr.n, pr.n_blocks)]
    lines.append(" ".join("%d %d %d" % (pr.block_kind[b], pr.block_frame[b], pr.block_idx[b]) for b in range(pr.n_blocks)))
    lines.append(fmt(np.array([list(pr.x0[b]) for b in range(pr.n_blocks)])))
    lines += [fmt(pr.J()), fmt(pr.r())]
    dump = str(tmp_path / "window.txt")
    open(dump, "w").write("\n".join(lines) + "\n")
    out = subprocess.check_output([exe, dump], text=True)
    vals = {ln.split()[0]: np.array([float(x) for x in ln.split()[1:]]) for ln in out.strip().splitlines()}

    # IntegrationBase + IMUFactor against the oracle
    wc = w.copy()
    o.preintegrate_windows([wc], opt)
    pc = wc.preint[1]
    ref = np.concatenate([[pc.sum_dt], list(pc.delta_p), list(pc.delta_q), list(pc.delta_v)])
    assert np.abs(vals["preint"] - ref).max() < 1e-12
    params = np.concatenate([w.pose[0], w.speed_bias[0], w.pose[1], w.speed_bias[1]])[None]
    pre1 = (v.capi.Preintegration * 1)()
    C.memmove(pre1, C.byref(wc.preint[1]), C.sizeof(v.capi.Preintegration))
    r, j = o.imu_factor(params, pre1)
    expect = np.concatenate([r[0], j[0, :105], j[0, 345:480]])
    assert np.abs(vals["imu"] - expect).max() < 1e-8 * max(1.0, np.abs(expect).max())

    # the marginalisation against the oracle's (optimizationwithLine body with zero iterations)
    o0 = v.default_options()
    o0.num_iterations = 0
    ref_w = w.copy()
    pc, rc = o.solve_window(ref_w, o0)
    m, n, nb = (int(x) for x in vals["mn"])
    assert (m, n, nb) == (rc.prior_m, pc.n, pc.n_blocks)
    blk = vals["blocks"].reshape(nb, 4).astype(int)
    for b in range(nb):
        # the returned address is the block of the NEXT window the prior will be attached to
        assert (blk[b, 0], blk[b, 1]) == (pc.block_kind[b], pc.block_frame[b])
        assert blk[b, 2] == (9 if pc.block_kind[b] == 1 else 7) and blk[b, 3] == pc.block_idx[b] + m
    J = vals["J0"].reshape(n, n)
    r0 = vals["r0"]
    Jc, rcv = pc.J(), pc.r()
    Ac = Jc.T @ Jc
    assert np.abs(J.T @ J - Ac).max() <= 1e-6 * np.abs(Ac).max()
    lam, V = np.linalg.eigh(0.5 * (Ac + Ac.T))
    sig = V[:, lam > 1e-6 * lam[-1]]
    bc = Jc.T @ rcv
    assert np.abs(sig.T @ (J.T @ r0 - bc)).max() <= 1e-5 * max(1.0, np.abs(bc).max())
    x0 = np.concatenate([np.array(pc.x0[b][:(9 if pc.block_kind[b] == 1 else 7)]) for b in range(nb)])
    assert np.abs(vals["x0"] - x0).max() < 1e-12
    assert np.abs(vals["prior_at_x0"] - r0).max() <= 1e-9 * max(1.0, np.abs(r0).max())
