"""Vanishing-point stage of the line tracker (vanishing_point_detection.cpp).
CPU: glibc rand() known answers, the fixed elementary functions against NumPy, the oracle on a synthetic Manhattan scene
(finds the three directions, classifies the lines) and against a NumPy restatement of the sphere grid.
GPU: vpl_vp_detect_batch against the oracle, bit for bit (pairs drawn, grid, winning hypothesis, VPs, line classes)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_api as o

HERE = os.path.dirname(os.path.abspath(__file__))
F, CX, CY = 458.654, 376.0, 240.0


def manhattan_lines(seed, n_per_dir=(30, 25, 20), noise=0.3, rot=None):
    """image segments of 3-D lines along three orthogonal directions R[:, k] seen by a pinhole camera"""
    rng = np.random.default_rng(seed)
    if rot is None:
        a, b, c = 0.2, -0.15, 0.1
        Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
        Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
        Rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
        rot = Rz @ Ry @ Rx
    lines, label = [], []
    for k in range(3):
        while sum(1 for l in label if l == k) < n_per_dir[k]:
            p = np.array([rng.uniform(-3, 3), rng.uniform(-2, 2), rng.uniform(4, 9)])
            q = p + rot[:, k] * rng.uniform(0.8, 2.0)
            if q[2] < 1:
                continue
            uv = [(F * x[0] / x[2] + CX, F * x[1] / x[2] + CY) for x in (p, q)]
            e = np.array([uv[0][0], uv[0][1], uv[1][0], uv[1][1]]) + rng.normal(0, noise, 4)
            if (e[[0, 2]] < 0).any() or (e[[0, 2]] > 751).any() or (e[[1, 3]] < 0).any() or (e[[1, 3]] > 479).any():
                continue
            if np.hypot(e[0] - e[2], e[1] - e[3]) < 35:
                continue
            lines.append(e)
            label.append(k)
    order = rng.permutation(len(lines))
    return np.array(lines, np.float32)[order], np.array(label)[order], rot


def test_glibc_rand_known_answers():
    lib = o.load()
    out = (C.c_int * 6)()
    lib.orc_glibc_rand(1, 6, out)
    assert list(out) == [1804289383, 846930886, 1681692777, 1714636915, 1957747793, 424238335]
    lib.orc_glibc_rand(0, 2, out)                                        # srand(0) is srand(1)
    assert list(out)[:2] == [1804289383, 846930886]


def test_detmath_against_numpy():
    lib = o.load()
    rng = np.random.default_rng(0)
    dp = C.POINTER(C.c_double)

    def run(fn, x, y=None):
        x = np.ascontiguousarray(x, np.float64)
        y = np.ascontiguousarray(y if y is not None else x, np.float64)
        out = np.empty_like(x)
        lib.orc_detmath(fn, len(x), x.ctypes.data_as(dp), y.ctypes.data_as(dp), out.ctypes.data_as(dp))
        return out
    x = np.concatenate([rng.uniform(-7, 7, 200000), np.arange(361) * (2 * np.pi / 360), [0.0, -0.0]])
    assert np.abs(run(0, x) - np.sin(x)).max() < 3e-16 and np.abs(run(1, x) - np.cos(x)).max() < 3e-16
    t = np.concatenate([rng.uniform(-60, 60, 200000), [0, 0.4375, 0.6875, 1.1875, 2.4375, 1e300, -1e300, np.inf, -np.inf]])
    assert np.abs(run(2, t) - np.arctan(t)).max() < 5e-16
    c = np.concatenate([rng.uniform(-1, 1, 200000), [1.0, -1.0, 0.0, 1 - 1e-16, 0.999999999]])
    assert np.abs(run(3, c) - np.arccos(c)).max() < 1e-15
    a, b = rng.uniform(-50, 50, 200000), rng.uniform(-50, 50, 200000)
    a[:4], b[:4] = [0.0, 0.0, 1.0, -1.0], [1.0, -1.0, 0.0, 0.0]
    assert np.abs(run(4, a, b) - np.arctan2(a, b)).max() < 1e-15


def numpy_grid(ends):
    e = ends.astype(np.float32)
    n = len(e)
    p1 = np.stack([e[:, 0], e[:, 1], np.ones(n)], 1).astype(np.float64)
    p2 = np.stack([e[:, 2], e[:, 3], np.ones(n)], 1).astype(np.float64)
    para = np.cross(p1, p2)
    dx, dy = (e[:, 0] - e[:, 1]).astype(np.float64), (e[:, 2] - e[:, 3]).astype(np.float64)     # the reference's lineinfo
    length = np.hypot(dx, dy)
    ori = np.arctan2(dy, dx)
    ori[ori < 0] += np.pi
    g = np.zeros((90, 360))
    deg = np.pi / 180
    for i in range(n - 1):
        for j in range(i + 1, n):
            pt = np.cross(para[i], para[j])
            if pt[2] == 0:
                continue
            X, Y, Z = pt[0] / pt[2] - CX, pt[1] / pt[2] - CY, F
            la = min(int(np.arccos(Z / np.sqrt(X * X + Y * Y + Z * Z)) / deg), 89)
            lo = min(int((np.arctan2(X, Y) + np.pi) / deg), 359)
            dev = abs(ori[i] - ori[j])
            dev = min(np.pi - dev, dev)
            if dev > 60 * deg:
                continue
            g[la, lo] += np.sqrt(length[i] * length[j]) * (np.sin(2 * dev) + 0.2)
    out = np.zeros_like(g)
    for i in range(1, 89):
        for j in range(1, 359):
            out[i, j] = g[i, j] + g[i - 1:i + 2, j - 1:j + 2].sum() / 9
    return out


def test_oracle_vp_detect_manhattan_scene():
    ends, label, rot = manhattan_lines(1)
    vps, ids, it, dbg = o.vp_detect(ends, ends, F, CX, CY, 12345, True, full=True)
    assert it == 105 and dbg["drawn"][0] >= 210 and dbg["drawn"][1] >= dbg["drawn"][0]
    assert np.allclose(np.linalg.norm(vps, axis=1), 1, atol=1e-12)
    assert abs(vps[0] @ vps[1]) < 1e-6 and abs(vps[0] @ vps[2]) < 1e-6 and abs(vps[1] @ vps[2]) < 1e-6
    # every detected VP is one of the scene directions (up to sign) within 2 degrees, all three are found
    cosang = np.abs(vps @ rot)
    assert (cosang.max(axis=1) > np.cos(np.deg2rad(2.0))).all() and sorted(cosang.argmax(axis=1)) == [0, 1, 2]
    # classified lines agree with the direction they were generated from
    perm = cosang.argmax(axis=1)                       # vp k <-> scene direction perm[k]
    cls = ids < 3
    assert cls.mean() > 0.5 and (perm[ids[cls]] == label[cls]).mean() > 0.95
    assert np.array_equal(dbg["hyp"][dbg["best_idx"]], vps)            # first frame: no swap
    assert dbg["scores"][dbg["best_idx"]] == dbg["scores"].max() and dbg["best_idx"] == int(dbg["scores"].argmax())
    assert np.allclose(dbg["grid"], numpy_grid(ends), rtol=1e-12, atol=1e-9)
    # later frames: vps[1] / vps[2] swap when |vps[1].y| <= 0.8; same seed, same draws
    vps2, _, _, dbg2 = o.vp_detect(ends, ends, F, CX, CY, 12345, False, full=True)
    assert dbg2["best_idx"] == dbg["best_idx"]
    want = vps if abs(vps[1, 1]) > 0.8 else vps[[0, 2, 1]]
    assert np.array_equal(vps2, want)
    # another seed: other pairs, (almost always) the same directions
    vps3, _, _, dbg3 = o.vp_detect(ends, ends, F, CX, CY, 777, True, full=True)
    assert not np.array_equal(dbg3["pairs"], dbg["pairs"])
    assert (np.abs(vps3 @ rot).max(axis=1) > np.cos(np.deg2rad(3.0))).all()
    # degenerate input: parallel lines only -> -1 instead of the reference's endless loop
    par = np.array([[10, 10 + 5 * k, 300, 10 + 5 * k] for k in range(6)], np.float32)
    assert o.vp_detect(par, par, F, CX, CY, 1, True)[2] == -1
    assert o.vp_detect(ends[:1], ends, F, CX, CY, 1, True)[2] == -1


@pytest.mark.gpu
def test_gpu_vp_detect_matches_oracle_bit_for_bit():
    import vplines_slam_amd as v
    imgs = [np.load(os.path.join(HERE, "golden", "mh04_%d.npy" % i)) for i in (1, 2)]
    real = [o.edlines(im)[:, :4] for im in imgs]
    cases = []
    for k in range(4):
        ends, _, _ = manhattan_lines(10 + k, noise=0.2 + 0.2 * k)
        cases.append((ends, ends, 1000 + k, k == 0))
    cases.append((real[0], real[0], 42, False))
    cases.append((real[1][::3], real[1], 43, False))                 # hypotheses from a subset, all lines classified
    e3, _, _ = manhattan_lines(20, n_per_dir=(1, 1, 1))
    cases.append((e3, e3, 5, True))                                   # three lines
    par = np.array([[10, 10 + 5 * k, 300, 10 + 5 * k] for k in range(6)], np.float32)
    cases.append((par, par, 6, True))                                 # no hypothesis possible
    cases.append((real[0][:1], real[0], 7, False))                    # one line
    fe = v.frontend.FrontendContext(device=0, max_images=len(cases), width=752, height=480, max_lines=512)
    vps, ids, status = fe.vp_detect([c[0] for c in cases], [c[1] for c in cases], F, CX, CY, [c[2] for c in cases],
                                    [int(c[3]) for c in cases])
    for k, (he, ae, seed, first) in enumerate(cases):
        wv, wi, it, dbg = o.vp_detect(he, ae, np.float32(F), np.float32(CX), np.float32(CY), seed, first, full=True)
        if it < 0:
            assert status[k] == -1 and not vps[k].any() and (ids[k] == 3).all()
            continue
        assert status[k] == 0
        grid, pairs, best, drawn = fe.vp_debug(k)
        assert np.array_equal(pairs, dbg["pairs"]) and drawn == dbg["drawn"][0]
        assert np.array_equal(grid, dbg["grid"])                      # ordered per-cell sums: the same doubles
        assert best == dbg["best_idx"]
        assert np.array_equal(vps[k], wv)
        assert np.array_equal(ids[k], wi)
    assert (ids[4] < 3).sum() > 10                                    # the real frame has classified lines
    fe.close()
