"""Estimator::slideWindow + FeatureManager::removeBackShiftDepth / removeFront.
CPU: the oracle against an independent NumPy restatement and against geometric invariants (the re-anchored point / line is
the same world point / line).  GPU: vpl_ba_slide_window against the oracle."""
import numpy as np
import pytest

import oracle_api as o
import vplines_slam_amd as v
from test_line_map import make, quat_R

WS = 10


def cam(w, f):
    ric, tic = quat_R(w.ex_pose), w.ex_pose[:3]
    R = quat_R(w.pose[f])
    return R @ ric, w.pose[f, :3] + R @ tic


def plk_to_pose(plk, R, t):
    n, d = plk[:3], plk[3:]
    return np.concatenate([R @ n + np.cross(t, R @ d), R @ d])


def numpy_slide_old(w, init_depth):
    R0, P0 = cam(w, 0)
    R1, P1 = cam(w, 1)
    invd, plk = w.inv_depth.copy(), w.line_plk.copy()
    ps, pn, pdrop = w.point_start.copy(), w.point_nobs.copy(), np.full(len(w.point_start), -1)
    off = np.concatenate([[0], np.cumsum(w.point_nobs)])
    for i in range(len(ps)):
        if ps[i] != 0:
            ps[i] -= 1
            continue
        pdrop[i] = 0
        pn[i] = pn[i] - 1 if pn[i] - 1 >= 2 else 0
        if pn[i]:
            pj = R1.T @ (R0 @ (w.point_obs[off[i]] / invd[i]) + P0 - P1)
            invd[i] = 1.0 / (pj[2] if pj[2] > 0 else init_depth)
    ls, ln, ldrop = w.line_start.copy(), w.line_nobs.copy(), np.full(len(w.line_start), -1)
    for i in range(len(ls)):
        if ls[i] != 0:
            ls[i] -= 1
            continue
        ldrop[i] = 0
        ln[i] = ln[i] - 1 if ln[i] - 1 >= 2 else 0
        if ln[i]:
            plk[i] = plk_to_pose(plk[i], R1.T @ R0, R1.T @ (P0 - P1))
    return invd, plk, (ps, pn, pdrop, ls, ln, ldrop)


def numpy_slide_new(w):
    def front(start, nobs):
        s, n, d = start.copy(), nobs.copy(), np.full(len(start), -1)
        for i in range(len(s)):
            if start[i] == WS:
                s[i] -= 1
            elif start[i] + nobs[i] - 1 >= WS - 1:
                d[i] = WS - 1 - start[i]
                n[i] -= 1
        return s, n, d
    return front(w.point_start, w.point_nobs) + front(w.line_start, w.line_nobs)


def tracks_equal(st, ref):
    got = (st.point_start, st.point_nobs, st.point_drop, st.line_start, st.line_nobs, st.line_drop)
    return all(np.array_equal(a, b) for a, b in zip(got, ref))


def short_tracks(w, rng):
    """cut some tracks down to 1-2 observations and move a few to the last frames, as the FeatureManager holds them"""
    pn, ln = w.point_nobs.copy(), w.line_nobs.copy()
    keep_p, keep_l = [], []
    for i in range(len(pn)):
        n = int(rng.integers(1, 3)) if i % 4 == 0 else int(pn[i])
        keep_p.append(n)
    for i in range(len(ln)):
        n = int(rng.integers(1, 3)) if i % 5 == 0 else int(ln[i])
        keep_l.append(n)
    po = np.concatenate([[0], np.cumsum(pn)])
    lo = np.concatenate([[0], np.cumsum(ln)])
    w.point_obs = np.concatenate([w.point_obs[po[i]:po[i] + keep_p[i]] for i in range(len(pn))])
    w.line_obs = np.concatenate([w.line_obs[lo[i]:lo[i] + keep_l[i]] for i in range(len(ln))])
    w.point_nobs[:] = keep_p
    w.line_nobs[:] = keep_l
    for i in range(0, len(pn), 8):                      # a one-observation track born in the newest frame
        if w.point_nobs[i] == 1:
            w.point_start[i] = WS
    for i in range(0, len(ln), 10):
        if w.line_nobs[i] == 1:
            w.line_start[i] = WS
    return w


def test_oracle_slide_old_matches_numpy_and_geometry():
    w, opt = make(41, sigma_px=0.0, depth_sigma=0.0, t=0.3)
    before = w.copy()
    st = o.slide_window(w, opt, v.MARGIN_OLD, 5.0)
    invd, plk, ref = numpy_slide_old(before, 5.0)
    assert tracks_equal(st, ref)
    assert np.allclose(w.inv_depth, invd, rtol=1e-12, atol=0) and np.allclose(w.line_plk, plk, rtol=1e-11, atol=1e-13)
    assert np.array_equal(w.pose[:WS], before.pose[1:]) and np.array_equal(w.pose[WS], before.pose[WS])
    assert np.array_equal(w.speed_bias[:WS], before.speed_bias[1:]) and np.array_equal(w.speed_bias[WS], before.speed_bias[WS])
    # noise-free: the re-anchored depth reproduces the second observation, the re-anchored line is the same world line
    off = np.concatenate([[0], np.cumsum(before.point_nobs)])
    moved = np.flatnonzero((st.point_drop == 0) & (st.point_nobs > 0))
    assert len(moved) > 5
    R0, P0 = cam(before, 0)
    R1, P1 = cam(before, 1)
    for i in moved:
        pj = R1.T @ (R0 @ (before.point_obs[off[i]] / before.inv_depth[i]) + P0 - P1)
        assert np.allclose(pj / pj[2], before.point_obs[off[i] + 1], atol=1e-9)
        assert abs(1.0 / w.inv_depth[i] - pj[2]) < 1e-12
    lmoved = np.flatnonzero((st.line_drop == 0) & (st.line_nobs > 0))
    assert len(lmoved) > 3
    for i in lmoved:
        a, b = plk_to_pose(before.line_plk[i], R0, P0), plk_to_pose(w.line_plk[i], R1, P1)
        assert np.allclose(a, b, rtol=1e-9, atol=1e-11)
    assert np.array_equal(w.inv_depth[st.point_drop != 0], before.inv_depth[st.point_drop != 0])


def test_oracle_slide_with_short_tracks_and_second_new():
    rng = np.random.default_rng(5)
    w, opt = make(42, sigma_px=0.4, pose_noise=True, t=0.6)
    w = short_tracks(w, rng)
    w.inv_depth[6::12] = -1.0                             # not yet triangulated: depth -1 goes through the same formula
    before = w.copy()
    st = o.slide_window(w, opt, v.MARGIN_OLD, 5.0)
    invd, plk, ref = numpy_slide_old(before, 5.0)
    assert tracks_equal(st, ref)
    assert (st.point_nobs == 0).sum() > 0 and (st.line_nobs == 0).sum() > 0      # tracks were erased
    assert np.allclose(w.inv_depth, invd, rtol=1e-12, atol=0) and np.allclose(w.line_plk, plk, rtol=1e-11, atol=1e-13)
    assert np.any(w.inv_depth[(before.inv_depth < 0) & (st.point_drop == 0) & (st.point_nobs > 0)] == 1.0 / 5.0)
    w2 = before.copy()
    st2 = o.slide_window(w2, opt, v.MARGIN_SECOND_NEW, 5.0)
    assert tracks_equal(st2, numpy_slide_new(before))
    assert np.array_equal(w2.pose[WS - 1], before.pose[WS]) and np.array_equal(w2.pose[:WS - 1], before.pose[:WS - 1])
    assert np.array_equal(w2.speed_bias[WS - 1], before.speed_bias[WS])
    assert np.array_equal(w2.inv_depth, before.inv_depth) and np.array_equal(w2.line_plk, before.line_plk)
    assert (st2.point_drop >= 0).sum() > 0 and np.any(st2.point_start == WS - 1)


@pytest.mark.gpu
def test_gpu_slide_window_matches_oracle(gpu_ctx):
    rng = np.random.default_rng(9)
    ws = []
    for i in range(6):
        w, opt = make(400 + i, P=(4 if i == 3 else 60), L=(2 if i == 3 else 40), sigma_px=0.5, pose_noise=(i % 2 == 0), t=0.2 * i)
        if i >= 2:
            w = short_tracks(w, rng)
        if i == 4:
            w.inv_depth[::3] = -1.0
        ws.append(w)
    for flag in (v.MARGIN_OLD, v.MARGIN_SECOND_NEW):
        wg = [w.copy() for w in ws]
        wc = [w.copy() for w in ws]
        sts = gpu_ctx.slide_window(wg, flag, 5.0)
        for g, c, st in zip(wg, wc, sts):
            ref = o.slide_window(c, opt, flag, 5.0)
            assert tracks_equal(st, (ref.point_start, ref.point_nobs, ref.point_drop, ref.line_start, ref.line_nobs, ref.line_drop))
            assert np.array_equal(g.pose, c.pose) and np.array_equal(g.speed_bias, c.speed_bias)
            assert np.allclose(g.inv_depth, c.inv_depth, rtol=1e-12, atol=0)
            assert np.allclose(g.line_plk, c.line_plk, rtol=1e-11, atol=1e-13)
            same = st.point_drop != 0 if flag == v.MARGIN_OLD else np.ones(len(st.point_drop), bool)
            assert np.array_equal(g.inv_depth[same], c.inv_depth[same])
