"""Asynchronous variants of the five line-map entry points (include/vplines_ba.h: vpl_ba_*_async + vpl_ba_collect; VERDICT r2
item 6): the call returns after enqueueing, the results land in the caller's arrays when it is collected -- explicitly, or by
the next call that touches the context.  Every variant must give the bits of its synchronous twin (which the other test files
compare with the oracle)."""
import ctypes as C

import numpy as np
import pytest

import vplines_slam_amd as v
from test_line_map import make
from test_slide_window import short_tracks, tracks_equal

pytestmark = pytest.mark.gpu


def _tri_line_windows():
    ws = []
    for i in range(5):
        w, opt = make(900 + i, sigma_px=0.5, pose_noise=(i % 2 == 1), t=0.2 * i)
        w.line_triangulated[:] = 0
        w.line_triangulated[i::5] = 1
        w.line_plk[w.line_triangulated[:40] == 0] = 0
        ws.append(w)
    return ws, opt


def test_triangulate_async_equals_sync_and_writes_nothing_before_collect(gpu_ctx):
    ws, opt = _tri_line_windows()
    a = [w.copy() for w in ws]
    b = [w.copy() for w in ws]
    gpu_ctx.triangulate_lines(a)
    gpu_ctx.triangulate_lines(b, async_=True)
    gpu_ctx.collect()
    for x, y in zip(a, b):
        assert np.array_equal(x.line_triangulated, y.line_triangulated) and np.array_equal(x.line_plk, y.line_plk)
    assert any(x.line_triangulated.sum() > w.line_triangulated.sum() for x, w in zip(a, ws))
    # points
    for w in ws:
        w.inv_depth[1::3] = -1.0
    a = [w.copy() for w in ws]
    b = [w.copy() for w in ws]
    gpu_ctx.triangulate_points(a, 5.0)
    gpu_ctx.triangulate_points(b, 5.0, async_=True)
    gpu_ctx.collect()
    for x, y in zip(a, b):
        assert np.array_equal(x.inv_depth, y.inv_depth) and (x.inv_depth > 0).all()


def test_only_line_opt_and_slide_async_equal_sync(gpu_ctx):
    rng = np.random.default_rng(5)
    ws = []
    for i in range(6):
        w, opt = make(950 + i, sigma_px=0.5, pose_noise=(i % 3 == 2), t=0.15 * i)
        w.line_plk += rng.normal(0, 0.02, w.line_plk.shape) * np.abs(w.line_plk)
        ws.append(w)
    opt.num_iterations = 5
    a = [w.copy() for w in ws]
    b = [w.copy() for w in ws]
    ra = gpu_ctx.only_line_opt(a, opt)
    rb = gpu_ctx.only_line_opt(b, opt, async_=True)
    gpu_ctx.collect()
    for i in range(len(ws)):
        assert ra[i].iterations == rb[i].iterations and ra[i].final_cost == rb[i].final_cost
        assert ra[i].n_lines_removed == rb[i].n_lines_removed
        assert np.array_equal(a[i].line_plk, b[i].line_plk) and np.array_equal(a[i].line_removed, b[i].line_removed)
    # slide: the track bookkeeping is complete at return, the states after the collect
    ws = [short_tracks(w, rng) if i >= 2 else w for i, w in enumerate(ws)]
    for flag in (v.MARGIN_OLD, v.MARGIN_SECOND_NEW):
        a = [w.copy() for w in ws]
        b = [w.copy() for w in ws]
        sa = gpu_ctx.slide_window(a, flag, 5.0)
        sb = gpu_ctx.slide_window(b, flag, 5.0, async_=True)
        for x, y in zip(sa, sb):
            assert tracks_equal(x, (y.point_start, y.point_nobs, y.point_drop, y.line_start, y.line_nobs, y.line_drop))
        gpu_ctx.collect()
        for x, y in zip(a, b):
            assert np.array_equal(x.pose, y.pose) and np.array_equal(x.speed_bias, y.speed_bias)
            assert np.array_equal(x.inv_depth, y.inv_depth) and np.array_equal(x.line_plk, y.line_plk)


def test_marginalize_async_and_implicit_completion_by_the_next_call(gpu_ctx):
    from test_gpu_solve import make_windows
    ws, opt = make_windows(4, 60, 20, True, seed0=640)
    solved = [w.copy() for w in ws]
    gpu_ctx.solve_windows(solved, opt)
    pa, ma, na = gpu_ctx.marginalize([w.copy() for w in solved], opt, v.capi.MARGIN_OLD)
    pb, mb, nb = gpu_ctx.marginalize([w.copy() for w in solved], opt, v.capi.MARGIN_OLD, async_=True)
    # not collected: the next call on the context (a solve of other windows) completes it first
    other = [w.copy() for w in ws]
    gpu_ctx.solve_windows(other, opt)
    assert np.array_equal(ma, mb) and np.array_equal(na, nb) and (nb > 0).all()
    for i in range(4):
        assert pa[i].n == pb[i].n and np.array_equal(pa[i].J(), pb[i].J()) and np.array_equal(pa[i].r(), pb[i].r())
    # and the solve that followed is the one a fresh context gives
    again = [w.copy() for w in ws]
    gpu_ctx.solve_windows(again, opt)
    for x, y in zip(other, again):
        assert np.array_equal(x.pose, y.pose)


def _odometry_windows(n):
    """windows as solveOdometry() finds them: some depths unset, some lines not yet triangulated, noisy line map"""
    rng = np.random.default_rng(21)
    ws = []
    for i in range(n):
        w, opt = make(1200 + i, L=40, P=60, sigma_px=0.5, pose_noise=True, t=0.2 * i)
        w.inv_depth[i % 4::4] = -1.0
        w.line_plk += rng.normal(0, 0.01, w.line_plk.shape) * np.abs(w.line_plk)
        w.line_triangulated[:] = 1
        w.line_triangulated[i % 5::5] = 0
        w.line_plk[w.line_triangulated[:40] == 0] = 0
        w.line_removed[:] = 0
        ws.append(w)
    opt.num_iterations = 5
    return ws, opt


def test_solve_odometry_is_the_four_stages_back_to_back_and_matches_the_oracle(gpu_ctx):
    """vpl_ba_solve_odometry = Estimator::solveOdometry (estimator.cpp:624-648): triangulate || (triangulateLine -> onlyLineOpt)
    -> optimizationwithLine.  Bit-identical to the single entry points called in that order; against the oracle's stages
    the usual bars (same lines triangulated and erased, same iteration pattern, poses within 1e-4 m / 1e-6 rad)."""
    import oracle_api as o
    from test_gpu_solve import POS_TOL, ROT_TOL, pose_err
    ws, opt = _odometry_windows(5)
    a = [w.copy() for w in ws]
    pri_a, lrep_a, rep_a = gpu_ctx.solve_odometry(a, opt, 5.0)
    # the stages one by one
    b = [w.copy() for w in ws]
    gpu_ctx.triangulate_points(b, 5.0)
    gpu_ctx.triangulate_lines(b)
    lrep_b = gpu_ctx.only_line_opt(b, opt)
    for w in b:
        w.line_triangulated[w.line_removed[:len(w.line_triangulated)] != 0] = 0
    pri_b, rep_b = gpu_ctx.solve_windows(b, opt)
    for i, (x, y) in enumerate(zip(a, b)):
        assert np.array_equal(x.pose, y.pose) and np.array_equal(x.speed_bias, y.speed_bias)
        assert np.array_equal(x.inv_depth, y.inv_depth) and np.array_equal(x.line_plk, y.line_plk)
        assert np.array_equal(x.line_triangulated, y.line_triangulated) and np.array_equal(x.line_removed, y.line_removed)
        assert rep_a[i].iterations == rep_b[i].iterations and rep_a[i].final_cost == rep_b[i].final_cost
        assert lrep_a[i].iterations == lrep_b[i].iterations and lrep_a[i].n_lines_removed == lrep_b[i].n_lines_removed
        assert np.array_equal(pri_a[i].J(), pri_b[i].J())
    # a window in which onlyLineOpt erases nothing is solved where the line stage left the batch (no third upload, round 4);
    # one in which it does is uploaded again.  One window per call, both kinds, against the stages one by one.
    kinds = set()
    for i in range(4):
        wq, _ = make(1300 + i, L=40, P=60, sigma_px=0.1 if i < 3 else 0.5, pose_noise=(i == 3), t=0.2 * i)
        wq.inv_depth[i % 4::4] = -1.0
        wq.line_triangulated[:] = 1
        wq.line_removed[:] = 0
        if i == 3:
            wq.line_plk += np.random.default_rng(5).normal(0, 0.01, wq.line_plk.shape) * np.abs(wq.line_plk)
        x, y = wq.copy(), wq.copy()
        pri_x, lrep_x, rep_x = gpu_ctx.solve_odometry([x], opt, 5.0)
        gpu_ctx.triangulate_points([y], 5.0)
        gpu_ctx.triangulate_lines([y])
        lrep_y = gpu_ctx.only_line_opt([y], opt)
        y.line_triangulated[y.line_removed[:len(y.line_triangulated)] != 0] = 0
        pri_y, rep_y = gpu_ctx.solve_windows([y], opt)
        kinds.add(lrep_y[0].n_lines_removed == 0)
        assert np.array_equal(x.pose, y.pose) and np.array_equal(x.speed_bias, y.speed_bias), i
        assert np.array_equal(x.inv_depth, y.inv_depth) and np.array_equal(x.line_plk, y.line_plk), i
        assert np.array_equal(x.line_triangulated, y.line_triangulated) and np.array_equal(x.line_removed, y.line_removed)
        assert rep_x[0].iterations == rep_y[0].iterations and rep_x[0].final_cost == rep_y[0].final_cost
        assert lrep_x[0].n_lines_removed == lrep_y[0].n_lines_removed
        assert np.array_equal(pri_x[0].J(), pri_y[0].J()) and np.array_equal(pri_x[0].r(), pri_y[0].r())
    assert kinds == {True, False}, "both the resident and the re-upload path should have been taken"
    # the oracle's stages
    for i, w in enumerate(ws):
        c = w.copy()
        o.triangulate_points(c, opt, 5.0)
        o.triangulate_lines(c, opt)
        lrc = o.only_line_opt(c, opt)
        c.line_triangulated[c.line_removed[:len(c.line_triangulated)] != 0] = 0
        pc, rc = o.solve_window(c, opt)
        assert np.array_equal(a[i].line_triangulated, c.line_triangulated) and np.array_equal(a[i].line_removed, c.line_removed)
        assert lrep_a[i].iterations == lrc.iterations and lrep_a[i].n_lines_removed == lrc.n_lines_removed
        assert rep_a[i].iterations == rc.iterations and rep_a[i].num_successful_steps == rc.num_successful_steps, i
        dp, dr = pose_err(a[i], c)
        assert dp <= POS_TOL and dr <= ROT_TOL, (i, dp, dr)
        assert pri_a[i].n == pc.n
