"""Line-map maintenance before the main solve: FeatureManager::triangulateLine and
Estimator::onlyLineOpt (line-only Levenberg-Marquardt, poses constant) + removeLineOutlier.
CPU: the oracle against an independent NumPy triangulation and against the ground truth of noise-free windows.
GPU: the HIP kernels against the oracle."""
import numpy as np
import pytest

import oracle_api as o
import vplines_slam_amd as v


def make(seed, L=40, P=60, sigma_px=0.3, pose_noise=False, t=0.3, orth_sigma=0.0):
    opt = v.default_options()
    cfg = v.workload.config(P, L, True)
    if not pose_noise:
        cfg.pose_sigma_p = cfg.pose_sigma_theta_deg = 0.0
    cfg.pix_sigma = sigma_px / 460.0
    cfg.orth_sigma = orth_sigma                  # 0: line_plk of the generated window is the true line
    w = v.workload.generate(seed, cfg, t)
    o.preintegrate_windows([w], opt)
    return w, opt


def quat_R(p):
    x, y, z, ww = p[3:7] / np.linalg.norm(p[3:7])
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * ww), 2 * (x * z + y * ww)],
                     [2 * (x * y + z * ww), 1 - 2 * (x * x + z * z), 2 * (y * z - x * ww)],
                     [2 * (x * z - y * ww), 2 * (y * z + x * ww), 1 - 2 * (x * x + y * y)]])


def numpy_triangulate(w, l):
    """independent restatement of feature_manager.cpp:413-563 for line l; returns plk or None"""
    ric, tic = quat_R(w.ex_pose), w.ex_pose[:3]
    off = int(np.sum(w.line_nobs[:l]))
    s, no = int(w.line_start[l]), int(w.line_nobs[l])
    cams = [(quat_R(w.pose[s + k]) @ ric, w.pose[s + k, :3] + quat_R(w.pose[s + k]) @ tic) for k in range(no)]
    R0, t0 = cams[0]
    ob = w.line_obs[off]
    p1, p2 = np.array([ob[0], ob[1], 1.0]), np.array([ob[2], ob[3], 1.0])
    pii = np.append(np.cross(p1, p2), 0.0)                      # plane through the origin and the two image points
    ni = pii[:3] / np.linalg.norm(pii[:3])
    best = None
    for k in range(1, no):
        R, t = R0.T @ cams[k][0], R0.T @ (cams[k][1] - t0)
        ob = w.line_obs[off + k]
        p3, p4 = R @ [ob[0], ob[1], 1.0] + t, R @ [ob[2], ob[3], 1.0] + t
        n = np.cross(p3 - t, p4 - t)
        pij = np.append(n, -t @ np.cross(p3, p4))
        c = ni @ (n / np.linalg.norm(n))
        if best is None or c < best[0]:
            best = (c, pij)
    if best is None or best[0] > 0.998:
        return None
    dp = np.outer(pii, best[1]) - np.outer(best[1], pii)
    return np.array([dp[0, 3], dp[1, 3], dp[2, 3], -dp[1, 2], dp[0, 2], -dp[0, 1]])


def line_dist_angle(a, b):
    """distance of the line from the camera origin (relative difference) and angle between the directions [deg]"""
    da, db = a[3:] / np.linalg.norm(a[3:]), b[3:] / np.linalg.norm(b[3:])
    ang = np.degrees(np.arccos(np.clip(abs(da @ db), 0, 1)))
    ra, rb = np.linalg.norm(a[:3]) / np.linalg.norm(a[3:]), np.linalg.norm(b[:3]) / np.linalg.norm(b[3:])
    return abs(ra - rb) / rb, ang


def test_oracle_triangulate_line_matches_numpy_and_truth():
    w, opt = make(77, sigma_px=0.0)
    truth = w.line_plk.copy()
    w.line_triangulated[:] = 0
    w.line_triangulated[3] = 1                                   # an already triangulated line is left alone
    w.line_plk[:] = 0
    w.line_plk[3] = truth[3] * 2.0
    n = o.triangulate_lines(w, opt)
    assert n == int(w.line_triangulated[:40].sum()) - 1 and 15 < n < 40
    assert np.array_equal(w.line_plk[3], truth[3] * 2.0)
    for l in range(40):
        ref = numpy_triangulate(w, l) if l != 3 else None
        if l == 3:
            continue
        assert (ref is not None) == bool(w.line_triangulated[l])
        if ref is not None:
            assert np.abs(w.line_plk[l] - ref).max() <= 1e-9 * np.abs(ref).max()
            rel, ang = line_dist_angle(w.line_plk[l], truth[l])   # noise-free observations: the true line
            assert rel < 1e-6 and ang < 1e-4


def test_oracle_only_line_opt_converges_and_flags_outliers():
    w, opt = make(78, sigma_px=0.0)               # noise-free observations: the optimum is the true map
    truth = w.line_plk.copy()
    rng = np.random.default_rng(4)
    w.line_plk += rng.normal(0, 0.02, w.line_plk.shape) * np.abs(w.line_plk)      # a perturbed line map
    off = np.concatenate([[0], np.cumsum(w.line_nobs)])
    w.line_obs[off[7] + 1, [1, 3]] += 0.05                        # one corrupted observation
    before = np.median([line_dist_angle(w.line_plk[l], truth[l])[1] for l in range(40)])
    opt.num_iterations = 8
    rep = o.only_line_opt(w, opt)
    assert rep.iterations >= 2 and rep.num_successful_steps >= 2 and rep.final_cost < rep.initial_cost
    keep = w.line_removed[:40] == 0
    after = np.median([line_dist_angle(w.line_plk[l], truth[l])[1] for l in range(40) if keep[l]])
    assert after < 0.05 * before
    assert w.line_removed[7] == 1 and rep.n_lines_removed == int(w.line_removed[:40].sum())
    # fewer than four lines: the function returns before touching anything (estimator.cpp:1019-1022)
    w3, opt3 = make(79, L=3)
    plk0 = w3.line_plk.copy()
    w3.line_obs[:, 0:4] += 0.2
    rep3 = o.only_line_opt(w3, opt3)
    assert rep3.iterations == 0 and rep3.n_lines_removed == 0 and np.array_equal(w3.line_plk, plk0)
    # untriangulated lines take no part
    w4, opt4 = make(80)
    w4.line_triangulated[::2] = 0
    plk0 = w4.line_plk.copy()
    o.only_line_opt(w4, opt4)
    assert np.array_equal(w4.line_plk[::2], plk0[::2]) and not np.array_equal(w4.line_plk[1::2], plk0[1::2])


@pytest.mark.gpu
def test_gpu_triangulate_lines_matches_oracle(gpu_ctx):
    ws = []
    for i in range(6):
        w, opt = make(100 + i, sigma_px=0.5, pose_noise=(i % 2 == 1), t=0.2 * i)
        w.line_triangulated[:] = 0
        w.line_triangulated[i::5] = 1
        w.line_plk[w.line_triangulated[:40] == 0] = 0
        ws.append(w)
    wg = [w.copy() for w in ws]
    gpu_ctx.triangulate_lines(wg)
    for g, c in zip(wg, ws):
        o.triangulate_lines(c, opt)
        assert np.array_equal(g.line_triangulated, c.line_triangulated)
        assert np.abs(g.line_plk - c.line_plk).max() <= 1e-9 * np.abs(c.line_plk).max()


@pytest.mark.gpu
def test_gpu_only_line_opt_matches_oracle(gpu_ctx):
    ws = []
    rng = np.random.default_rng(11)
    for i in range(8):
        w, opt = make(200 + i, L=(3 if i == 6 else 40), sigma_px=0.5, pose_noise=(i % 3 == 2), t=0.15 * i)
        w.line_plk += rng.normal(0, 0.02, w.line_plk.shape) * np.abs(w.line_plk)
        if i == 4:
            w.line_triangulated[1::3] = 0
        ws.append(w)
    for iters in (5, 1, 12):
        opt.num_iterations = iters
        wg = [w.copy() for w in ws]
        wc = [w.copy() for w in ws]
        reps = gpu_ctx.only_line_opt(wg, opt)
        for i in range(len(ws)):
            rc = o.only_line_opt(wc[i], opt)
            assert reps[i].iterations == rc.iterations and reps[i].num_successful_steps == rc.num_successful_steps
            assert reps[i].termination == rc.termination
            assert abs(reps[i].initial_cost - rc.initial_cost) <= 1e-9 * max(1.0, rc.initial_cost)
            assert abs(reps[i].final_cost - rc.final_cost) <= 1e-7 * max(1.0, rc.final_cost)
            assert np.array_equal(wg[i].line_removed, wc[i].line_removed)
            assert reps[i].n_lines_removed == rc.n_lines_removed
            scale = np.abs(wc[i].line_plk).max(axis=1, keepdims=True) + 1e-300
            assert (np.abs(wg[i].line_plk - wc[i].line_plk) / scale).max() < 1e-7
