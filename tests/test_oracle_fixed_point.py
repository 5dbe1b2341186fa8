"""A solver-independent check of the oracle's restatement of ceres' trust-region solve (CPU only).

The reference's arithmetic for the solve lives in ceres-solver 1.12, which is absent; the oracle restates it.  What can be
checked without ceres is the solve's FIXED POINT: run to convergence, the oracle must end where any least-squares solver
ends.  scipy.optimize.least_squares (trust-region reflective, finite-difference Jacobian -- it never sees the oracle's
Jacobians or its accept / reject logic) minimises the same sum of squares, built from the oracle's own single-factor
residual evaluators through the reference's local parameterisations.  Compared: final cost, and the gauge-invariant state
(poses relative to frame 0, body-frame velocities, biases, inverse depths, lines in their start camera frame).
Windows: points + IMU + prior, and points + lines + IMU + prior (no VP factors: their Jacobian in the reference is a literal matrix, not
the derivative, so their fixed point is not a least-squares minimum -- DESIGN.md section 2b).  Losses off (huber delta
1e9): scipy's robust losses act per scalar residual, ceres' per residual block."""
import os
import sys

import numpy as np
import pytest
from scipy.optimize import least_squares

import oracle_api as o
import vplines_slam_amd as v

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden as mg   # independent NumPy statement of the parameterisations / line maps   # noqa: E402

NF = 11


def _Twc(pose, ex):
    R, ric = mg.quat_R(pose[3:]), mg.quat_R(ex[3:])
    return R @ ric, pose[:3] + R @ ex[:3]


def _plk_to_world(Lc, R, t):          # L_w = [R n + t x (R v); R v]
    n, d = Lc[:3], Lc[3:]
    return np.concatenate([R @ n + np.cross(t, R @ d), R @ d])


class _Problem:
    def __init__(self, w, opt):
        self.w, self.opt = w, opt
        self.P, self.L = len(w.point_start), len(w.line_start)
        self.pose0, self.sb0, self.ex0 = w.pose.copy(), w.speed_bias.copy(), w.ex_pose.copy()
        self.invd0 = w.inv_depth.copy()
        self.orth0 = np.zeros((self.L, 4))
        for l in range(self.L):           # vector2double: start-camera-frame Pluecker -> world orthonormal at the initial poses
            R, t = _Twc(w.pose[w.line_start[l]], w.ex_pose)
            self.orth0[l] = mg.plk_to_orth(_plk_to_world(w.line_plk[l], R, t))
        self.n = 66 + 99 + 6 + self.P + 4 * self.L

    def state(self, z):
        pose = np.array([mg.pose_plus(self.pose0[f], z[6 * f:6 * f + 6]) for f in range(NF)])
        sb = self.sb0 + z[66:165].reshape(NF, 9)
        ex = mg.pose_plus(self.ex0, z[165:171])
        invd = self.invd0 + z[171:171 + self.P]
        orth = np.array([mg.orth_plus(self.orth0[l], z[171 + self.P + 4 * l:175 + self.P + 4 * l]) for l in range(self.L)])
        return pose, sb, ex, invd, orth.reshape(self.L, 4)

    def residuals(self, z):
        w, opt = self.w, self.opt
        pose, sb, ex, invd, orth = self.state(z)
        out = []
        prm = np.array([np.concatenate([pose[j - 1], sb[j - 1], pose[j], sb[j]]) for j in range(1, NF)])
        pre = (v.capi.Preintegration * 10)(*[w.preint[j] for j in range(1, NF)])
        out.append(o.imu_factor(prm, pre, opt.g_norm, want_jac=False)[0].ravel())
        poff = np.concatenate([[0], np.cumsum(w.point_nobs)])
        prm, pts = [], []
        for p in range(self.P):
            s = w.point_start[p]
            for k in range(1, w.point_nobs[p]):
                prm.append(np.concatenate([pose[s], pose[s + k], ex, [invd[p]]]))
                pts.append(np.concatenate([w.point_obs[poff[p]], w.point_obs[poff[p] + k]]))
        if prm:
            out.append(o.projection_factor(np.array(prm), np.array(pts), opt.focal_length / 1.5, want_jac=False)[0].ravel())
        loff = np.concatenate([[0], np.cumsum(w.line_nobs)])
        prm, obs = [], []
        for l in range(self.L):
            s = w.line_start[l]
            for k in range(w.line_nobs[l]):
                prm.append(np.concatenate([pose[s + k], ex, orth[l]]))
                obs.append(w.line_obs[loff[l] + k][:4])
        if prm:
            out.append(o.line_factor(np.array(prm), np.array(obs), opt.line_factor, want_jac=False)[0].ravel())
        if w.prior is not None:
            out.append(o.prior_factor(w.prior, np.concatenate([pose[0], sb[0], ex]), want_jac=False)[0])
        return np.concatenate(out)

    def invariants(self, pose, sb, invd, line_plk):
        R0, p0 = mg.quat_R(pose[0, 3:]), pose[0, :3]
        rel_p = np.array([R0.T @ (pose[f, :3] - p0) for f in range(NF)])
        rel_R = np.array([R0.T @ mg.quat_R(pose[f, 3:]) for f in range(NF)])
        v_body = np.array([mg.quat_R(pose[f, 3:]).T @ sb[f, :3] for f in range(NF)])
        return rel_p, rel_R, v_body, sb[:, 3:], invd, line_plk

    def line_plk_cam(self, pose, ex, orth):
        out = np.zeros((self.L, 6))
        for l in range(self.L):
            R, t = _Twc(pose[self.w.line_start[l]], ex)
            out[l] = mg.plk_from_pose(mg.orth_to_plk(orth[l]), R, t)
            out[l] /= np.linalg.norm(out[l][3:])
        return out


@pytest.mark.parametrize("P,L", [(14, 0), (10, 6)])
def test_converged_oracle_solve_is_the_least_squares_minimum(P, L):
    opt = v.default_options()
    opt.num_iterations = 200
    opt.huber_delta = 1e9
    opt.marginalization_flag = v.capi.MARGIN_NONE
    cfg = v.workload.config(P, L, False)
    w = v.workload.generate(v.workload.seed_for(2 if L == 0 else 3, 8100 + P), cfg, 0.7)
    o.preintegrate_windows([w], opt)
    # A prior on pose 0, speed / bias 0 and the extrinsic (diagonal J0, r0 = 0 at the initial values).  Without it the window
    # has a free gauge and, worse for this test, directions (accelerometer bias against gravity and scale) along which ceres'
    # dogleg with its mu >= 1e-8 regularisation crawls: 200 accepted steps of 1e-5 cost each, 10 % above the minimum.
    pr = v.Prior()
    pr.n, pr.n_blocks = 21, 3
    for b, (kind, idx, x0) in enumerate([(0, 0, w.pose[0]), (1, 6, w.speed_bias[0]), (2, 15, w.ex_pose)]):
        pr.block_kind[b], pr.block_frame[b], pr.block_idx[b] = kind, 0, idx
        for j in range(len(x0)):
            pr.x0[b][j] = float(x0[j])
    wts = [300.0] * 3 + [1000.0] * 3 + [30.0] * 3 + [300.0] * 3 + [3000.0] * 3 + [1000.0] * 6
    for i, wt in enumerate(wts):
        pr.J0[i * 21 + i] = wt
    w.prior = pr
    prob = _Problem(w, opt)
    r0 = prob.residuals(np.zeros(prob.n))
    wo = w.copy()
    lib = o.load()
    lib.orc_set_tolerance_scale.argtypes = [__import__("ctypes").c_double]
    lib.orc_set_tolerance_scale(1e-9)          # test hook: ceres' three tolerances x 1e-9, so that the solve runs to its fixed point
    try:
        _, rep = o.solve_window(wo, opt)
    finally:
        lib.orc_set_tolerance_scale(0.0)
    assert rep.iterations >= 5
    assert abs(0.5 * r0 @ r0 - rep.initial_cost) <= 1e-9 * rep.initial_cost     # the two sides minimise the same function
    sol = least_squares(prob.residuals, np.zeros(prob.n), jac="2-point", method="trf", x_scale="jac", xtol=1e-15, ftol=1e-15,
                        gtol=1e-12, max_nfev=400)
    cost_s = sol.cost
    pose, sb, ex, invd, orth = prob.state(sol.x)
    assert rep.final_cost >= cost_s * (1 - 1e-9), (rep.final_cost, cost_s)
    assert rep.final_cost - cost_s <= 1e-8 * max(cost_s, 1e-12) + 1e-9, (rep.final_cost, cost_s, rep.iterations)
    a = prob.invariants(pose, sb, invd, prob.line_plk_cam(pose, ex, orth))
    lo = wo.line_plk / np.linalg.norm(wo.line_plk[:, 3:], axis=1, keepdims=True) if L else wo.line_plk
    b = prob.invariants(wo.pose, wo.speed_bias, wo.inv_depth, lo)
    names = ["relative position", "relative rotation", "body velocity", "biases", "inverse depth", "line (start camera frame)"]
    tols = [1e-6, 1e-7, 1e-6, 1e-7, 1e-6, 1e-4]      # measured: 4e-8 m, 1e-9 rad, 5e-8 m/s, 1e-9, 2e-8, 2e-6
    worst = {}
    for nm, x, y, tol in zip(names, a, b, tols):
        if x.size == 0:
            continue
        worst[nm] = float(np.abs(x - y).max())
        assert worst[nm] <= tol, (nm, worst[nm])
    print("P=%d L=%d: oracle %d iterations, cost %.9g vs scipy %.9g; worst gauge-invariant differences %s"
          % (P, L, rep.iterations, rep.final_cost, cost_s, {k: "%.2e" % x for k, x in worst.items()}))
