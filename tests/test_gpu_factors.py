"""GPU parity: every factor kernel behind the C ABI against the CPU oracle on the same
seeded inputs.  Tolerance: FP64 results, relative 1e-10 (different summation
order / fused multiply-add only)."""
import ctypes as C

import numpy as np
import pytest

import oracle_api as o
import vplines_slam_amd as v
from test_oracle_factors import rand_pose, _line_setup

pytestmark = pytest.mark.gpu
RTOL = 1e-10


def close(a, b, rtol=RTOL):
    scale = max(1.0, np.abs(b).max())
    return np.abs(a - b).max() <= rtol * scale


def test_projection_factor_parity(gpu_ctx):
    rng = np.random.default_rng(101)
    n = 1000
    params = np.zeros((n, 22))
    pts = np.zeros((n, 6))
    for i in range(n):
        Pi, Pj = rand_pose(rng, 0.5), rand_pose(rng, 0.5)
        Pj[3:] = Pi[3:] + 0.1 * rng.normal(size=4)
        Pj[3:] /= np.linalg.norm(Pj[3:])
        params[i] = np.concatenate([Pi, Pj, rand_pose(rng, 0.05), [rng.uniform(0.1, 0.5)]])
        pts[i] = np.concatenate([rng.uniform(-0.5, 0.5, 2), [1.0], rng.uniform(-0.5, 0.5, 2), [1.0]])
    pts[0, 3:5] = 0.0  # pts_j on the optical axis: tangent-base special case (projection_factor.cpp:12-13)
    rg, jg = gpu_ctx.projection_factor(params, pts)
    rc, jc = o.projection_factor(params, pts)
    assert close(rg, rc) and close(jg, jc)
    assert np.all(jg[:, [6, 13, 20, 27, 34, 41]] == 0.0)
    rg2, _ = gpu_ctx.projection_factor(params, pts, want_jac=False)
    assert np.array_equal(rg, rg2)


def test_line_and_vp_factor_parity(gpu_ctx):
    rng = np.random.default_rng(102)
    n = 600
    params = np.zeros((n, 18))
    obs = rng.uniform(-0.5, 0.5, (n, 4))
    vp = rng.normal(size=(n, 3))
    vp[:, 2] = np.abs(vp[:, 2]) + 0.5
    for i in range(n):
        pose, ex, orth = _line_setup(rng)
        params[i] = np.concatenate([pose, ex, orth])
    rg, jg = gpu_ctx.line_factor(params, obs)
    rc, jc = o.line_factor(params, obs)
    assert close(rg, rc) and close(jg, jc)
    rg, jg = gpu_ctx.vp_factor(params, vp)
    rc, jc = o.vp_factor(params, vp)
    assert close(rg, rc) and close(jg, jc)


def test_parameterisations_parity(gpu_ctx):
    rng = np.random.default_rng(103)
    n = 500
    x = np.array([rand_pose(rng) for _ in range(n)])
    d = rng.normal(size=(n, 6)) * 0.1
    assert np.abs(gpu_ctx.pose_plus(x, d) - o.pose_plus(x, d)).max() < 1e-14
    assert np.abs(gpu_ctx.pose_plus(x, np.zeros((n, 6))) - x).max() < 1e-15
    orth = np.array([_line_setup(rng)[2] for _ in range(n)])
    d4 = rng.normal(size=(n, 4)) * 0.1
    assert np.abs(gpu_ctx.line_orth_plus(orth, d4) - o.line_orth_plus(orth, d4)).max() < 1e-13


def _windows(k, cfg_args=(8, 0, False)):
    opt = v.default_options()
    cfg = v.workload.config(*cfg_args)
    return [v.workload.generate(500 + i, cfg, 0.37 * i) for i in range(k)], opt


def test_preintegration_parity(gpu_ctx):
    ws, opt = _windows(6)
    pre = gpu_ctx.preintegrate(*v.workload.imu_batch_arrays(ws), opt)
    v.workload.set_preintegrations(ws, pre)
    ws2 = [w.copy() for w in ws]
    o.preintegrate_windows(ws2, opt)
    for a, b in zip(ws, ws2):
        for j in range(1, 11):
            for f in ("delta_p", "delta_q", "delta_v", "jacobian", "covariance"):
                x, y = np.array(getattr(a.preint[j], f)), np.array(getattr(b.preint[j], f))
                assert np.abs(x - y).max() <= 1e-12 * max(1e-30, np.abs(y).max()), f
            assert abs(a.preint[j].sum_dt - b.preint[j].sum_dt) < 1e-15


def test_imu_factor_parity(gpu_ctx):
    ws, opt = _windows(5)
    o.preintegrate_windows(ws, opt)
    rng = np.random.default_rng(104)
    n = 5 * 10
    pre = (v.Preintegration * n)()
    params = np.zeros((n, 32))
    k = 0
    for w in ws:
        for j in range(1, 11):
            C.memmove(C.byref(pre[k]), C.byref(w.preint[j]), C.sizeof(pre[k]))
            params[k] = np.concatenate([w.pose[j - 1], w.speed_bias[j - 1] + 0.01 * rng.normal(size=9), w.pose[j],
                                        w.speed_bias[j]])
            k += 1
    rg, jg = gpu_ctx.imu_factor(params, pre)
    rc, jc = o.imu_factor(params, pre)
    # the whitening matrix comes from inverting a covariance with condition ~1e10: compare relative to its scale
    assert np.abs(rg - rc).max() <= 1e-6 * max(1.0, np.abs(rc).max())
    assert np.abs(jg - jc).max() <= 1e-6 * max(1.0, np.abs(jc).max())


def test_prior_factor_parity(gpu_ctx):
    rng = np.random.default_rng(105)
    pr = v.Prior()
    kinds = [0, 0, 0, 1, 2]
    idx = 0
    pr.n_blocks = len(kinds)
    params = []
    for b, kd in enumerate(kinds):
        pr.block_kind[b] = kd
        pr.block_frame[b] = b if kd == 0 else 0
        pr.block_idx[b] = idx
        idx += 9 if kd == 1 else 6
        x0 = rng.normal(size=9) if kd == 1 else rand_pose(rng)
        for k in range(len(x0)):
            pr.x0[b][k] = x0[k]
        x = x0.copy()
        if kd == 1:
            x += 0.01 * rng.normal(size=9)
        else:
            x = o.pose_plus(x0[None], 0.05 * rng.normal(size=(1, 6)))[0]
            if b == 1:
                x[3:] *= -1.0   # q and -q are the same rotation: exercises the w < 0 sign flip (:516-520)
        params.append(x)
    n = idx
    pr.n = n
    J0 = rng.normal(size=(n, n))
    r0 = rng.normal(size=n)
    for i in range(n * n):
        pr.J0[i] = J0.flat[i]
    for i in range(n):
        pr.r0[i] = r0[i]
    params = np.concatenate(params)
    rg, jg = gpu_ctx.prior_factor(pr, params)
    rc, jc = o.prior_factor(pr, params)
    assert close(rg, rc) and close(jg, jc)
