"""GPU parity of the whole window solve (the optimizationwithLine body) against the CPU
oracle on identical synthetic windows.  Bar (BASELINE.json north_star): pose translation
<= 1e-4 m and rotation <= 1e-6 rad on every frame of the window."""
import ctypes as C

import numpy as np
import pytest

import oracle_api as o
import vplines_slam_amd as v

pytestmark = pytest.mark.gpu

POS_TOL = 1e-4
ROT_TOL = 1e-6


def rot_angle(qa, qb):
    """angle of qa^-1 * qb, quaternions stored x,y,z,w"""
    xa, ya, za, wa = qa
    xb, yb, zb, wb = qb
    w = wa * wb + xa * xb + ya * yb + za * zb
    vx = wa * xb - xa * wb - (ya * zb - za * yb)
    vy = wa * yb - ya * wb - (za * xb - xa * zb)
    vz = wa * zb - za * wb - (xa * yb - ya * xb)
    return 2.0 * np.arctan2(np.sqrt(vx * vx + vy * vy + vz * vz), abs(w))


def pose_err(a, b):
    dp = np.linalg.norm(a.pose[:, :3] - b.pose[:, :3], axis=1).max()
    dr = max(rot_angle(a.pose[i, 3:], b.pose[i, 3:]) for i in range(11))
    return dp, dr


def make_windows(n, P, L, vp, seed0=0, config_id=3):
    opt = v.default_options()
    cfg = v.workload.config(P, L, vp)
    ws = [v.workload.generate(v.workload.seed_for(config_id, seed0 + i), cfg, 0.61 * (seed0 + i)) for i in range(n)]
    o.preintegrate_windows(ws, opt)
    return ws, opt


@pytest.mark.parametrize("P,L,vp", [(200, 0, False), (200, 80, False), (200, 80, True), (30, 40, True), (37, 11, True)])
def test_window_solve_parity_no_prior(gpu_ctx, P, L, vp):
    ws, opt = make_windows(4, P, L, vp)
    wg = [w.copy() for w in ws]
    wc = [w.copy() for w in ws]
    pri_g, rep_g = gpu_ctx.solve_windows(wg, opt)
    for i in range(len(ws)):
        pri_c, rep_c = o.solve_window(wc[i], opt)
        assert rep_g[i].iterations == rep_c.iterations
        assert rep_g[i].num_successful_steps == rep_c.num_successful_steps
        assert abs(rep_g[i].initial_cost - rep_c.initial_cost) <= 1e-9 * rep_c.initial_cost
        assert abs(rep_g[i].final_cost - rep_c.final_cost) <= 1e-4 * max(1.0, rep_c.final_cost)
        dp, dr = pose_err(wg[i], wc[i])
        assert dp <= POS_TOL and dr <= ROT_TOL, (dp, dr)
        assert np.abs(wg[i].speed_bias - wc[i].speed_bias).max() < 1e-4
        if P:
            assert np.abs(wg[i].inv_depth - wc[i].inv_depth).max() < 1e-5
        assert rep_g[i].prior_n == rep_c.prior_n and rep_g[i].prior_m == rep_c.prior_m
        assert pri_g[i].n == pri_c.n and pri_g[i].n_blocks == pri_c.n_blocks
        nb = pri_c.n_blocks
        assert list(pri_g[i].block_kind[:nb]) == list(pri_c.block_kind[:nb])
        assert list(pri_g[i].block_frame[:nb]) == list(pri_c.block_frame[:nb])
        assert list(pri_g[i].block_idx[:nb]) == list(pri_c.block_idx[:nb])
        # the prior is defined up to an orthogonal transform of its rows: compare J0^T J0 and J0^T r0
        Jg, Jc = pri_g[i].J(), pri_c.J()
        Ag, Ac = Jg.T @ Jg, Jc.T @ Jc
        bg, bc = Jg.T @ pri_g[i].r(), Jc.T @ pri_c.r()
        assert np.abs(Ag - Ac).max() <= 1e-5 * np.abs(Ac).max()
        # J0^T r0 is the projection of b on the kept eigenvectors.  Which of the noise-level eigenvalues
        # pass the reference's `> 1e-8` test is decided by rounding in ANY eigen-solver, and the two A's
        # themselves agree only to ~1e-8 |A| (different summation orders), so directions with
        # lambda < 1e-6 lambda_max are below the noise floor of the comparison: compare on the rest
        lam, V = np.linalg.eigh(0.5 * (Ac + Ac.T))
        sig = V[:, lam > 1e-6 * lam[-1]]
        assert np.abs(sig.T @ (bg - bc)).max() <= 1e-5 * max(1.0, np.abs(bc).max())
        assert np.abs(bg - bc).max() <= 1e-3 * max(1.0, np.abs(bc).max())


def test_window_solve_parity_with_prior(gpu_ctx):
    """window A (no prior) -> prior -> window B (same trajectory, one keyframe later)."""
    opt = v.default_options()
    cfg = v.workload.config(200, 80, True)
    n = 4
    A = [v.workload.generate(v.workload.seed_for(3, 100 + i), cfg, 0.5 * i) for i in range(n)]
    Bw = [v.workload.generate(v.workload.seed_for(3, 200 + i), cfg, 0.5 * i + cfg.kf_dt) for i in range(n)]
    o.preintegrate_windows(A + Bw, opt)
    priors_c = []
    for w in [a.copy() for a in A]:
        p, _ = o.solve_window(w, opt)
        priors_c.append(p)
    # both sides start window B from the SAME prior (the oracle's), so the comparison isolates the solve
    wg, wc = [], []
    for i in range(n):
        b = Bw[i].copy()
        b.prior = priors_c[i]
        wg.append(b)
        b2 = Bw[i].copy()
        b2.prior = priors_c[i]
        wc.append(b2)
    pri_g, rep_g = gpu_ctx.solve_windows(wg, opt)
    for i in range(n):
        pri_c, rep_c = o.solve_window(wc[i], opt)
        assert rep_g[i].iterations == rep_c.iterations and rep_g[i].num_successful_steps == rep_c.num_successful_steps
        dp, dr = pose_err(wg[i], wc[i])
        assert dp <= POS_TOL and dr <= ROT_TOL, (dp, dr)
        Jg, Jc = pri_g[i].J(), pri_c.J()
        assert np.abs(Jg.T @ Jg - Jc.T @ Jc).max() <= 1e-5 * np.abs(Jc.T @ Jc).max()


def test_chained_priors_gpu_only_vs_oracle(gpu_ctx):
    """A -> B with each side using ITS OWN prior: end-to-end drift between the two paths."""
    opt = v.default_options()
    cfg = v.workload.config(200, 80, True)
    A = v.workload.generate(v.workload.seed_for(3, 300), cfg, 1.0)
    Bw = v.workload.generate(v.workload.seed_for(3, 301), cfg, 1.0 + cfg.kf_dt)
    o.preintegrate_windows([A, Bw], opt)
    ag, ac = A.copy(), A.copy()
    pg, _ = gpu_ctx.solve_windows([ag], opt)
    pc, _ = o.solve_window(ac, opt)
    bg, bc = Bw.copy(), Bw.copy()
    keep = v.Prior()
    C.memmove(C.byref(keep), C.byref(pg[0]), C.sizeof(keep))
    bg.prior = keep
    bc.prior = pc
    gpu_ctx.solve_windows([bg], opt)
    o.solve_window(bc, opt)
    dp, dr = pose_err(bg, bc)
    assert dp <= POS_TOL and dr <= ROT_TOL, (dp, dr)


def test_fixed_extrinsic_and_iteration_caps(gpu_ctx):
    ws, opt = make_windows(2, 120, 30, True, seed0=40)
    for iters, ex in [(0, 1), (1, 1), (8, 0), (5, 0)]:
        opt.num_iterations = iters
        opt.estimate_extrinsic = ex
        wg = [w.copy() for w in ws]
        wc = [w.copy() for w in ws]
        _, rep_g = gpu_ctx.solve_windows(wg, opt)
        for i in range(len(ws)):
            _, rep_c = o.solve_window(wc[i], opt)
            assert rep_g[i].iterations == rep_c.iterations
            dp, dr = pose_err(wg[i], wc[i])
            assert dp <= POS_TOL and dr <= ROT_TOL, (iters, ex, dp, dr)
            if not ex:
                assert np.abs(wg[i].ex_pose - wc[i].ex_pose).max() < 1e-12


def test_marginalization_flag_none_and_reset(gpu_ctx):
    ws, opt = make_windows(3, 60, 20, True, seed0=60)
    opt.marginalization_flag = v.capi.MARGIN_NONE
    wg = [w.copy() for w in ws]
    gpu_ctx.upload(wg, opt)
    gpu_ctx.solve()
    gpu_ctx.synchronize()
    gpu_ctx.download()
    first = [w.pose.copy() for w in wg]
    gpu_ctx.reset_state()
    gpu_ctx.solve()
    gpu_ctx.synchronize()
    gpu_ctx.download()
    for a, w in zip(first, wg):
        assert np.abs(a - w.pose).max() < 1e-9   # re-solving from the uploaded state reproduces the result
