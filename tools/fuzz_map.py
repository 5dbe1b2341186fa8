"""Randomised sweep of the map-maintenance calls around the solve (GPU box): vpl_ba_triangulate_lines, vpl_ba_only_line_opt
(+ removeLineOutlier), vpl_ba_triangulate_points, vpl_ba_slide_window on windows of random shape against the oracle, with
the bars of tests/test_line_map.py and tests/test_slide_window.py.

Shapes: 0..256 points, 0..128 lines, track lengths 1..11 (as the FeatureManager holds them: one-observation tracks, tracks
born in the newest frames), untriangulated lines, unset depths, perturbed lines, both marginalisation modes, 1..12
iterations of the line-only solve, several shapes in one batch.

    python tools/fuzz_map.py [batches=30] [windows per batch=6] [seed=1]
Exit status 1 on the first difference.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import oracle_api as o
import vplines_slam_amd as v
from vplines_slam_amd.capi import Window

NF = 11


def draw(rng, idx, t, raw):
    """raw: tracks as the FeatureManager holds them (one observation, born in the newest frame) -- what vpl_ba_slide_window
    takes; otherwise packed as for the solve (points with >= 2 observations; lines with >= 2 here, the reference's filter
    asks for LINE_MIN_OBS = 5 and the formulas hold for fewer)"""
    P = int(rng.choice([0, 1, 4, 30, 60, 200, 256]))
    L = int(rng.choice([0, 1, 2, 7, 40, 128]))
    if P == 0 and L == 0:
        L = 5
    cfg = v.workload.config(P, L, bool(rng.integers(0, 2)))
    cfg.track_len = int(rng.choice([3, 5, 6, 6, 8, 11]))
    if rng.random() < 0.5:
        cfg.pose_sigma_p = cfg.pose_sigma_theta_deg = 0.0
    # (with 2 px of noise removeLineOutlier erases 37 of 39 lines after one iteration and the line-only solve amplifies rounding
    # by ~30 x per iteration -- 1e-10 of the cost after 2 iterations, 1e-4 after 8, same surviving lines to 1e-14 on either
    # side (tools/cases/lineopt_case1.npz, tools/dbg_lineopt_case.py): not a case a cost comparison to 1e-7 can be made on)
    cfg.pix_sigma = float(rng.choice([0.1, 0.5])) / 460.0
    cfg.orth_sigma = float(rng.choice([0.0, 0.0, 0.01]))
    w = v.workload.generate(v.workload.seed_for(7, idx), cfg, t)

    def cut(start, nobs, obs, width, extra):
        off = np.concatenate([[0], np.cumsum(nobs)])
        ns, nn, no, keep = [], [], [], []
        for k in range(len(start)):
            r = rng.random()
            n, s = int(nobs[k]), int(start[k])
            if r < 0.25:
                n = int(rng.integers(1 if raw else min(2, n), n + 1))
            if raw and n == 1 and rng.random() < 0.5:
                s = NF - 1                                   # born in the newest frame
            ns.append(s); nn.append(n); no.append(obs[off[k]:off[k] + n]); keep.append(k)
        return (np.array(ns, np.int32), np.array(nn, np.int32), np.concatenate(no) if no else np.zeros((0, width)),
                [extra[k] for k in keep])
    ps, pn, po, invd = cut(w.point_start, w.point_nobs, w.point_obs, 3, list(w.inv_depth))
    ls, ln, lo, plk = cut(w.line_start, w.line_nobs, w.line_obs, 8, list(w.line_plk))
    r = Window(w.pose, w.speed_bias, w.ex_pose, ps, pn, po, np.array(invd), ls, ln, lo,
               np.array(plk).reshape(-1, 6) if plk else np.zeros((0, 6)))
    r.extra = dict(w.extra)
    return r, (P, L, cfg.track_len)


def dump_window(path, w, opt=None):
    """the arrays of a Window (+ the raw IMU samples for the pre-integration) as an .npz, for a replay off the sweep"""
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez(path, pose=w.pose, speed_bias=w.speed_bias, ex_pose=w.ex_pose, point_start=w.point_start, point_nobs=w.point_nobs,
             point_obs=w.point_obs, inv_depth=w.inv_depth, line_start=w.line_start, line_nobs=w.line_nobs, line_obs=w.line_obs,
             line_plk=w.line_plk, line_triangulated=w.line_triangulated,
             num_iterations=(opt.num_iterations if opt else 0), remove_line_outliers=(opt.remove_line_outliers if opt else 0))


def load_window(path):
    z = np.load(path)
    w = Window(z["pose"], z["speed_bias"], z["ex_pose"], z["point_start"], z["point_nobs"], z["point_obs"], z["inv_depth"],
               z["line_start"], z["line_nobs"], z["line_obs"], z["line_plk"])
    w.line_triangulated[:len(z["line_triangulated"])] = z["line_triangulated"]
    return w, int(z["num_iterations"]), int(z["remove_line_outliers"])


def fail(tag, what):
    print(tag, "DIFFERS:", what)
    return 1


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
    ctx = v.Context(device=0, max_windows=per)
    for b in range(nb):
        opt = v.default_options()
        ws, shapes, wraw = [], [], []
        for i in range(per):
            w, sh = draw(rng, 100 * b + i, 0.23 * (b * per + i), False)
            ws.append(w); shapes.append(sh)
            wraw.append(draw(rng, 100 * b + 50 + i, 0.23 * (b * per + i), True)[0])
        o.preintegrate_windows(ws + wraw, opt)
        tag = "batch %d %s" % (b, shapes)

        # ---- triangulate lines: a random subset starts untriangulated -------------------------------------------------
        t1 = [w.copy() for w in ws]
        for w in t1:
            nl = len(w.line_start)
            if nl:
                m = rng.random(nl) < 0.6
                w.line_triangulated[:nl] = (~m).astype(np.int32)
                w.line_plk[m] = 0
        g1, c1 = [w.copy() for w in t1], [w.copy() for w in t1]
        ctx.triangulate_lines(g1)
        for i, (g, c) in enumerate(zip(g1, c1)):
            o.triangulate_lines(c, opt)
            if not np.array_equal(g.line_triangulated, c.line_triangulated):
                return fail(tag, "triangulate_lines flags, window %d" % i)
            if len(c.line_plk) and np.abs(g.line_plk - c.line_plk).max() > 1e-9 * max(1e-300, np.abs(c.line_plk).max()):
                return fail(tag, "triangulate_lines plk, window %d: %.2e" % (i, np.abs(g.line_plk - c.line_plk).max()))

        # ---- line-only optimisation + outlier flags ----------------------------------------------------------------------
        opt2 = v.default_options()
        opt2.num_iterations = int(rng.choice([1, 2, 5, 12]))
        opt2.remove_line_outliers = int(rng.integers(0, 2))
        t2 = [w.copy() for w in c1]                      # the triangulated windows of the oracle side, perturbed
        for w in t2:
            if len(w.line_plk):
                w.line_plk += rng.normal(0, 0.02, w.line_plk.shape) * np.abs(w.line_plk)
        g2, c2 = [w.copy() for w in t2], [w.copy() for w in t2]
        reps = ctx.only_line_opt(g2, opt2)
        for i in range(per):
            rc = o.only_line_opt(c2[i], opt2)
            if (reps[i].iterations, reps[i].num_successful_steps, reps[i].termination) != (rc.iterations, rc.num_successful_steps, rc.termination):
                return fail(tag, "only_line_opt report, window %d: %s vs %s" % (
                    i, (reps[i].iterations, reps[i].num_successful_steps, reps[i].termination), (rc.iterations, rc.num_successful_steps, rc.termination)))
            if abs(reps[i].final_cost - rc.final_cost) > 1e-7 * max(1.0, rc.final_cost):
                sc = np.abs(c2[i].line_plk).max(axis=1, keepdims=True) + 1e-300
                d = (np.abs(g2[i].line_plk - c2[i].line_plk) / sc).max(axis=1)
                worst = np.argsort(-d)[:6]
                print("   iterations %d, outliers %d, initial cost %.10e / %.10e, accepted %d / %d" % (
                    opt2.num_iterations, opt2.remove_line_outliers, reps[i].initial_cost, rc.initial_cost, reps[i].num_successful_steps, rc.num_successful_steps))
                print("   lines %d, triangulated %d; worst lines %s rel diff %s nobs %s start %s tri %s removed %s/%s" % (
                    len(d), int(t2[i].line_triangulated[:len(d)].sum()), worst, np.array2string(d[worst], precision=1), t2[i].line_nobs[worst],
                    t2[i].line_start[worst], t2[i].line_triangulated[worst], g2[i].line_removed[worst], c2[i].line_removed[worst]))
                dump_window(os.path.join(ROOT, "gpurun_out", "fuzz_map_lineopt_case.npz"), t2[i], opt2)
                return fail(tag, "only_line_opt final cost, window %d: %.10e vs %.10e" % (i, reps[i].final_cost, rc.final_cost))
            if not np.array_equal(g2[i].line_removed, c2[i].line_removed) or reps[i].n_lines_removed != rc.n_lines_removed:
                return fail(tag, "only_line_opt removed flags, window %d" % i)
            if len(c2[i].line_plk):
                scale = np.abs(c2[i].line_plk).max(axis=1, keepdims=True) + 1e-300
                if (np.abs(g2[i].line_plk - c2[i].line_plk) / scale).max() > 1e-7:
                    return fail(tag, "only_line_opt lines, window %d: %.2e" % (i, (np.abs(g2[i].line_plk - c2[i].line_plk) / scale).max()))

        # ---- triangulate points: a random subset of the depths unset ------------------------------------------------------
        t3 = [w.copy() for w in ws]
        for w in t3:
            if len(w.inv_depth):
                w.inv_depth[rng.random(len(w.inv_depth)) < 0.5] = -1.0
        g3, c3 = [w.copy() for w in t3], [w.copy() for w in t3]
        ctx.triangulate_points(g3, 5.0)
        for i, (g, c) in enumerate(zip(g3, c3)):
            o.triangulate_points(c, opt, 5.0)
            if len(c.inv_depth) and np.abs(g.inv_depth / c.inv_depth - 1).max() > 1e-9:
                return fail(tag, "triangulate_points, window %d: %.2e" % (i, np.abs(g.inv_depth / c.inv_depth - 1).max()))

        # ---- slide window -------------------------------------------------------------------------------------------------
        t4 = [w.copy() for w in wraw]
        for w in t4:
            if len(w.inv_depth):
                w.inv_depth[rng.random(len(w.inv_depth)) < 0.3] = -1.0
        for flag in (v.MARGIN_OLD, v.MARGIN_SECOND_NEW):
            g4, c4 = [w.copy() for w in t4], [w.copy() for w in t4]
            sts = ctx.slide_window(g4, flag, 5.0)
            for i, (g, c, st) in enumerate(zip(g4, c4, sts)):
                ref = o.slide_window(c, opt, flag, 5.0)
                got = (st.point_start, st.point_nobs, st.point_drop, st.line_start, st.line_nobs, st.line_drop)
                exp = (ref.point_start, ref.point_nobs, ref.point_drop, ref.line_start, ref.line_nobs, ref.line_drop)
                if not all(np.array_equal(a, e) for a, e in zip(got, exp)):
                    return fail(tag, "slide_window(%d) tracks, window %d" % (flag, i))
                if not (np.array_equal(g.pose, c.pose) and np.array_equal(g.speed_bias, c.speed_bias)):
                    return fail(tag, "slide_window(%d) states, window %d" % (flag, i))
                if len(c.inv_depth) and not np.allclose(g.inv_depth, c.inv_depth, rtol=1e-12, atol=0):
                    return fail(tag, "slide_window(%d) depths, window %d" % (flag, i))
                if len(c.line_plk) and not np.allclose(g.line_plk, c.line_plk, rtol=1e-11, atol=1e-13):
                    return fail(tag, "slide_window(%d) lines, window %d" % (flag, i))
        print(tag, "ok (line-only solve: %d iterations, outliers %d)" % (opt2.num_iterations, opt2.remove_line_outliers))
    ctx.close()
    print("fuzz_map: %d batches of %d windows: triangulate_lines, only_line_opt, triangulate_points, slide_window identical to the oracle "
          "within the bars of tests/test_line_map.py / tests/test_slide_window.py" % (nb, per))
    return 0


if __name__ == "__main__":
    sys.exit(main())
