"""Diagnostic (GPU box): one batch of tools/fuzz_parity.py again, the chained stage with the priors swapped between the sides.
    python tools/dbg_fuzz_case.py <batch> [seed=1] [per=8] [second_new]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import ctypes as C
import numpy as np
import oracle_api as o, vplines_slam_amd as v
import fuzz_parity as f
from vplines_slam_amd.capi import Prior

target = int(sys.argv[1]); seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1; per = int(sys.argv[3]) if len(sys.argv) > 3 else 8
rng = np.random.default_rng(seed)
ctx = v.Context(device=0, max_windows=per)
np.set_printoptions(precision=3, linewidth=200)
for b in range(target + 1):
    opt = v.default_options()
    opt.num_iterations = int(rng.choice([1, 2, 3, 5, 5, 8]))
    opt.estimate_extrinsic = int(rng.integers(0, 2))
    opt.marginalization_flag = int(rng.choice([v.MARGIN_OLD, v.MARGIN_OLD, v.MARGIN_SECOND_NEW, v.MARGIN_NONE]))
    chained = opt.marginalization_flag != v.MARGIN_NONE and rng.random() < 0.5
    ws, shapes = [], []
    for i in range(per):
        w, sh = f.draw_window(rng, 1000 * b + i, 0.37 * (b * per + i)); ws.append(w); shapes.append(sh)
    if b < target:
        continue
    o.preintegrate_windows(ws, opt)
    wg, wc = [w.copy() for w in ws], [w.copy() for w in ws]
    pg, rg = ctx.solve_windows(wg, opt)
    keep = (Prior * per)(); C.memmove(keep, pg, C.sizeof(keep))
    pcs = [o.solve_window(wc[i], opt)[0] for i in range(per)]
    second_new = len(sys.argv) > 4 and sys.argv[4] == "second_new"
    if second_new:     # stage 2 of the sweep: the same frames again behind the prior, MARGIN_SECOND_NEW
        w2 = [f.draw_window(np.random.default_rng(9000 + 100 * b + i), 1000 * b + 700 + i, 0.37 * (b * per + i))[0] for i in range(per)]
        opt0 = opt
        opt = v.default_options()
        opt.num_iterations, opt.estimate_extrinsic, opt.marginalization_flag = opt0.num_iterations, opt0.estimate_extrinsic, v.MARGIN_SECOND_NEW
    else:
        w2 = [f.draw_window(np.random.default_rng(7000 + 100 * b + i), 1000 * b + 500 + i, 0.37 * (b * per + i) + 0.1)[0] for i in range(per)]
    o.preintegrate_windows(w2, opt)
    def run_dev(priors):
        g = [w.copy() for w in w2]
        for i in range(per): g[i].prior = priors[i] if priors[i].n > 0 else None
        ctx.solve_windows(g, opt); return g
    def run_orc(priors):
        c = [w.copy() for w in w2]
        for i in range(per):
            c[i].prior = priors[i] if priors[i].n > 0 else None
            o.solve_window(c[i], opt)
        return c
    dd, do_, od, oo = run_dev(keep), run_dev(pcs), run_orc(keep), run_orc(pcs)
    for i in range(per):
        Jg, Jc = keep[i].J(), pcs[i].J()
        Ag, Ac = Jg.T @ Jg, Jc.T @ Jc
        bg, bc = Jg.T @ keep[i].r(), Jc.T @ pcs[i].r()
        lam = np.linalg.eigvalsh(0.5 * (Ac + Ac.T))
        print("w%d %s n %d/%d | stage0 dp %.1e | dev(own) vs orc(own) %.1e | dev(orc prior) vs orc(orc prior) %.1e | dev(dev prior) vs orc(dev prior) %.1e | "
              "orc(dev prior) vs orc(orc prior) %.1e | dA %.1e db %.1e | eig min %.1e max %.1e, < 1e-6 max: %d" % (
                  i, shapes[i], keep[i].n, pcs[i].n, f.pose_err(wg[i], wc[i])[0], f.pose_err(dd[i], oo[i])[0], f.pose_err(do_[i], oo[i])[0],
                  f.pose_err(dd[i], od[i])[0], f.pose_err(od[i], oo[i])[0], np.abs(Ag - Ac).max() / np.abs(Ac).max(),
                  np.abs(bg - bc).max() / max(1.0, np.abs(bc).max()), lam[0], lam[-1], int((lam < 1e-6 * lam[-1]).sum())))
        x0g = np.ctypeslib.as_array(keep[i].x0)[:keep[i].n_blocks]; x0c = np.ctypeslib.as_array(pcs[i].x0)[:pcs[i].n_blocks]
        print("     x0 max diff %.2e   blocks %s" % (np.abs(x0g - x0c).max(), list(zip(keep[i].block_kind[:keep[i].n_blocks], keep[i].block_frame[:keep[i].n_blocks]))[:6]))
