"""Diagnostic (GPU box): pose parity of the HIP solve vs the oracle per config/window."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import vplines_slam_amd as v, oracle_api as o
from test_gpu_solve import make_windows, pose_err

ctx = v.Context(device=0, max_windows=64)
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for (P, L, vp) in [(200, 0, False), (200, 80, False), (200, 80, True), (0, 40, True), (37, 11, True)]:
    ws, opt = make_windows(nw, P, L, vp)
    wg = [w.copy() for w in ws]; wc = [w.copy() for w in ws]
    pg, rg = ctx.solve_windows(wg, opt)
    for i in range(nw):
        pc, rc = o.solve_window(wc[i], opt)
        dp, dr = pose_err(wg[i], wc[i])
        Jg, Jc = pg[i].J(), pc.J()
        dA = np.abs(Jg.T @ Jg - Jc.T @ Jc).max() / np.abs(Jc.T @ Jc).max()
        print("P%d L%d vp%d w%d it %d/%d succ %d/%d cost0 %.6e fin %.8e/%.8e dp %.2e dr %.2e dsb %.1e dA %.1e" % (
            P, L, vp, i, rg[i].iterations, rc.iterations, rg[i].num_successful_steps, rc.num_successful_steps,
            rc.initial_cost, rg[i].final_cost, rc.final_cost, dp, dr, np.abs(wg[i].speed_bias - wc[i].speed_bias).max(), dA))
