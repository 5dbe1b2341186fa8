"""Diagnostic (GPU box): windows with no points and ONE line, device against oracle, iteration by iteration."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import oracle_api as o, vplines_slam_amd as v
import fuzz_parity as f
ctx = v.Context(device=0, max_windows=8)
for (P, L) in ((0, 1), (0, 2), (0, 5), (3, 1)):
    for it in (1, 2, 3, 5, 8):
        opt = v.default_options(); opt.num_iterations = it; opt.marginalization_flag = v.MARGIN_NONE
        cfg = v.workload.config(P, L, False)
        ws = [v.workload.generate(v.workload.seed_for(6, 77000 + i), cfg, 0.3 * i) for i in range(8)]
        o.preintegrate_windows(ws, opt)
        wg, wc = [w.copy() for w in ws], [w.copy() for w in ws]
        pg, rg = ctx.solve_windows(wg, opt)
        out = []
        for i in range(8):
            pc, rc = o.solve_window(wc[i], opt)
            dp, dr = f.pose_err(wg[i], wc[i])
            dl = np.abs(wg[i].line_plk / np.linalg.norm(wg[i].line_plk[:, 3:], axis=1)[:, None] - wc[i].line_plk / np.linalg.norm(wc[i].line_plk[:, 3:], axis=1)[:, None]).max()
            out.append("%d/%d %.3e/%.3e dp %.0e dl %.0e" % (rg[i].num_successful_steps, rc.num_successful_steps, rg[i].final_cost, rc.final_cost, dp, dl))
        print("P%d L%d it%d | " % (P, L, it) + " | ".join(out[:4]))
