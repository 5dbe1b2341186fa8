"""Diagnostic (GPU box): vpl_ba_only_line_opt against the oracle for line tracks of 6 / 8 / 11 observations."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_api as o, vplines_slam_amd as v
ctx = v.Context(device=0, max_windows=4)
rng = np.random.default_rng(3)
for TL in (6, 8, 11):
    for iters in (1, 2, 5):
        opt = v.default_options(); opt.num_iterations = iters; opt.remove_line_outliers = 1
        cfg = v.workload.config(60, 40, True); cfg.track_len = TL; cfg.pix_sigma = 0.5 / 460.0
        ws = [v.workload.generate(v.workload.seed_for(7, 900 + i), cfg, 0.2 * i) for i in range(4)]
        o.preintegrate_windows(ws, opt)
        for w in ws:
            w.line_plk += np.random.default_rng(5).normal(0, 0.02, w.line_plk.shape) * np.abs(w.line_plk)
        wg, wc = [w.copy() for w in ws], [w.copy() for w in ws]
        reps = ctx.only_line_opt(wg, opt)
        out = []
        for i in range(4):
            rc = o.only_line_opt(wc[i], opt)
            sc = np.abs(wc[i].line_plk).max(axis=1, keepdims=True) + 1e-300
            d = (np.abs(wg[i].line_plk - wc[i].line_plk) / sc).max(axis=1)
            out.append("it %d/%d succ %d/%d cost0 %.6e/%.6e fin %.8e/%.8e dl %.1e (line %d) rem %d/%d" % (
                reps[i].iterations, rc.iterations, reps[i].num_successful_steps, rc.num_successful_steps, reps[i].initial_cost, rc.initial_cost,
                reps[i].final_cost, rc.final_cost, d.max(), int(d.argmax()), reps[i].n_lines_removed, rc.n_lines_removed))
        print("TL %d iters %d | " % (TL, iters) + " | ".join(out[:2]))
