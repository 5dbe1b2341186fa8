"""Diagnostic (GPU box): one dumped window of tools/fuzz_map.py through vpl_ba_only_line_opt for 0..12 iterations, device vs oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import oracle_api as o, vplines_slam_amd as v
import fuzz_map as f
w, it, rem = f.load_window(sys.argv[1])
ctx = v.Context(device=0, max_windows=2)
for iters in range(0, 13):
    opt = v.default_options(); opt.num_iterations = iters; opt.remove_line_outliers = 0
    g, c = w.copy(), w.copy()
    rg = ctx.only_line_opt([g], opt)[0]
    rc = o.only_line_opt(c, opt)
    sc = np.abs(c.line_plk).max(axis=1, keepdims=True) + 1e-300
    d = (np.abs(g.line_plk - c.line_plk) / sc).max(axis=1)
    print("iters %2d | it %d/%d succ %d/%d term %d/%d | final %.10e / %.10e | worst line %d: %.1e (nobs %d)" % (
        iters, rg.iterations, rc.iterations, rg.num_successful_steps, rc.num_successful_steps, rg.termination, rc.termination,
        rg.final_cost, rc.final_cost, int(d.argmax()), d.max(), w.line_nobs[int(d.argmax())]))
print("---- cost of the OUTPUT lines, re-evaluated with a 0-iteration call on either side")
for iters in (1, 2, 4, 8):
    opt = v.default_options(); opt.num_iterations = iters; opt.remove_line_outliers = 0
    g, c = w.copy(), w.copy()
    rg = ctx.only_line_opt([g], opt)[0]; rc = o.only_line_opt(c, opt)
    opt0 = v.default_options(); opt0.num_iterations = 0; opt0.remove_line_outliers = 0
    gg, gc, cg, cc = g.copy(), g.copy(), c.copy(), c.copy()
    a = ctx.only_line_opt([gg], opt0)[0].initial_cost; b = o.only_line_opt(gc, opt0).initial_cost
    d = ctx.only_line_opt([cg], opt0)[0].initial_cost; e = o.only_line_opt(cc, opt0).initial_cost
    print("iters %d: reported final dev %.10e orc %.10e | cost(dev lines) by dev %.10e by orc %.10e | cost(orc lines) by dev %.10e by orc %.10e" % (iters, rg.final_cost, rc.final_cost, a, b, d, e))
    # per line: which lines changed at all
    moved_g = np.abs(g.line_plk - w.line_plk).max(axis=1) > 0; moved_c = np.abs(c.line_plk - w.line_plk).max(axis=1) > 0
    print("        lines moved: dev %d orc %d of %d (triangulated %d); not moved on one side only: %s" % (moved_g.sum(), moved_c.sum(), len(moved_g), int(w.line_triangulated[:len(moved_g)].sum()), np.nonzero(moved_g != moved_c)[0]))
