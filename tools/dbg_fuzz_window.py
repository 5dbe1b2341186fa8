"""Diagnostic (GPU box): one first-stage window of tools/fuzz_parity.py, device against oracle for 1..N iterations.
    python tools/dbg_fuzz_window.py <batch> <window> [seed=1] [per=8]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import oracle_api as o, vplines_slam_amd as v
import fuzz_parity as f
target, wi = int(sys.argv[1]), int(sys.argv[2])
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1; per = int(sys.argv[4]) if len(sys.argv) > 4 else 8
rng = np.random.default_rng(seed)
for b in range(target + 1):
    opt = v.default_options()
    opt.num_iterations = int(rng.choice([1, 2, 3, 5, 5, 8]))
    opt.estimate_extrinsic = int(rng.integers(0, 2))
    opt.marginalization_flag = int(rng.choice([v.MARGIN_OLD, v.MARGIN_OLD, v.MARGIN_SECOND_NEW, v.MARGIN_NONE]))
    chained = opt.marginalization_flag != v.MARGIN_NONE and rng.random() < 0.5
    ws = [f.draw_window(rng, 1000 * b + i, 0.37 * (b * per + i))[0] for i in range(per)]
w = ws[wi]
o.preintegrate_windows([w], opt)
np.savez(os.path.join(ROOT, "gpurun_out", "fuzz_window_b%d_w%d_s%d.npz" % (target, wi, seed)), pose=w.pose, speed_bias=w.speed_bias, ex_pose=w.ex_pose,
         point_start=w.point_start, point_nobs=w.point_nobs, point_obs=w.point_obs, inv_depth=w.inv_depth, line_start=w.line_start,
         line_nobs=w.line_nobs, line_obs=w.line_obs, line_plk=w.line_plk, line_triangulated=w.line_triangulated,
         imu=w.extra["imu_samples"], acc0=w.extra["imu_acc0"], gyr0=w.extra["imu_gyr0"],
         opt=np.array([opt.num_iterations, opt.estimate_extrinsic, opt.marginalization_flag]))
ctx = v.Context(device=0, max_windows=1)
for it in range(1, opt.num_iterations + 1):
    op = v.default_options(); op.num_iterations = it; op.estimate_extrinsic = opt.estimate_extrinsic; op.marginalization_flag = v.MARGIN_NONE
    g, c = w.copy(), w.copy()
    _, rg = ctx.solve_windows([g], op)
    _, rc = o.solve_window(c, op)
    print("iterations %d: it %d/%d succ %d/%d term %d/%d cost %.10e / %.10e dp %.2e" % (
        it, rg[0].iterations, rc.iterations, rg[0].num_successful_steps, rc.num_successful_steps, rg[0].termination, rc.termination,
        rg[0].final_cost, rc.final_cost, f.pose_err(g, c)[0]))
