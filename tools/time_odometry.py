"""wall time of vpl_ba_solve_odometry / vpl_ba_slide_window per keyframe (one window) over the sequence of tests/test_gpu_sequence.py"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import numpy as np
import vplines_slam_amd as v
import test_gpu_sequence as T
opt = v.default_options()
M = T.Measurements(T.NF + T.N_KEYFRAMES)
ctx = T._ctx()
be = T.Backend(ctx)
acc = {"solve_odometry": [], "slide": [], "solve_odometry_c_call": []}
so, sl = be.solve_odometry, be.slide
def t_so(w, o):
    t = time.perf_counter(); r = so(w, o); acc["solve_odometry"].append(time.perf_counter() - t)
    acc["solve_odometry_c_call"].append(ctx.last_call_s)
    ms3 = (C.c_double * 3)()
    ctx.lib.vpl_ba_debug_odometry_ms(ctx.h, ms3)
    for k, nm in enumerate(("stage_triangulate", "stage_only_line_opt", "stage_solve")):
        acc.setdefault(nm, []).append(ms3[k] * 1e-3)
    return r
def t_sl(w, o):
    t = time.perf_counter(); r = sl(w, o); acc["slide"].append(time.perf_counter() - t); return r
be.solve_odometry, be.slide = t_so, t_sl
dev = T.Run(be, M, opt)
for k in range(T.N_KEYFRAMES):
    dev.keyframe()
for k, a in acc.items():
    a = np.array(a[4:]) * 1e3
    print("%s: median %.2f ms, min %.2f, max %.2f over %d keyframes (Python binding included)" % (k, np.median(a), a.min(), a.max(), len(a)))
