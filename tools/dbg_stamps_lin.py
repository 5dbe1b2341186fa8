"""Phase stamps (cycles) of k_lin on the benchmark shape.  Run on the GPU box: python tools/dbg_stamps_lin.py
(rebuilds the library with -DVPL_STAMPS first)"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib.util
spec = importlib.util.spec_from_file_location("_b", os.path.join(ROOT, "vplines-slam_amd", "_build.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
os.environ["VPL_STAMPS"] = "1"
b.build_hip(force=True)
import numpy as np
import vplines_slam_amd as v
from test_gpu_solve import make_windows
nw = int(os.environ.get("NW", "512"))
ctx = v.Context(device=0, max_windows=nw, max_points=200, max_point_obs=1200, max_lines=80, max_line_obs=480)
ws, opt = make_windows(nw, 200, 80, True)
opt.num_iterations = 1
pri, _ = ctx.solve_windows(ws, opt)
keep = []
for i, w in enumerate(ws):
    p = v.Prior(); C.memmove(C.byref(p), C.byref(pri[i]), C.sizeof(p)); keep.append(p); w.prior = p
opt.marginalization_flag = v.capi.MARGIN_NONE
opt.num_iterations = int(os.environ.get("NIT", "1"))
ctx.solve_windows(ws, opt)
ctx.lib.vpl_ba_debug_stamps.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_longlong)]
for w in (0, 1, nw // 2, nw - 1):
    out = (C.c_longlong * 64)()
    ctx.lib.vpl_ba_debug_stamps(ctx.h, w, out)
    s = list(out)
    print("window", w)
    print("   k_lin: init %d prior %d zero %d points %d lines+fold %d imu %d (raw %d) assemble %d (table wait %d, rows %d, gradient %d) | total %d" % (
        s[16] - s[17] if s[17] else 0, s[18]-s[16], s[23]-s[18], s[24]-s[23], s[25]-s[22], s[20]-s[25], s[28]-s[25], s[21]-s[20], s[26]-s[20], s[27]-s[26], s[21]-s[27], s[21]-s[16]))
    print("   point phase (wave 0): factor math %d per-track chain %d staging %d mfma %d ticket wait %d commit %d" % tuple(s[44:50]))
    print("   point phase: rounds of the 8 waves end at", s[56:64], "all joined at", s[51] - s[23], "folded at", s[50] - s[23], "sums stored at", s[55] - s[23], "phase end", s[24] - s[23])
