import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import vplines_slam_amd as v, oracle_api as o
from test_gpu_solve import make_windows
nw = 512
ctx = v.Context(device=0, max_windows=nw)
ws, opt = make_windows(nw, 200, 80, True)
opt.num_iterations = 1
pri, _ = ctx.solve_windows(ws, opt)
print('prior n', pri[0].n)
# second pass WITH the priors (the benchmark's situation)
keep = []
for i, w in enumerate(ws):
    p = v.Prior(); C.memmove(C.byref(p), C.byref(pri[i]), C.sizeof(p)); keep.append(p); w.prior = p
opt.marginalization_flag = int(os.environ.get("VPL_DBG_MARG", "0"))
ctx.solve_windows(ws, opt)
ctx.lib.vpl_ba_debug_stamps.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_longlong)]
for w in (0, 1, 300):
    out = (C.c_longlong * 64)()
    ctx.lib.vpl_ba_debug_stamps(ctx.h, w, out)
    s = list(out)
    print("window", w, "k_solve phases (cycles):", [s[i + 1] - s[i] for i in range(0, 7)])
    print("   back substitution of the landmarks: cams->uc %d points %d lines %d sums %d" % (s[8] - s[5], s[9] - s[8], s[10] - s[9], s[6] - s[10]))
    print("   cholesky split: update(a) %d diag(b) %d (16 steps alone: %d) trsm(c) %d" % (s[40], s[41], s[43], s[42]))
    print("   schur chunk loop (thread 0): stage %d mfma %d barrier wait %d" % (s[52], s[53], s[54]))
    print("   point phase (wave 0): factor math %d per-factor terms %d staging %d mfma %d ticket wait %d commit %d" % tuple(s[44:50]))
    print("   point phase: rounds of the 8 waves end at", s[56:64], "all joined at", s[51] - s[23], "folded at", s[50] - s[23], "sums stored at", s[55] - s[23], "phase end", s[24] - s[23])
    print("   k_lin: prior %d zero %d points %d lines+fold %d imu %d (raw %d) assemble %d (table wait %d, rows %d, gradient %d)" % (s[18]-s[16], s[23]-s[18], s[24]-s[23], s[25]-s[22], s[20]-s[25], s[28]-s[25], s[21]-s[20], s[26]-s[20], s[27]-s[26], s[21]-s[27]))
    print("   k_marg elimination: gather %d list %d points staged %d (+product) lines staged %d end %d" % (s[38]-s[32], s[39]-s[38], s[29]-s[39], s[30]-s[29], s[33]-s[30]))
    print("   k_marg: setup+Ad %d landmark-elim %d E15 %d schur %d G-factor %d out %d" % (s[33]-s[32], 0, s[34]-s[33], s[35]-s[34], s[36]-s[35], s[37]-s[36]))
