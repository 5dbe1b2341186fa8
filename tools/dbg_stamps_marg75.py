"""phase stamps of k_marg<512> on windows with the steady-state prior (n = 75): VPL_STAMPS=1 build needed"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import vplines_slam_amd as v
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 128
opt = v.default_options()
cfg = v.workload.config(200, 80, True)
ws = [v.workload.graft_long_tracks(v.workload.seed_for(3, 50000 + i), cfg, 0.37 * i) for i in range(nw)]
ctx = v.Context(device=0, max_windows=nw, max_points=200, max_point_obs=v.workload.steady_point_obs(cfg), max_lines=80, max_line_obs=480)
v.workload.set_preintegrations(ws, ctx.preintegrate(*v.workload.imu_batch_arrays(ws), opt))
ctx.upload(ws, opt); ctx.solve(); ctx.synchronize()
ctx.upload(ws, opt, chained=True); ctx.solve(); ctx.synchronize()
_, rep = ctx.download()
print("prior n out", rep[0].prior_n)
ctx.lib.vpl_ba_debug_stamps.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_longlong)]
for w in (0, 1, nw - 1):
    out = (C.c_longlong * 64)()
    ctx.lib.vpl_ba_debug_stamps(ctx.h, w, out)
    s = list(out)
    print("window", w, "k_marg: gather %d list %d points staged %d lines staged %d elim end %d | E15 %d schur %d G-factor %d out %d | total %d" % (
        s[38] - s[32], s[39] - s[38], s[29] - s[39], s[30] - s[29], s[33] - s[30], s[34] - s[33], s[35] - s[34], s[36] - s[35], s[37] - s[36], s[37] - s[32]))
    print("   four-wave factor: cycles waiting at A | B per wave:", [(s[40 + 2 * q], s[41 + 2 * q]) for q in range(4)])
