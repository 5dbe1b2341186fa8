"""Phase stamps (cycles) of the three step kernels k_schur / k_chol / k_back on the benchmark shape.
Run on the GPU box:  VPL_STAMPS=1 python tools/dbg_stamps_step.py   (rebuilds the library with -DVPL_STAMPS first)"""
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
import oracle_api as o
nw = int(os.environ.get("NW", "512"))
ctx = v.Context(device=0, max_windows=nw, max_points=200, max_point_obs=1200, max_lines=80, max_line_obs=480)
ws, opt = make_windows(nw, 200, 80, True)
opt.num_iterations = 1
pri, _ = ctx.solve_windows(ws, opt)
keep = []
for i, w in enumerate(ws):
    p = v.Prior(); C.memmove(C.byref(p), C.byref(pri[i]), C.sizeof(p)); keep.append(p); w.prior = p
opt.marginalization_flag = v.capi.MARGIN_NONE
ctx.solve_windows(ws, opt)
ctx.lib.vpl_ba_debug_stamps.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_longlong)]
for w in (0, 1, nw // 2, nw - 1):
    out = (C.c_longlong * 64)()
    ctx.lib.vpl_ba_debug_stamps(ctx.h, w, out)
    s = list(out)
    print("window", w)
    print("  k_schur: scaling+cauchy %d constants %d product %d tail %d | total %d" % (s[1]-s[0], s[2]-s[1], s[3]-s[2], s[4]-s[3], s[4]-s[0]))
    print("  k_chol : chains %d assemble %d cholesky %d dense back-sub %d chains back + out %d | total %d" % (s[9]-s[8], s[10]-s[9], s[11]-s[10], s[12]-s[11], s[13]-s[12], s[13]-s[8]))
    print("  k_chol assemble: loads+zero %d puts %d subtract %d" % (s[14]-s[9], s[15]-s[14], s[10]-s[15]))
    print("  k_back : landmark back-sub %d dogleg+candidate %d | total %d" % (s[6]-s[5], s[7]-s[6], s[7]-s[5]))
    print("  k_schur product, wave 0: transform %d load-issue %d mfma %d flush %d (ticket wait %d) entries %d | wave 1: %d %d %d %d (%d) %d" % tuple(s[30:42]))
