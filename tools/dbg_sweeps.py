import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import vplines_slam_amd as v, oracle_api as o
from test_gpu_solve import make_windows
ctx = v.Context(device=0, max_windows=64)
ws, opt = make_windows(16, 200, 80, True)
ctx.solve_windows(ws, opt)
out = (C.c_int * 16)()
ctx.lib.vpl_ba_debug_sweeps(ctx.h, out)
print("sweeps", list(out))
