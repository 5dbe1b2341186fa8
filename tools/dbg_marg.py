import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import vplines_slam_amd as v, oracle_api as o
from test_gpu_solve import make_windows
ctx = v.Context(device=0, max_windows=64)
ws, opt = make_windows(2, 200, 0, False)
wg=[w.copy() for w in ws]; wc=[w.copy() for w in ws]
pg, rg = ctx.solve_windows(wg, opt)
dp=C.POINTER(C.c_double)
ctx.lib.vpl_ba_debug_marg_Ab.argtypes=[C.c_void_p, C.c_int, dp, dp]
for i in range(2):
    pc, rc, Ac, bc = o.solve_window(wc[i], opt, want_Ab=True)
    n=pc.n
    A=np.zeros(n*n); b=np.zeros(n)
    ctx.lib.vpl_ba_debug_marg_Ab(ctx.h, i, A.ctypes.data_as(dp), b.ctypes.data_as(dp))
    A=A.reshape(n,n)
    Jg=pg[i].J(); Jc=pc.J()
    s=np.abs(Ac).max()
    print("A gpu vs oracle", np.abs(A-Ac).max()/s, " JtJ gpu vs A gpu", np.abs(Jg.T@Jg-A).max()/s, " JtJ orc vs A orc", np.abs(Jc.T@Jc-Ac).max()/s)
    w=np.linalg.eigvalsh(Ac); print("eig min/max", w[:6], w[-1])
    wg_=np.linalg.eigvalsh(A); print("eig gpu A", wg_[:6])
    sv=np.linalg.svd(Jg,compute_uv=False); print("J0 gpu sv^2 small", np.sort(sv**2)[:6])
