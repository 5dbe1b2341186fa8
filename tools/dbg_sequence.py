"""per-keyframe differences of the sequence test (device vs oracle), no asserts"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import vplines_slam_amd as v
import test_gpu_sequence as T
opt = v.default_options()
M = T.Measurements(T.NF + T.N_KEYFRAMES)
ctx = v.Context(device=0, max_windows=1, max_points=256, max_point_obs=256 * T.NF, max_lines=128, max_line_obs=128 * T.NF)
dev, orc = T.Run(T.Backend(ctx), M, opt), T.Run(T.Backend(None), M, opt)
for k in range(T.N_KEYFRAMES):
    a, b = dev.keyframe(), orc.keyframe()
    dp = np.linalg.norm(a[0][:, :3] - b[0][:, :3], axis=1).max()
    dr = max(T.rot_angle(T.quat_R(x), T.quat_R(y)) for x, y in zip(a[0][:, 3:], b[0][:, 3:]))
    same = a[1:3] == b[1:3] and a[3:6] == b[3:6] and a[6] == b[6] and a[7] == b[7]
    dd = max([abs(dev.pts[l].invd - orc.pts[l].invd) / abs(orc.pts[l].invd) for l in dev.pts if l in orc.pts] or [0])
    dl = max([np.abs(dev.lns[l].plk - orc.lns[l].plk).max() / (np.abs(orc.lns[l].plk).max() + 1e-300) for l in dev.lns if l in orc.lns and orc.lns[l].tri] or [0])
    pn = (dev.prior.n, orc.prior.n)
    dsb = np.abs(dev.sb - orc.sb).max()
    print("k %2d same %s dp %.3g dr %.3g  rel depth diff %.3g  plk diff %.3g  prior n %s  sb diff %.3g  solve %s | %s" % (
        k, same, dp, dr, dd, dl, pn, dsb, a[1:6], b[1:6]))
    if not same:
        break
