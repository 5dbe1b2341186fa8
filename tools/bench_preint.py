"""k_preintegrate on 10 240 keyframe intervals (512 windows x 2 x 10) of 20 samples: device time (HIP events inside the
library) and the worst deviation from the oracle on a sample."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import vplines_slam_amd as v, oracle_api as o

opt = v.default_options()
cfg = v.workload.config(8, 2, True)
ws = [v.workload.generate(v.workload.seed_for(3, i), cfg, 0.37 * i) for i in range(1024)]
ctx = v.Context(device=0, max_windows=1)
arrs = v.workload.imu_batch_arrays(ws)
ctx.preintegrate(*arrs, opt)
ctx.enable_kernel_timing(True)
t = time.perf_counter()
pre = ctx.preintegrate(*arrs, opt)
t = time.perf_counter() - t
print("intervals", len(arrs[0]), "k_preintegrate ms", ctx.kernel_times().get("k_preintegrate"), "call incl. transfers ms", 1e3 * t)
v.workload.set_preintegrations(ws[:4], pre)
wc = [w.copy() for w in ws[:4]]
o.preintegrate_windows(wc, opt)
worst = 0.0
for a, b in zip(ws[:4], wc):
    for j in range(1, 11):
        for f in ("delta_p", "delta_q", "delta_v", "jacobian", "covariance"):
            x, y = np.array(getattr(a.preint[j], f)), np.array(getattr(b.preint[j], f))
            worst = max(worst, np.abs(x - y).max() / max(1e-300, np.abs(y).max()))
print("worst relative deviation from the oracle", worst)
