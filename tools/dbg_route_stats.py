import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, vplines_slam_amd as v
F = v.workload.frame_stream(64)
fe = v.frontend.FrontendContext(device=0, max_images=64, width=752, height=480, max_lines=512)
fe.upload(F); fe.detect(); fe.synchronize()
import ctypes as C
tot = np.zeros(4)
for i in range(64):
    out = (C.c_ulonglong * 4)()
    fe.lib.vpl_edlines_debug_route_stats(fe.h, i, out)
    tot += np.array(out[:], float)
    if i < 4: print(i, list(out))
print('mean per frame: steps %.0f tile loads %.0f walks %.0f cycles %.3g' % tuple(tot / 64))
