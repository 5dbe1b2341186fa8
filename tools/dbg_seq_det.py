"""hash of the device's free-running sequence results (run twice: the bits must agree across processes)"""
import sys, os, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import vplines_slam_amd as v
import test_gpu_sequence as T
opt = v.default_options()
M = T.Measurements(T.NF + T.N_KEYFRAMES)
ctx = T._ctx()
dev = T.Run(T.Backend(ctx), M, opt)
for k in range(3):
    a = dev.keyframe()
    h = hashlib.sha1(); h.update(a[0].tobytes()); h.update(dev.sb.tobytes())
    hp = hashlib.sha1(); hp.update(np.ctypeslib.as_array(dev.pre[T.NF - 1].delta_p).tobytes())
    print(k, h.hexdigest()[:16], hp.hexdigest()[:16], dev.sb[3, 3:6])
