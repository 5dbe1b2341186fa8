"""The front-end side of the oracle (remap, CLAHE, EDLines, LineFilter, Matching, vanishing points) under AddressSanitizer + UBSan
on frames of random size and content (CPU only; build as in tools/asan_sweep_ba.py; round 4: 4 seeds x 40 trials, clean).

    LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python tools/asan_sweep_frontend.py [seed=1] [trials=40]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/tools")
import numpy as np
import oracle_api as o
o.load("/tmp/orc_asan/liboracle.so")
from fuzz_frontend import draw_frame, SIZES
from fuzz_frontend2 import draw_maps, draw_lines
from test_preproc import oracle_clahe, oracle_remap
from test_vpdetect import CX, CY, F
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1; nt = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(seed)
for t in range(nt):
    W, H = SIZES[int(rng.integers(0, len(SIZES)))]
    if W * H > 500000: W, H = 333, 200
    a, ka = draw_frame(rng, W, H); b, kb = draw_frame(rng, W, H)
    mx, my, km = draw_maps(rng, W, H)
    a2 = oracle_clahe(oracle_remap(a, mx, my), float(rng.choice([0.5, 3.0, 40.0])), (int(rng.integers(1, 13)), int(rng.integers(1, 13))))
    prm = dict(grad_th=int(rng.integers(10, 81)), anchor_th=int(rng.integers(1, 13)), scan=int(rng.integers(1, 5)),
               min_len=int(rng.integers(8, 46)), fit_err=float(np.round(rng.uniform(1.0, 3.0), 2)))
    sm = bool(rng.integers(0, 2))
    la = o.edlines(a2, smoothed=sm, ksize=int(rng.choice([3, 5, 7])), sigma=1.0, cap_lines=8192, **prm)
    lb = o.edlines(b, smoothed=sm, ksize=5, sigma=1.2, cap_lines=8192, **prm)
    if len(la): la = o.line_filter(la, float(rng.uniform(0.5, 6)))
    if len(la) and len(lb) and min(W, H) >= 64 and len(la) <= 1024 and len(lb) <= 1024:
        o.line_match(a2, b, la, lb, o.lm_default_param(bool(rng.integers(0, 2)), bool(rng.integers(0, 2))))
    he, ae, kl = draw_lines(rng, 0)
    o.vp_detect(he, ae, np.float32(F), np.float32(CX), np.float32(CY), int(rng.integers(0, 1 << 31)), bool(rng.integers(0, 2)), full=True)
    print("trial", t, W, H, ka, kb, km, kl, len(la), len(lb), "ok", flush=True)
print("asan fe sweep seed", seed, "done")
