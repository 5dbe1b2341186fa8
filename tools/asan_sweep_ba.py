"""The BA side of the oracle under AddressSanitizer + UBSan on windows of random shape (CPU only; round 4: 4 seeds x 40
batches -- solve with every option, a chained solve behind its prior, NaN inputs, triangulate_lines, only_line_opt,
triangulate_points, slide_window; one finding, a zero-byte memcpy from a null pointer for a prior without rows, fixed).

    cd oracle && mkdir -p /tmp/orc_asan && for f in *.cpp; do g++ -O1 -g -std=c++17 -fPIC -pthread -fsanitize=address,undefined \
        -fno-omit-frame-pointer -c $f -o /tmp/orc_asan/${f%.cpp}.o; done
    g++ -shared -pthread -fsanitize=address,undefined -o /tmp/orc_asan/liboracle.so /tmp/orc_asan/*.o
    LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python tools/asan_sweep_ba.py [seed=1] [batches=40]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/tools")
import numpy as np
import oracle_api as o
o.load("/tmp/orc_asan/liboracle.so")
import vplines_slam_amd as v
from fuzz_parity import draw_window
import fuzz_map
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1; nb = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(seed)
for b in range(nb):
    opt = v.default_options()
    opt.num_iterations = int(rng.choice([0, 1, 2, 5]))
    opt.estimate_extrinsic = int(rng.integers(0, 2))
    opt.marginalization_flag = int(rng.choice([v.MARGIN_OLD, v.MARGIN_SECOND_NEW, v.MARGIN_NONE]))
    opt.remove_line_outliers = int(rng.integers(0, 2))
    w, sh = draw_window(rng, 100 * b + seed * 7919, 0.3 * b)
    w2, _ = draw_window(rng, 100 * b + 50 + seed * 7919, 0.3 * b + 0.1)
    raw, shr = fuzz_map.draw(rng, 100 * b + 77, 0.3 * b, True)
    o.preintegrate_windows([w, w2, raw], opt)
    if rng.random() < 0.1:
        w.pose[int(rng.integers(0, 11)), int(rng.integers(0, 7))] = np.nan
    p, r = o.solve_window(w, opt)
    if opt.marginalization_flag != v.MARGIN_NONE:
        w2.prior = p
        o.solve_window(w2, opt)
    m = raw.copy()
    nl = len(m.line_start)
    if nl:
        msk = rng.random(nl) < 0.6
        m.line_triangulated[:nl] = (~msk).astype(np.int32); m.line_plk[msk] = 0
    o.triangulate_lines(m, opt)
    o2 = v.default_options(); o2.num_iterations = int(rng.choice([1, 5])); o2.remove_line_outliers = int(rng.integers(0, 2))
    o.only_line_opt(m, o2)
    if len(m.inv_depth):
        m.inv_depth[rng.random(len(m.inv_depth)) < 0.5] = -1.0
    o.triangulate_points(m, opt, 5.0)
    for flag in (v.MARGIN_OLD, v.MARGIN_SECOND_NEW):
        o.slide_window(m.copy(), opt, flag, 5.0)
    print("batch", b, sh, shr, "it", opt.num_iterations, "flag", opt.marginalization_flag, "ok", flush=True)
print("asan sweep seed", seed, "done")
