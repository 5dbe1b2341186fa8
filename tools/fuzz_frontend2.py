"""Randomised sweep of the two remaining front-end stages (GPU box), bit for bit against the oracle:

* image preparation of readImage (cv::remap + CLAHE, line_feature_tracker.cpp:62-68): frame sizes that are no multiple of
  4 or of the tile grid, maps of four kinds (EuRoC's undistortion, an affine map, a wavy map, coordinates that leave the
  frame / land exactly on its border), clip limits 0.5..40, tile grids 1..12 x 1..12, CLAHE on / off, maps on / off;
* vanishing-point detection (vanishing_point_detection.cpp): Manhattan scenes with pixel noise 0..3, random segments, three
  lines, parallel lines only (no hypothesis), one line, hypotheses from a subset, random seeds, first-frame flag on / off.

    python tools/fuzz_frontend2.py [trials=40] [seed=1]
Exit status 1 on the first difference.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import oracle_api as o
import vplines_slam_amd as v
from test_preproc import euroc_maps, oracle_clahe, oracle_remap
from test_vpdetect import CX, CY, F, manhattan_lines

FRAMES = np.load(os.path.join(ROOT, "tests", "golden", "mh04_frames.npz"))["frames"]
SIZES = [(752, 480), (301, 203), (640, 480), (97, 61), (128, 128), (333, 201), (50, 37)]


def draw_maps(rng, W, H):
    kind = int(rng.integers(0, 5))
    u, vv = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
    if kind == 0:
        return euroc_maps(W, H, W / 752.0) + ("euroc",)
    if kind == 1:
        a, b = rng.uniform(0.9, 1.1, 2)
        return (u * np.float32(a) + np.float32(rng.uniform(-5, 5)), vv * np.float32(b) + np.float32(rng.uniform(-5, 5)), "affine")
    if kind == 2:
        return ((u + 3 * np.sin(vv / 17)).astype(np.float32), (vv + 2 * np.cos(u / 23)).astype(np.float32), "wavy")
    if kind == 3:
        return (rng.uniform(-40, W + 40, (H, W)).astype(np.float32), rng.uniform(-40, H + 40, (H, W)).astype(np.float32), "anywhere")
    mx, my = u.copy(), vv.copy()                                   # identity with coordinates exactly on / just off the border
    mx[:, 0], mx[:, -1], my[0, :], my[-1, :] = -1.0, W - 1.0, -0.5, H - 0.5
    return mx, my, "border"


def draw_lines(rng, k):
    kind = int(rng.integers(0, 6))
    if kind <= 1:
        n = tuple(int(x) for x in rng.integers(1, 40, 3))
        ends, _, _ = manhattan_lines(int(rng.integers(0, 1 << 30)), n_per_dir=n, noise=float(rng.uniform(0, 3)))
        return ends, ends, "manhattan%s" % (n,)
    if kind == 2:
        n = int(rng.integers(2, 200))
        e = np.stack([rng.uniform(0, 751, n), rng.uniform(0, 479, n), rng.uniform(0, 751, n), rng.uniform(0, 479, n)], 1).astype(np.float32)
        return e, e, "random%d" % n
    if kind == 3:
        par = np.array([[10, 10 + 5 * j, 300, 10 + 5 * j] for j in range(int(rng.integers(2, 9)))], np.float32)
        return par, par, "parallel"
    if kind == 4:
        ends, _, _ = manhattan_lines(int(rng.integers(0, 1 << 30)), noise=0.5)
        m = int(rng.integers(1, 4))
        return ends[:m], ends, "hyp%d" % m
    ends, _, _ = manhattan_lines(int(rng.integers(0, 1 << 30)), noise=1.0)
    return ends[::3], ends, "subset"


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctxs = {}
    for t in range(trials):
        # ---- image preparation ---------------------------------------------------------------------------------------------
        W, H = SIZES[int(rng.integers(0, len(SIZES)))]
        if (W, H) not in ctxs:
            ctxs[(W, H)] = v.frontend.FrontendContext(device=0, max_images=3, width=W, height=H, max_lines=512)
        fe = ctxs[(W, H)]
        n = int(rng.integers(1, 4))
        raw = []
        for _ in range(n):
            f = FRAMES[int(rng.integers(0, len(FRAMES)))]
            ys = (np.arange(H) * f.shape[0] / H).astype(int); xs = (np.arange(W) * f.shape[1] / W).astype(int)
            a = f[ys][:, xs]
            if rng.random() < 0.2:
                a = rng.integers(0, 256, (H, W), dtype=np.uint8)
            elif rng.random() < 0.1:
                a = np.full((H, W), int(rng.integers(0, 256)), np.uint8)
            raw.append(np.ascontiguousarray(a))
        raw = np.stack(raw)
        use_maps = rng.random() < 0.8
        mx = my = None
        kind = "none"
        if use_maps:
            mx, my, kind = draw_maps(rng, W, H)
        eq = bool(rng.integers(0, 2))
        clip = float(np.round(rng.choice([0.5, 1.0, 3.0, 3.0, 8.0, 40.0]), 1))
        tiles = (int(rng.integers(1, 13)), int(rng.integers(1, 13)))
        tag = "trial %d prep %dx%d n%d maps=%s clahe=%d clip %.1f tiles %s" % (t, W, H, n, kind, eq, clip, tiles)
        fe.set_maps(mx, my)
        fe.pre_upload(raw)
        try:
            fe.pre_run(eq, clip, tiles)
        except RuntimeError as e:
            print(tag, "REFUSED:", e)
            return 1
        got = fe.pre_download()
        for i in range(n):
            exp = raw[i]
            if use_maps:
                exp = oracle_remap(exp, mx, my)
            if eq:
                exp = oracle_clahe(exp, clip, tiles)
            if not np.array_equal(got[i], exp):
                bad = np.argwhere(got[i] != exp)
                print(tag, "frame", i, "DIFFERS in %d pixels, first %s: %d vs %d" % (len(bad), bad[0], got[i][tuple(bad[0])], exp[tuple(bad[0])]))
                return 1
        print(tag, "ok")

        # ---- vanishing points ------------------------------------------------------------------------------------------------
        if (752, 480) not in ctxs:
            ctxs[(752, 480)] = v.frontend.FrontendContext(device=0, max_images=3, width=752, height=480, max_lines=512)
        fv = ctxs[(752, 480)]
        k = int(rng.integers(1, 4))
        cases = [draw_lines(rng, j) + (int(rng.integers(0, 1 << 31)), int(rng.integers(0, 2))) for j in range(k)]
        tag = "trial %d vp %s" % (t, [(c[2], c[3], c[4]) for c in cases])
        vps, ids, status = fv.vp_detect([c[0] for c in cases], [c[1] for c in cases], F, CX, CY, [c[3] for c in cases], [c[4] for c in cases])
        for j, (he, ae, _, seed, first) in enumerate(cases):
            wv, wi, it, dbg = o.vp_detect(he, ae, np.float32(F), np.float32(CX), np.float32(CY), seed, bool(first), full=True)
            if it < 0:
                if not (status[j] == -1 and not vps[j].any() and (ids[j] == 3).all()):
                    print(tag, "case", j, "DIFFERS: the oracle finds no hypothesis, the device status", status[j]); return 1
                continue
            grid, pairs, best, drawn = fv.vp_debug(j)
            if status[j] != 0 or not np.array_equal(pairs, dbg["pairs"]) or drawn != dbg["drawn"][0]:
                print(tag, "case", j, "DIFFERS: pairs drawn"); return 1
            if not np.array_equal(grid, dbg["grid"]):
                print(tag, "case", j, "DIFFERS: sphere grid, %d cells" % int((grid != dbg["grid"]).sum())); return 1
            if best != dbg["best_idx"] or not np.array_equal(vps[j], wv) or not np.array_equal(ids[j], wi):
                print(tag, "case", j, "DIFFERS: winning hypothesis / VPs / line classes"); return 1
        print(tag, "ok")
    for fe in ctxs.values():                           # (VPL_DEBUG_GUARDS=1: close() checks the pads behind the device arrays)
        fe.close()
    print("fuzz_frontend2: %d trials: prepared frames, sphere grids, vanishing points and line classes identical to the oracle" % trials)
    return 0


if __name__ == "__main__":
    sys.exit(main())
