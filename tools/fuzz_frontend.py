"""Randomised sweep of the line front-end (GPU box): EDLines (+ the Gaussian pre-blur, + LineFilter) and the KLT line matcher on
frames of random SIZE and CONTENT with random PARAMETERS, every stage against the oracle, bit for bit.

Frames: the reference's MH_04 frames resampled to the drawn size (nearest neighbour), mirrored, warped; uniform noise; noise
smoothed to blobs; a flat frame; a frame of straight bars at random angles; a frame with one saturated half.
Parameters: gradient threshold 10..80, anchor threshold 1..12, scan interval 1..4, minimum line length 8..45, fit error 1..3,
smoothed on / off with kernel sizes 3 / 5 / 7 / automatic and sigma 0.6..2, both roundings of the blur's taps.
Matcher: pairs (frame, warped frame) with the detector's lines, illumination adaptation / topological filter on or off.

    python tools/fuzz_frontend.py [trials=40] [seed=1]
Exit status 1 on the first difference (after printing which stage of which trial).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import oracle_api as o
import vplines_slam_amd as v

FRAMES = np.load(os.path.join(ROOT, "tests", "golden", "mh04_frames.npz"))["frames"]
SIZES = [(752, 480), (640, 480), (320, 240), (97, 61), (128, 128), (333, 200), (1024, 768), (500, 31), (33, 400), (848, 480)]


def resample(img, W, H, flip=0):
    ys = (np.arange(H) * img.shape[0] / H).astype(int)
    xs = (np.arange(W) * img.shape[1] / W).astype(int)
    a = img[ys][:, xs]
    if flip & 1:
        a = a[::-1]
    if flip & 2:
        a = a[:, ::-1]
    return np.ascontiguousarray(a)


def box_blur(a, r):
    a = a.astype(np.float64)
    k = 2 * r + 1
    for ax in (0, 1):
        c = np.cumsum(np.concatenate([np.zeros_like(np.take(a, [0], ax)), a], ax), ax)
        n = a.shape[ax]
        hi = np.clip(np.arange(n) + r + 1, 0, n)
        lo = np.clip(np.arange(n) - r, 0, n)
        a = (np.take(c, hi, ax) - np.take(c, lo, ax)) / k
    return a


def draw_frame(rng, W, H):
    kind = int(rng.integers(0, 7))
    if kind <= 2:
        return resample(FRAMES[int(rng.integers(0, len(FRAMES)))], W, H, int(rng.integers(0, 4))), "mh04"
    if kind == 3:
        return rng.integers(0, 256, (H, W), dtype=np.uint8), "noise"
    if kind == 4:
        b = box_blur(rng.integers(0, 256, (H, W)).astype(np.float64), int(rng.integers(1, 6)))
        b = (b - b.min()) / max(1e-9, b.max() - b.min())
        return np.ascontiguousarray((255 * b).astype(np.uint8)), "blobs"
    if kind == 5:
        a = np.full((H, W), int(rng.integers(0, 256)), np.uint8)
        yy, xx = np.mgrid[0:H, 0:W]
        for _ in range(int(rng.integers(1, 12))):
            th = rng.uniform(0, np.pi)
            d = (xx - rng.uniform(0, W)) * np.cos(th) + (yy - rng.uniform(0, H)) * np.sin(th)
            a[np.abs(d) < rng.uniform(1, 8)] = int(rng.integers(0, 256))
        return a, "bars"
    a = np.zeros((H, W), np.uint8)
    if rng.random() < 0.5:
        a[:, W // 2:] = 255
        a[H // 3, :] = 128
    return a, "flat/half"


def canon(lines):
    if len(lines) == 0:
        return lines
    key = np.lexsort((np.round(lines[:, 3], 2), np.round(lines[:, 2], 2), np.round(lines[:, 1], 2), np.round(lines[:, 0], 2)))
    return lines[key]


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctxs = {}
    for t in range(trials):
        W, H = SIZES[int(rng.integers(0, len(SIZES)))]
        n = int(rng.integers(1, 4))
        frames, kinds = zip(*[draw_frame(rng, W, H) for _ in range(n)])
        imgs = np.stack(frames)
        prm = dict(grad_th=int(rng.integers(10, 81)), anchor_th=int(rng.integers(1, 13)), scan=int(rng.integers(1, 5)),
                   min_len=int(rng.integers(8, 46)), fit_err=float(np.round(rng.uniform(1.0, 3.0), 2)))
        smoothed = bool(rng.integers(0, 2))
        ksize = int(rng.choice([3, 5, 7, 0]))
        # (automatic size = cvRound(6 sigma + 1) | 1; the device holds a blur radius of up to 3 and refuses larger kernels)
        sigma = float(np.round(rng.uniform(0.6, 1.05 if ksize == 0 else 2.0), 2))
        mode = int(rng.integers(0, 2))
        if (W, H) not in ctxs:
            ctxs[(W, H)] = v.frontend.FrontendContext(device=0, max_images=3, width=W, height=H, max_lines=4096)
            ctxs[(W, H)].match_reserve(3, 16384)
        fe = ctxs[(W, H)]
        p = v.frontend.default_param()
        p.gradientThreshold, p.anchorThreshold, p.scanIntervals = prm["grad_th"], prm["anchor_th"], prm["scan"]
        p.minLineLen, p.lineFitErrThreshold, p.ksize, p.sigma = prm["min_len"], prm["fit_err"], ksize, sigma
        fe.set_blur_kernel(mode)
        tag = "trial %d %dx%d %s %s smoothed=%d k%d s%.2f m%d" % (t, W, H, kinds, prm, smoothed, ksize, sigma, mode)
        try:
            out = fe.detect_batch(imgs, p, smoothed=smoothed)
        except RuntimeError as e:
            # a refusal (capacity) must be the oracle's overflow too; everything else is a failure
            print(tag, "REFUSED:", e)
            if "CAPACITY" in str(e).upper() or "capacity" in str(e) or "exceeds" in str(e) or "more " in str(e):
                continue
            return 1
        nl = []
        for i in range(n):
            lo, st = o.edlines(imgs[i], want_stages=True, smoothed=smoothed, ksize=ksize, sigma=sigma, blur_mode=mode,
                               cap_lines=8192, **prm)
            sg = fe.debug_stage(i)
            for k in ("dx", "dy", "g", "dir", "anchors", "sid", "chain_x", "chain_y"):
                if not np.array_equal(sg[k], st[k]):
                    print(tag, "frame", i, "STAGE", k, "differs"); return 1
            if len(out[i]) != len(lo):
                print(tag, "frame", i, "line count", len(out[i]), len(lo)); return 1
            if len(lo) and (np.abs(out[i][:, :4] - lo[:, :4]).max() > 1e-3 or np.abs(out[i][:, 4:7] - lo[:, 4:7]).max() > 1e-9):
                print(tag, "frame", i, "lines differ (in list order)"); return 1
            nl.append(len(lo))
        # LineFilter on what was detected
        dth = float(np.round(rng.uniform(0.5, 6.0), 1))
        if max(nl) > 0:
            kept = fe.line_filter_batch(out, dth)
            for i in range(n):
                ko = o.line_filter(out[i], dth) if len(out[i]) else out[i]
                if len(kept[i]) != len(ko) or (len(ko) and not np.array_equal(kept[i][:, :4].astype(np.float32), ko[:, :4].astype(np.float32))):
                    print(tag, "frame", i, "LineFilter(%.1f) differs: %d / %d" % (dth, len(kept[i]), len(ko))); return 1
        # matcher: frame 0 against the other frames of the trial (whatever they are), the detector's lines
        mp = v.frontend.default_match_param(bool(rng.integers(0, 2)), bool(rng.integers(0, 2)))
        pairs = [(0, j) for j in range(1, n) if nl[0] and nl[j] and nl[0] <= 1024 and nl[j] <= 1024]
        if pairs and min(W, H) >= 64:
            r2c, ok = fe.match_batch(imgs, pairs, [out[0]] * len(pairs), [out[j] for _, j in pairs], mp)
            for q, (_, j) in enumerate(pairs):
                oko, ro, _ = o.line_match(imgs[0], imgs[j], out[0], out[j], o.lm_default_param(bool(mp.illumination_adapt), bool(mp.topological_filter)))
                if bool(ok[q]) != bool(oko) or (oko and not np.array_equal(r2c[q][:len(ro)], ro)):
                    print(tag, "pair", (0, j), "matches differ: ok %s/%s, %d / %d matched" % (ok[q], oko, int((r2c[q] >= 0).sum()), int((ro >= 0).sum()))); return 1
        print(tag, "lines", nl, "ok")
    for fe in ctxs.values():                           # (VPL_DEBUG_GUARDS=1: close() checks the pads behind the device arrays)
        fe.close()
    print("fuzz_frontend: %d trials, every stage, line list, LineFilter and match identical to the oracle" % trials)
    return 0


if __name__ == "__main__":
    sys.exit(main())
