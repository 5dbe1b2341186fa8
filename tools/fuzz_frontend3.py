"""Randomised sweep of the front-end's DEVICE-RESIDENT chain (GPU box) -- what bench.py's config 4 / 5 and a tracker loop run:

    vpl_pre_upload -> vpl_pre_run -> vpl_edlines_detect_ex -> [vpl_line_filter_detected] -> vpl_match_from_detected ->
    vpl_match_run -> vpl_edlines_download / vpl_match_download

on ONE context per frame size that lives through all trials, with the number of frames (1 .. capacity), the pairs (any two
frames, repeated frames, a frame paired with itself), the cut `max_lines`, the detector's and matcher's parameters, remap /
CLAHE on or off, the LineFilter step and the stream (switched between trials and between two enqueued stages) drawn per trial -- so every trial runs on whatever the one before left in the frame
batch, the line table, the pyramids and the matcher's buffers.  Held, bit for bit, to the oracle run stage by stage on the
host (remap + CLAHE -> EDLines -> LineFilter -> Matching on the first `max_lines` lines of either frame).

    python tools/fuzz_frontend3.py [trials=40] [seed=1]
Exit status 1 on the first difference.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ctypes as C

import numpy as np

import oracle_api as o
import vplines_slam_amd as v
from fuzz_frontend import draw_frame, resample, FRAMES
from test_preproc import euroc_maps, oracle_clahe, oracle_remap

SIZES = [(752, 480), (640, 480), (320, 240), (333, 200), (128, 128)]
CAP = 6


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctxs = {}
    import torch
    keep = [torch.cuda.Stream(device=0) for _ in range(3)]
    streams = [0] + [k.cuda_stream for k in keep]
    done = dict(pairs=0, matched=0, filtered=0, prepared=0)
    for t in range(trials):
        W, H = SIZES[int(rng.integers(0, len(SIZES)))]
        if (W, H) not in ctxs:
            ctxs[(W, H)] = v.frontend.FrontendContext(device=0, max_images=CAP, width=W, height=H, max_lines=1024)
            ctxs[(W, H)].match_reserve(CAP, 16384)
        fe = ctxs[(W, H)]
        n = int(rng.integers(1, CAP + 1))
        # consecutive frames of the reference's stream (so that the matcher has something to find), or anything
        if rng.random() < 0.7:
            k0 = int(rng.integers(0, len(FRAMES) - n + 1))
            raw = np.stack([resample(FRAMES[k0 + i], W, H) for i in range(n)])
            kinds = "mh04[%d:%d]" % (k0, k0 + n)
        else:
            fr = [draw_frame(rng, W, H) for _ in range(n)]
            raw = np.stack([f[0] for f in fr]); kinds = tuple(f[1] for f in fr)
        prep = rng.random() < 0.6
        use_maps, eq = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        clip, tiles = float(rng.choice([1.0, 3.0, 8.0])), (int(rng.integers(1, 10)), int(rng.integers(1, 10)))
        p = v.frontend.default_param()
        prm = dict(grad_th=int(rng.integers(15, 61)), anchor_th=int(rng.integers(2, 10)), scan=int(rng.integers(1, 4)),
                   min_len=int(rng.integers(12, 40)), fit_err=float(np.round(rng.uniform(1.0, 2.5), 2)))
        smoothed = bool(rng.integers(0, 2))
        ksize, sigma = int(rng.choice([3, 5, 7])), float(np.round(rng.uniform(0.8, 1.6), 2))
        p.gradientThreshold, p.anchorThreshold, p.scanIntervals = prm["grad_th"], prm["anchor_th"], prm["scan"]
        p.minLineLen, p.lineFitErrThreshold, p.ksize, p.sigma = prm["min_len"], prm["fit_err"], ksize, sigma
        dth = float(np.round(rng.uniform(1.0, 5.0), 1)) if rng.random() < 0.5 else None
        cut = int(rng.choice([16, 64, 200, 1024]))
        npairs = int(rng.integers(0, CAP + 1))
        pairs = [(int(rng.integers(0, n)), int(rng.integers(0, n))) for _ in range(npairs)]
        if n > 1 and npairs and rng.random() < 0.6:
            pairs = [(i, i + 1) for i in range(min(npairs, n - 1))]
        mp = v.frontend.default_match_param(bool(rng.integers(0, 2)), bool(rng.integers(0, 2)))
        tag = "trial %d %dx%d n%d %s prep=%d(maps %d clahe %d clip %.0f tiles %s) det(%s smoothed=%d k%d s%.2f) filter=%s cut %d pairs %s" % (
            t, W, H, n, kinds, prep, use_maps, eq, clip, tiles, prm, smoothed, ksize, sigma, dth, cut, pairs)
        # ---- the oracle's frames and lines -------------------------------------------------------------------------------------
        mx = my = None
        if prep and use_maps:
            mx, my = euroc_maps(W, H, W / 752.0)
        imgs = raw
        if prep:
            imgs = np.stack([oracle_clahe(oracle_remap(f, mx, my) if use_maps else f, clip, tiles) if eq else
                             (oracle_remap(f, mx, my) if use_maps else f) for f in raw])
            done["prepared"] += 1
        found = [o.edlines(imgs[i], smoothed=smoothed, ksize=ksize, sigma=sigma, cap_lines=8192, **prm) for i in range(n)]
        too_many = max(len(l) for l in found) > 1024
        want = [o.line_filter(l, dth) if dth is not None and len(l) else l for l in found]
        done["filtered"] += n if dth is not None else 0
        if rng.random() < 0.2:
            fe._check(fe.lib.vpl_fe_set_stream(fe.h, C.c_void_p(streams[int(rng.integers(0, len(streams)))])), "vpl_fe_set_stream")
        # ---- the device chain ----------------------------------------------------------------------------------------------
        matched = bool(pairs) and min(W, H) >= 64
        try:
            if prep:
                fe.set_maps(mx, my)
                fe.pre_upload(raw)
                fe.pre_run(eq, clip, tiles)
            else:
                fe.upload(raw)
            fe.detect(p, smoothed)
            if rng.random() < 0.2:                       # the stream changed between two enqueued stages
                fe._check(fe.lib.vpl_fe_set_stream(fe.h, C.c_void_p(streams[int(rng.integers(0, len(streams)))])), "vpl_fe_set_stream")
            if dth is not None:
                fe.line_filter_detected(dth)
            if matched:
                fe.match_from_detected(pairs, cut)
                fe.match_run(mp)
            fe.synchronize()
            lines = fe.download()
            if matched:
                r2c, ok = fe.match_download()
                nr, nc = fe.match_counts()
        except RuntimeError as e:
            if too_many and "lines found" in str(e):     # more lines than the table holds: refused, and the context goes on
                print(tag, "refused as it must be:", e)
                done["refused"] = done.get("refused", 0) + 1
                continue
            print(tag, "REFUSED:", e)
            return 1
        if too_many:
            print(tag, "DIFFERS: a frame with %d lines was accepted" % max(len(l) for l in found)); return 1
        for i in range(n):
            lo = want[i]
            if len(lines[i]) != len(lo) or (len(lo) and (np.abs(lines[i][:, :4] - lo[:, :4]).max() > 1e-3 or np.abs(lines[i][:, 4:7] - lo[:, 4:7]).max() > 1e-9)):
                print(tag, "frame", i, "DIFFERS: lines (%d / %d)" % (len(lines[i]), len(lo))); return 1
        if matched:
            for q, (a, b) in enumerate(pairs):
                la, lb = lines[a][:cut], lines[b][:cut]  # the matcher's input: the lines as the device holds them
                if (nr[q], nc[q]) != (len(la), len(lb)):
                    print(tag, "pair", q, "DIFFERS: counts", (nr[q], nc[q]), (len(la), len(lb))); return 1
                if len(la) == 0 or len(lb) == 0:
                    if ok[q]:
                        print(tag, "pair", q, "DIFFERS: matched with an empty list"); return 1
                    continue
                oko, ro, _ = o.line_match(imgs[a], imgs[b], la, lb, o.lm_default_param(bool(mp.illumination_adapt), bool(mp.topological_filter)))
                if bool(ok[q]) != bool(oko) or (oko and not np.array_equal(r2c[q][:len(ro)], ro)):
                    print(tag, "pair", q, (a, b), "DIFFERS: matches, ok %s/%s, %d / %d matched" % (ok[q], oko, int((r2c[q] >= 0).sum()), int((np.asarray(ro) >= 0).sum())))
                    return 1
                done["pairs"] += 1; done["matched"] += int((np.asarray(ro) >= 0).sum()) if oko else 0
        print(tag, "lines", [len(l) for l in want], "ok")
    for fe in ctxs.values():                           # (VPL_DEBUG_GUARDS=1: close() checks the pads behind the device arrays)
        fe.close()
    print("fuzz_frontend3: %d trials of the device-resident chain on long-lived contexts %s: lines and matches identical to the oracle" % (trials, done))
    return 0


if __name__ == "__main__":
    sys.exit(main())
