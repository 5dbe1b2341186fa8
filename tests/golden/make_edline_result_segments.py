"""Recover the line segments the REFERENCE drew into line_matching/data/edline_result.png.

That picture is the only expected OUTPUT the reference tree holds for the line front-end: it is what
line_matching/src/test_edline_detector.cpp:13-74 shows (and, by the commented imwrite at :72, once saved) for
    EDLineParam{5, 1.0, 30, 5, 2, 25, 1.8};  EDline(imread("data/mh04/imgs/1.png", 0), lines, /*smoothed=*/false)
Every line i is drawn as cv::line(trunc(endpoints), Scalar(rand, rand, rand), thickness 2, CV_AA) in a colour of
its own, followed by two 10-pixel arrows from its centre (normal (255,0,0) BGR, direction (0,0,255) BGR, 1 px, CV_AA).
Later lines overdraw earlier lines and arrows.

What is recovered (data, not source): for every colour that forms a thin straight stripe of fully covered pixels --
the segment's two end points, the number of stripe pixels, the colour, and list_index: the position of that line in
the reference's output list (the colours are rand() % 256 triples after srand(time(0)); the one seed that explains them
was found with find_srand_seed.c, and it explains all 258); plus a label image of the picture
(-1 = the grey frame's own pixel, -2 = painted with a blend or an arrow, >= 0 = fully covered by stripe i).  Run in
the build container only (reads /root/reference); the result is committed as tests/golden/edline_result_segments.npz.

Two properties of OpenCV's drawing (imgproc/drawing.cpp, ThickLine) enter the recovery:
 * a line of thickness 2 is the polygon of half-width 1 around the segment plus, at each end, EllipseEx with axes (1, 1):
   ellipse2Poly with delta = 90 degrees -- a diamond whose vertex lies 1 px BEYOND the end point.  The extreme fully covered
   pixels of a stripe are therefore 1 px outside the drawn end points (measured on the stripes: 295 of 388 ends of
   unoccluded lines, the rest 0 or 2); the end points written here are the extremes pulled in by 1 px along the axis.
 * the anti-aliased rim of an axis-parallel or 45-degree line has the same coverage along its whole length; over a
   flat background the blend is one colour and forms a "stripe" 1 px beside the real one.  Stripes whose two ends lie
   within 1.6 px of a stripe with more pixels are such rims and are dropped.

    python tests/golden/make_edline_result_segments.py
"""
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/line_matching/data"

MIN_PIXELS = 15      # a 25-pixel line of width 2 has >= ~40 fully covered pixels; partly overdrawn ones fewer
MAX_MINOR_SD = 1.05  # a 2-pixel stripe: 0.5 (axis parallel) .. 0.8
MIN_MAJOR_SD = 3.5   # ~12 px long
CAP = 1.0            # the end cap's vertex beyond the end point
RIM = 1.6


def recover(png, frame):
    im = np.array(Image.open(png))[..., :3].astype(np.int64)
    g = frame.astype(np.int64)
    assert im.shape[:2] == g.shape
    key = (im[..., 0] << 16) | (im[..., 1] << 8) | im[..., 2]          # R G B as PIL gives them
    gk = (g << 16) | (g << 8) | g
    painted = key != gk
    vals, inv, cnt = np.unique(key[painted], return_inverse=True, return_counts=True)
    ys, xs = np.nonzero(painted)
    segs = []
    for ci in np.nonzero(cnt >= MIN_PIXELS)[0]:
        sel = inv == ci
        pts = np.stack([xs[sel], ys[sel]], 1).astype(np.float64)
        m = pts.mean(0)
        w, V = np.linalg.eigh(np.cov((pts - m).T))
        if np.sqrt(max(w[0], 0.0)) > MAX_MINOR_SD or np.sqrt(w[1]) < MIN_MAJOR_SD:
            continue
        d = V[:, 1]
        t = (pts - m) @ d
        # the stripe must be contiguous along its axis apart from what later drawing covers: keep the run structure
        # out of the test (arrows cut stripes in two); what is required is thinness
        off = np.abs((pts - m) @ V[:, 0])
        if off.max() > 2.0:
            continue
        p0 = m + (t.min() + CAP) * d
        p1 = m + (t.max() - CAP) * d
        k = int(vals[ci])
        segs.append((p0[0], p0[1], p1[0], p1[1], len(pts), (k >> 16) & 255, (k >> 8) & 255, k & 255))
    segs = np.array(sorted(segs), np.float64)
    # anti-aliased rims of a longer stripe
    keep = np.ones(len(segs), bool)
    for i, a in enumerate(segs):
        for j, b in enumerate(segs):
            if i == j or b[4] <= a[4]:
                continue
            if max(pt_seg(a[0:2], b), pt_seg(a[2:4], b)) <= RIM:
                keep[i] = False
    segs = segs[keep]
    labels = np.full(painted.shape, -1, np.int16)
    labels[painted] = -2
    for i, sg in enumerate(segs):
        kk = (int(sg[5]) << 16) | (int(sg[6]) << 8) | int(sg[7])
        labels[key == kk] = i
    return segs, labels


def pt_seg(p, sg):
    a, e = sg[0:2], sg[2:4]
    v = e - a
    t = np.clip(((p - a) @ v) / max(v @ v, 1e-9), 0.0, 1.0)
    return float(np.linalg.norm(p - (a + t * v)))


SRAND_SEED = 1612579976      # the demo's srand(time(0)), found by find_srand_seed.c from the stripes' colours


def glibc_rand(seed, n):
    """the first n values of glibc rand() after srand(seed) (TYPE_3 additive feedback generator)"""
    seed = seed or 1
    r = [0] * (344 + n)
    r[0] = seed
    for i in range(1, 31):
        r[i] = (16807 * r[i - 1]) % 2147483647
    for i in range(31, 34):
        r[i] = r[i - 31]
    for i in range(34, 344 + n):
        r[i] = (r[i - 31] + r[i - 3]) & 0xFFFFFFFF
    return [x >> 1 for x in r[344:]]


def list_positions(colour_rgb):
    """stripe k -> i, the position of its line in the reference's output list: the demo colours line i with the i-th
    triple of rand() % 256 (test_edline_detector.cpp:52-59; Scalar(r, g, b) is B, G, R)"""
    rnd = glibc_rand(SRAND_SEED, 3 * len(colour_rgb))
    where = {(rnd[3 * i + 2] % 256, rnd[3 * i + 1] % 256, rnd[3 * i] % 256): i for i in range(len(colour_rgb))}
    assert len(where) == len(colour_rgb)
    pos = np.array([where.get(tuple(int(v) for v in c), -1) for c in colour_rgb], np.int32)
    assert (pos >= 0).all() and len(set(pos.tolist())) == len(pos), "the seed does not explain every stripe"
    return pos


def main():
    frame = np.load(os.path.join(HERE, "mh04_1.npy"))
    segs, labels = recover(os.path.join(REF, "edline_result.png"), frame)
    out = os.path.join(HERE, "edline_result_segments.npz")
    colour = segs[:, 5:8].astype(np.uint8)
    np.savez_compressed(out, segments=segs[:, :4], pixels=segs[:, 4].astype(np.int32),
                        colour_rgb=colour, labels=labels, list_index=list_positions(colour))
    print("recovered %d segments, %d painted pixels -> %s" % (len(segs), int((labels != -1).sum()), out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
