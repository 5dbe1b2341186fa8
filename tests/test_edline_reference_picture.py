"""The oracle's EDLines against the ONE expected output the reference tree holds for the line front-end.

line_matching/data/edline_result.png is what the reference's own demo (line_matching/src/test_edline_detector.cpp:13-74)
rendered for  EDLineParam{5, 1.0, 30, 5, 2, 25, 1.8},  EDline(imread("mh04/imgs/1.png", 0), lines, smoothed = false).
tests/golden/make_edline_result_segments.py recovered the drawn segments and a label image from it (committed as
tests/golden/edline_result_segments.npz; the picture itself does not travel).  The oracle runs the same call on the
same frame (tests/golden/mh04_1.npy) and must
  * find the same number of lines,
  * have every pixel of every line painted in the picture,
  * explain every painted pixel of the picture by one of its lines or that line's two 10-px arrows,
  * have both end points of (nearly) every line within 1.5 px of a recovered stripe's (242 of 258) -- the others are within
    2.5 px (the end cap of cv::line reaches 0 - 2 px beyond the end point, 1 is assumed) or belong to stripes that later
    drawing covers at one end (then the missing stretch is painted in another colour).
  * list its lines in the reference's order up to the interleaving its parallel loop produces (list_index).
The same checks FAIL without the Gaussian pre-blur and with the other published rounding of OpenCV's fixed-point kernel
(the last test), i.e. the picture discriminates at the level of the blur's 8.8 taps.
"""
import os

import numpy as np
import pytest

import oracle_api as o

HERE = os.path.dirname(os.path.abspath(__file__))
PARAM = dict(grad_th=30, anchor_th=5, scan=2, min_len=25, fit_err=1.8)   # test_edline_detector.cpp:15


@pytest.fixture(scope="module")
def fx():
    z = np.load(os.path.join(HERE, "golden", "edline_result_segments.npz"))
    frame = np.load(os.path.join(HERE, "golden", "mh04_1.npy"))
    return dict(seg=z["segments"], labels=z["labels"], frame=frame, list_index=z["list_index"])


def drawn(lines):
    """what the demo hands to cv::line / cv::arrowedLine: cv::Point(float, float) truncates"""
    return np.trunc(lines[:, :4])


def end_distance(P, seg):
    """[n_lines, n_segments]: max over the two ends of the end-point distance, the better of the two orientations"""
    D = np.zeros((len(P), len(seg)))
    for i, p in enumerate(P):
        d1 = np.maximum(np.hypot(seg[:, 0] - p[0], seg[:, 1] - p[1]), np.hypot(seg[:, 2] - p[2], seg[:, 3] - p[3]))
        d2 = np.maximum(np.hypot(seg[:, 0] - p[2], seg[:, 1] - p[3]), np.hypot(seg[:, 2] - p[0], seg[:, 3] - p[1]))
        D[i] = np.minimum(d1, d2)
    return D


def raster(p):
    x0, y0, x1, y1 = p
    n = int(max(abs(x1 - x0), abs(y1 - y0))) + 1
    return (np.rint(np.linspace(x0, x1, n)).astype(int), np.rint(np.linspace(y0, y1, n)).astype(int))


def dist_to_segments(pts, segs):
    d = np.full(len(pts), 1e9)
    for a, e in segs:
        v = e - a
        t = np.clip(((pts - a) @ v) / max(v @ v, 1e-9), 0, 1)
        d = np.minimum(d, np.linalg.norm(pts - (a + t[:, None] * v), axis=1))
    return d


def test_same_number_of_lines(fx):
    lines = o.edlines(fx["frame"], smoothed=False, ksize=5, sigma=1.0, **PARAM)
    assert len(lines) == len(fx["seg"]) == 258


def test_every_line_pixel_is_painted_in_the_reference_picture(fx):
    lines = o.edlines(fx["frame"], smoothed=False, **PARAM)
    H, W = fx["labels"].shape
    total = hit = 0
    for p in drawn(lines):
        xs, ys = raster(p)
        ok = (xs >= 0) & (xs < W) & (ys >= 0) & (ys < H)
        v = fx["labels"][ys[ok], xs[ok]] != -1
        total += len(v)
        hit += int(v.sum())
    assert total > 13000
    assert hit == total, (hit, total)


def test_every_painted_pixel_is_explained_by_a_line_or_its_arrows(fx):
    lines = o.edlines(fx["frame"], smoothed=False, **PARAM)
    ys, xs = np.nonzero(fx["labels"] != -1)
    pts = np.stack([xs, ys], 1).astype(np.float64)
    segs = []
    for l in lines:
        p = np.trunc(l[:4])
        c = np.trunc(l[7:9])                     # cv::Point(mid_x, mid_y), test_edline_detector.cpp:69-70
        n = l[4:6]
        segs.append((p[:2], p[2:]))
        segs.append((c, np.trunc(l[7:9] + 10 * n)))
        segs.append((c, np.trunc(l[7:9] + 10 * np.array([-n[1], n[0]]))))
    d = dist_to_segments(pts, segs)
    # thickness 2 + anti-aliased rim: 2 px; arrow heads (tipLength 0.1 -> 1 px) and caps: a little more
    assert d.max() <= 3.5, d.max()
    assert (d <= 2.5).mean() >= 0.995, (d <= 2.5).mean()


def test_end_points_match_the_recovered_stripes(fx):
    lines = o.edlines(fx["frame"], smoothed=False, **PARAM)
    P, seg, labels = drawn(lines), fx["seg"], fx["labels"]
    D = end_distance(P, seg)
    both = D.min(1) <= 1.5
    assert both.sum() >= 240, both.sum()                    # measured: 242 of 258
    assert (D.min(0) <= 1.5).sum() == both.sum()            # one to one
    # the others: the stripe is the line with one end under later paint
    H, W = labels.shape
    for i in np.nonzero(~both)[0]:
        p = P[i]
        a, e = p[:2], p[2:]
        u = (e - a) / np.linalg.norm(e - a)
        nrm = np.array([-u[1], u[0]])
        # candidate stripes: both ends within 1.5 px of the line's axis and inside its extent
        off = np.maximum(np.abs((seg[:, :2] - a) @ nrm), np.abs((seg[:, 2:] - a) @ nrm))
        t0, t1 = (seg[:, :2] - a) @ u, (seg[:, 2:] - a) @ u
        lo, hi = np.minimum(t0, t1), np.maximum(t0, t1)
        length = np.linalg.norm(e - a)
        cand = np.nonzero((off <= 1.5) & (lo >= -1.5) & (hi <= length + 1.5))[0]
        assert len(cand) >= 1, (i, p)
        j = cand[np.argmax(hi[cand] - lo[cand])]
        assert hi[j] - lo[j] >= 0.5 * length, (i, p, seg[j])
        # what the stripe does not cover (beyond the +-1 px the cap correction is uncertain by, and the 1.5 px bar) is
        # painted, and not in the stripe's own colour
        xs, ys = raster(p)
        t = (np.stack([xs, ys], 1) - a) @ u
        out = ((t < lo[j] - 2.5) | (t > hi[j] + 2.5)) & (xs >= 0) & (xs < W) & (ys >= 0) & (ys < H)
        lab = labels[ys[out], xs[out]]
        assert (lab != -1).all(), (i, p)
        assert (lab != j).all(), (i, p)


def fewest_increasing_runs(seq):
    """= length of the longest strictly decreasing subsequence"""
    import bisect
    tails = []
    for v in seq:
        k = bisect.bisect_left(tails, -int(v))
        if k == len(tails):
            tails.append(-int(v))
        else:
            tails[k] = -int(v)
    return len(tails)


def test_list_order_is_an_interleaving_of_the_oracles_order(fx):
    """The demo colours line i of the detector's output with the i-th rand() triple after srand(time(0)); the seed is
    recoverable from the colours (tests/golden/find_srand_seed.c: one seed fits, and it explains all 258 stripes), so
    the picture also holds every line's POSITION in the reference's list (list_index).  The reference emits lines from
    cv::parallel_for_ stripes of the edge list under a lock (edline_detector.cpp:1081-1083, 1165-1167, 1195): its list is
    an interleaving of a few in-order runs of the serial order.  The oracle's list is the serial order: read in the
    reference's order it falls into 3 increasing runs (a random order of 254 needs about 30)."""
    lines = o.edlines(fx["frame"], smoothed=False, **PARAM)
    D = end_distance(drawn(lines), fx["seg"])
    stripe = D.argmin(1)
    good = D.min(1) <= 1.5
    assert good.sum() >= 240 and len(set(stripe[good])) == good.sum()
    assert sorted(fx["list_index"]) == list(range(len(lines)))
    position = fx["list_index"][stripe[good]]                 # oracle line (in oracle order) -> place in the reference's list
    oracle_index = np.nonzero(good)[0]
    in_reference_order = oracle_index[np.argsort(position)]
    assert fewest_increasing_runs(in_reference_order) <= 3
    rng = np.random.default_rng(1)
    assert min(fewest_increasing_runs(rng.permutation(in_reference_order)) for _ in range(20)) >= 18
    # and the first line of either list is the same one
    assert in_reference_order[0] == 0 and position.min() == 0


def test_the_picture_discriminates_the_blur(fx):
    """teeth: without the blur, and with the un-normalised 8.8 kernel of OpenCV 3.4.1-3.4.8 (14 63 103 63 14), the same
    checks fail by a wide margin -- the picture pins the blur down to its fixed-point taps"""
    seg = fx["seg"]
    frame = fx["frame"]
    n_ok = {}
    for name, kw in (("normalised", dict(smoothed=False)), ("opencv341", dict(smoothed=False, blur_mode=o.BLUR_OPENCV_341)),
                     ("none", dict(smoothed=True))):
        lines = o.edlines(frame, **PARAM, **kw)
        n_ok[name] = (int((end_distance(drawn(lines), seg).min(1) <= 1.5).sum()), len(lines))
    assert n_ok["normalised"][0] >= 240 and n_ok["normalised"][1] == 258
    assert n_ok["opencv341"][0] <= 200, n_ok                 # measured 163 of 260
    assert n_ok["none"][0] <= 100, n_ok                      # measured 45 of 281
    _, k0 = o.gaussian_blur(frame, 5, 1.0, o.BLUR_NORMALISED, want_kernel=True)
    _, k1 = o.gaussian_blur(frame, 5, 1.0, o.BLUR_OPENCV_341, want_kernel=True)
    assert k0.tolist() == [14, 62, 104, 62, 14] and k1.tolist() == [14, 63, 103, 63, 14]


def test_blur_against_an_independent_numpy_restatement():
    """the fixed-point blur against exact rational arithmetic in NumPy (float64 is exact here: all values < 2^53)"""
    rng = np.random.default_rng(7)
    for (H, W, n, sigma, mode) in ((37, 53, 5, 1.0, 0), (37, 53, 5, 1.0, 1), (16, 9, 7, 1.5, 0), (8, 8, 3, 0.8, 0),
                                   (5, 3, 5, 0.0, 0), (480, 752, 5, 1.0, 0)):
        img = rng.integers(0, 256, (H, W)).astype(np.uint8)
        if H == 37:
            img[5:20, 10:30] = 255    # saturation with the 257/256 kernel
        out, k = o.gaussian_blur(img, n, sigma, mode, want_kernel=True)
        assert len(k) == n and (mode == 1 or k.sum() == 256)

        def refl(i, m):
            i = np.abs(i)
            return np.where(i >= m, 2 * m - 2 - i, i)
        r = n // 2
        xi = refl(np.arange(-r, W + r), W)
        yi = refl(np.arange(-r, H + r), H)
        a = img.astype(np.float64)[:, xi]
        h = np.minimum(sum(k[j] * a[:, j:j + W] for j in range(n)), 65535.0)[yi, :]
        v = sum(k[j] * h[j:j + H, :] for j in range(n))
        want = np.minimum(np.floor((v + 32768.0) / 65536.0), 255).astype(np.uint8)
        assert np.array_equal(out, want)
