"""Line front-end (EDLines).
CPU: the oracle's gradient stage against an independent NumPy restatement (integer-exact), structural
properties of the routing and of the extracted lines on the two EuRoC MH_04 fixtures.
GPU: every stage of the HIP extractor against the oracle: gradient / direction / anchors / edge chains
bit-exact, lines as a set."""
import os

import numpy as np
import pytest

import oracle_api as o
import vplines_slam_amd as v

HERE = os.path.dirname(os.path.abspath(__file__))
IMGS = [np.load(os.path.join(HERE, "golden", "mh04_%d.npy" % i)) for i in (1, 2)]


def numpy_gradient(img, grad_th=30):
    """independent restatement of edline_detector.cpp:125-136 with array ops"""
    a = np.pad(img.astype(np.int32), 1, mode="reflect")   # numpy 'reflect' == BORDER_REFLECT_101
    gx = (a[:-2, 2:] + 2 * a[1:-1, 2:] + a[2:, 2:]) - (a[:-2, :-2] + 2 * a[1:-1, :-2] + a[2:, :-2])
    gy = (a[2:, :-2] + 2 * a[2:, 1:-1] + a[2:, 2:]) - (a[:-2, :-2] + 2 * a[:-2, 1:-1] + a[:-2, 2:])
    s = np.abs(gx) + np.abs(gy)
    t = np.where(s > grad_th + 1, s, 0)
    g = np.rint(t * 0.25).astype(np.int16)                 # np.rint rounds half to even, like cv::saturate_cast
    d = np.where(np.abs(gx) < np.abs(gy), 255, 0).astype(np.uint8)
    return gx.astype(np.int16), gy.astype(np.int16), g, d


def test_oracle_gradient_stage_is_integer_exact():
    for img in IMGS:
        _, st = o.edlines(img, want_stages=True)
        gx, gy, g, d = numpy_gradient(img)
        assert np.array_equal(st["dx"], gx) and np.array_equal(st["dy"], gy)
        assert np.array_equal(st["g"], g) and np.array_equal(st["dir"], d)


def test_oracle_routing_and_lines_properties():
    img = IMGS[0]
    lines, st = o.edlines(img, want_stages=True)
    H, W = img.shape
    # anchors sit on the odd lattice, in the reference's scan order (w outer, h inner)
    a = st["anchors"].astype(np.int64)
    assert np.all(a[:, 0] % 2 == 1) and np.all(a[:, 1] % 2 == 1)
    key = a[:, 0] * 10000 + a[:, 1]
    assert np.all(np.diff(key) > 0)
    # chains: 8-connected, every pixel has a positive thresholded gradient, no pixel is used twice inside a chain
    sid = st["sid"].astype(np.int64)
    cx, cy = st["chain_x"].astype(np.int64), st["chain_y"].astype(np.int64)
    assert np.all(st["g"][cy, cx] > 0)
    for e in range(len(sid) - 1):
        xs, ys = cx[sid[e]:sid[e + 1]], cy[sid[e]:sid[e + 1]]
        assert len(xs) >= 35
        assert np.all(np.maximum(np.abs(np.diff(xs)), np.abs(np.diff(ys))) == 1)
        assert len(set(zip(xs.tolist(), ys.tolist()))) == len(xs)
    # lines: unit normal, endpoints on the line, length consistent, longer than ~minLineLen
    assert len(lines) > 50
    n = np.hypot(lines[:, 4], lines[:, 5])
    assert np.abs(n - 1).max() < 1e-12
    for k in (0, 2):
        res = lines[:, 4] * lines[:, k] + lines[:, 5] * lines[:, k + 1] + lines[:, 6]
        assert np.abs(res).max() < 1e-3
    assert np.abs(np.hypot(lines[:, 2] - lines[:, 0], lines[:, 3] - lines[:, 1]) - lines[:, 9]).max() < 1e-3
    assert lines[:, 9].min() > 30
    # idempotence / determinism
    lines2 = o.edlines(img)
    assert np.array_equal(lines, lines2)


def test_oracle_edge_cases():
    flat = np.full((64, 96), 17, np.uint8)
    lines, st = o.edlines(flat, want_stages=True)
    assert len(lines) == 0 and len(st["anchors"]) == 0 and np.all(st["g"] == 0)
    step = np.zeros((64, 96), np.uint8)
    step[:, 47] = 50                                     # a vertical edge whose gradient peaks in column 47
    step[:, 48:] = 200                                   # (a perfect step gives a 2-px plateau and no anchor)
    lines = o.edlines(step, min_len=20)
    assert len(lines) >= 1
    assert np.all(np.abs(np.abs(lines[:, 4]) - 1) < 1e-6)   # vertical lines: normal along x
    assert np.all(np.abs(lines[:, 0] - 47.0) < 1.0)
    # NFA known answers: n == k and k == 0 closed forms of edline_detector.h:253-254
    lib = o.load()
    lib.orc_nfa.restype = __import__("ctypes").c_double
    lib.orc_nfa.argtypes = [__import__("ctypes").c_int, __import__("ctypes").c_int, __import__("ctypes").c_double, __import__("ctypes").c_double]
    logNT = 2 * (np.log10(752) + np.log10(480))
    assert abs(lib.orc_nfa(40, 40, 0.125, logNT) - (-logNT - 40 * np.log10(0.125))) < 1e-12
    assert abs(lib.orc_nfa(40, 0, 0.125, logNT) + logNT) < 1e-12
    # against the exact binomial tail (scipy) within the documented 10 % truncation of the series
    from scipy.stats import binom
    for (n, k) in [(60, 30), (100, 40), (35, 20)]:
        exact = -np.log10(binom.sf(k - 1, n, 0.125)) - logNT
        assert abs(lib.orc_nfa(n, k, 0.125, logNT) - exact) < 0.1 * abs(exact) + 0.05


def canon(lines):
    """canonical order for set comparison: sort by rounded endpoints"""
    key = np.lexsort((np.round(lines[:, 3], 2), np.round(lines[:, 2], 2), np.round(lines[:, 1], 2), np.round(lines[:, 0], 2)))
    return lines[key]


@pytest.mark.gpu
def test_gpu_edlines_stages_and_lines_match_oracle():
    imgs = np.stack(IMGS + [IMGS[0][::-1].copy(), np.ascontiguousarray(IMGS[1][:, ::-1])])   # + flipped variants
    fe = v.frontend.FrontendContext(device=0, max_images=len(imgs), width=752, height=480, max_lines=1024)
    out = fe.detect_batch(imgs)
    for i in range(len(imgs)):
        lo, st = o.edlines(imgs[i], want_stages=True)
        sg = fe.debug_stage(i)
        for k in ("dx", "dy", "g", "dir"):
            assert np.array_equal(sg[k], st[k]), k
        assert np.array_equal(sg["anchors"], st["anchors"])
        assert np.array_equal(sg["sid"], st["sid"])
        assert np.array_equal(sg["chain_x"], st["chain_x"]) and np.array_equal(sg["chain_y"], st["chain_y"])
        lg = out[i]
        assert len(lg) == len(lo), (len(lg), len(lo))
        a, b = canon(lg), canon(lo)
        assert np.abs(a[:, :4] - b[:, :4]).max() < 1e-3       # endpoints (float32 in the reference's Line)
        assert np.abs(a[:, 4:7] - b[:, 4:7]).max() < 1e-9     # line equation (double)
        assert np.abs(a[:, 9] - b[:, 9]).max() < 1e-3
        # and in the SAME ORDER, the serial order of the reference's loop over edge chains, which the reference-held
        # pictures pin up to the interleaving of its parallel_for (test_edline_reference_picture.py, list_index)
        assert np.abs(lg[:, :4] - lo[:, :4]).max() < 1e-3
    fe.close()


@pytest.mark.gpu
def test_gpu_edlines_parameters_and_degenerate_frames():
    p = v.frontend.default_param()
    p.minLineLen, p.lineFitErrThreshold, p.gradientThreshold = 20, 1.4, 50.0
    rng = np.random.default_rng(3)
    flat = np.full((480, 752), 9, np.uint8)
    noise = rng.integers(0, 255, (480, 752), dtype=np.uint8)
    imgs = np.stack([IMGS[0], flat, noise])
    fe = v.frontend.FrontendContext(device=0, max_images=3, width=752, height=480, max_lines=2048)
    out = fe.detect_batch(imgs, p)
    assert len(out[1]) == 0
    for i in (0, 2):
        lo = o.edlines(imgs[i], grad_th=50, min_len=20, fit_err=1.4)
        assert len(out[i]) == len(lo)
        if len(lo):
            assert np.abs(canon(out[i])[:, 4:7] - canon(lo)[:, 4:7]).max() < 1e-9
    fe.close()


@pytest.mark.gpu
def test_gpu_edlines_other_frame_sizes():
    """Frame sizes around the routing strip's limits (round 4): 1024 x 768 (edge bitmap 98 KB: the strip of routing bytes is
    lower than the frame and is reloaded vertically as well), 752 x 480 (full-height strip), 160 x 120 and 70 x 50 (strip
    higher / wider than the frame).  Every stage and the line set against the oracle, with and without the pre-blur."""
    base = IMGS[0]
    for (W, H) in ((1024, 768), (160, 120), (70, 50)):
        ys = (np.arange(H) * base.shape[0] / H).astype(int)
        xs = (np.arange(W) * base.shape[1] / W).astype(int)
        a = base[ys][:, xs]                                   # nearest-neighbour resize of the real frame
        b = np.ascontiguousarray(a[::-1, ::-1])
        imgs = np.stack([a, b])
        fe = v.frontend.FrontendContext(device=0, max_images=2, width=W, height=H, max_lines=2048)
        p = v.frontend.default_param()
        p.minLineLen = 15
        for smoothed in (True, False):
            out = fe.detect_batch(imgs, p, smoothed=smoothed)
            for i in range(2):
                lo, st = o.edlines(imgs[i], min_len=15, want_stages=True, smoothed=smoothed)
                sg = fe.debug_stage(i)
                for k in ("dx", "dy", "g", "dir", "anchors", "sid", "chain_x", "chain_y"):
                    assert np.array_equal(sg[k], st[k]), (W, H, smoothed, i, k)
                assert len(out[i]) == len(lo), (W, H, smoothed, i)
                if len(lo):
                    assert np.abs(canon(out[i])[:, 4:7] - canon(lo)[:, 4:7]).max() < 1e-9
        if (W, H) == (1024, 768):
            assert len(lo) > 50 and fe.route_stats(0)["tile_loads"] > 0      # strip loads happened
        fe.close()


def test_more_edges_than_the_reference_buffers_hold():
    """Noise with a small minLineLen produces more edge chains than maxNumOfEdge = W H / 100 (edline_detector.cpp:92-93, :117): the
    reference writes past its buffers and reports the error afterwards (:655-659).  Here (oracle and device alike) an edge that
    does not fit is dropped; the oracle used to hand more chain starts to the caller than the caller's buffer held (found by
    tools/fuzz_frontend.py)."""
    W, H = 333, 200
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (H, W)).astype(np.float64)
    for ax in (0, 1):                                  # 3-tap box blur, borders repeated: noise turned into small blobs
        pad = np.concatenate([np.take(a, [0], ax), a, np.take(a, [-1], ax)], ax)
        a = (np.take(pad, range(0, pad.shape[ax] - 2), ax) + np.take(pad, range(1, pad.shape[ax] - 1), ax)
             + np.take(pad, range(2, pad.shape[ax]), ax)) / 3.0
    img = np.ascontiguousarray((255 * (a - a.min()) / (a.max() - a.min())).astype(np.uint8))
    lines, st = o.edlines(img, grad_th=16, anchor_th=2, scan=2, min_len=8, want_stages=True)
    cap_edges = (W * H // 5) // 20
    assert len(st["sid"]) - 1 == cap_edges            # the buffer is full, not overrun (more chains were found)
    assert st["sid"][-1] == len(st["chain_x"]) <= 2 * (W * H // 5)


@pytest.mark.gpu
def test_gpu_frontend_random_sizes_contents_parameters():
    """a short run of tools/fuzz_frontend.py (its header lists what is drawn): every stage, the line lists in order, LineFilter
    and the matches identical to the oracle.  Round 4's sweeps: 260 trials over three seeds, no difference
    (gpurun_out/r4_fuzzfe*.log)."""
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(HERE), "tools", "fuzz_frontend.py")
    r = subprocess.run([sys.executable, tool, "24", "17"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "24 trials" in r.stdout


@pytest.mark.gpu
def test_gpu_more_lines_than_the_table_holds_is_refused():
    """the reference's line list is a std::vector; the device's table has max_lines_per_image rows per frame.  A frame with more
    lines is refused at the download (which lines were kept is not defined), for the detector and for a match fed from it;
    the context goes on working."""
    fe = v.frontend.FrontendContext(device=0, max_images=2, width=752, height=480, max_lines=64)
    fe.match_reserve(1, 4096)
    with pytest.raises(RuntimeError, match="lines found"):
        fe.detect_batch(np.stack(IMGS))                    # ~230 lines per frame
    fe.upload(np.stack(IMGS))
    fe.detect()
    fe.match_from_detected([(0, 1)])
    fe.match_run()
    fe.synchronize()
    with pytest.raises(RuntimeError, match="lines found"):
        fe.match_download()
    p = v.frontend.default_param()
    p.minLineLen = 120                                      # few long lines: fits
    out = fe.detect_batch(np.stack(IMGS), p)
    for i in range(2):
        lo = o.edlines(IMGS[i], min_len=120)
        assert 0 < len(lo) <= 64 and len(out[i]) == len(lo)
    fe.close()


SINGULAR_FIT = dict(grad_th=77, anchor_th=12, scan=1, min_len=12, fit_err=1.5655807874942627)


def test_a_singular_initial_fit_ends_the_chain():
    """tests/golden/edlines_singular_fit_33x400.npy (a frame of tools/fuzz_frontend.py, blurred noise): the last edge chain ends
    in 12 pixels with one and the same abscissa, the 2 x 2 normal equations of the initial fit are singular, the fit error is
    NaN.  The reference's loop (edline_detector.cpp:989-997) then leaves with offS two pixels on and extends a "line" from
    beyond the end of the chain: a read past the edge's pixels (seen as a wild index under the address sanitizer).  Oracle and
    device end the chain there."""
    img = np.load(os.path.join(HERE, "golden", "edlines_singular_fit_33x400.npy"))
    lines, st = o.edlines(img, smoothed=False, want_stages=True, **SINGULAR_FIT)
    assert len(st["sid"]) - 1 == 99 and st["sid"][-1] == 2074 and len(lines) == 0


@pytest.mark.gpu
def test_gpu_a_singular_initial_fit_ends_the_chain():
    img = np.load(os.path.join(HERE, "golden", "edlines_singular_fit_33x400.npy"))
    fe = v.frontend.FrontendContext(device=0, max_images=1, width=33, height=400, max_lines=256)
    p = v.frontend.default_param()
    p.gradientThreshold, p.anchorThreshold, p.scanIntervals = SINGULAR_FIT["grad_th"], SINGULAR_FIT["anchor_th"], SINGULAR_FIT["scan"]
    p.minLineLen, p.lineFitErrThreshold = SINGULAR_FIT["min_len"], SINGULAR_FIT["fit_err"]
    for _ in range(3):      # (whatever an earlier frame left behind the chain's end must not matter)
        out = fe.detect_batch(img[None], p, smoothed=False)
        assert len(out[0]) == 0
    fe.close()


@pytest.mark.gpu
def test_gpu_half_height_routing_strip_gives_the_same_chains():
    """Contexts that hold more frames than the device has CUs take a routing strip of at most half the LDS (256 rows for
    752 x 480), so that two frames' walkers share a CU (14.9 k instead of 11.5 k frames/s with 512 frames in flight,
    tools/fe_scale.py).  The strip then moves vertically too; forced here on a small context: same chains, same lines."""
    imgs = np.stack(IMGS + [np.ascontiguousarray(IMGS[0][::-1])])
    old = os.environ.get("VPL_FE_ROUTE_HS")
    os.environ["VPL_FE_ROUTE_HS"] = "256"
    try:
        fe = v.frontend.FrontendContext(device=0, max_images=len(imgs), width=752, height=480, max_lines=1024)
    finally:
        if old is None:
            del os.environ["VPL_FE_ROUTE_HS"]
        else:
            os.environ["VPL_FE_ROUTE_HS"] = old
    out = fe.detect_batch(imgs)
    for i in range(len(imgs)):
        lo, st = o.edlines(imgs[i], want_stages=True)
        sg = fe.debug_stage(i)
        for k in ("anchors", "sid", "chain_x", "chain_y"):
            assert np.array_equal(sg[k], st[k]), (i, k)
        assert len(out[i]) == len(lo) and np.abs(out[i][:, :4] - lo[:, :4]).max() < 1e-3
    assert fe.route_stats(0)["tile_loads"] > 143       # more strip loads than the full-height strip needs
    fe.close()
