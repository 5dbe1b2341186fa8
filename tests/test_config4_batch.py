"""BASELINE config 4 -- the 64-frame 752x480 batch (SURVEY 8d "Frame stream") and the 15 real EuRoC MH_04 frames the
reference ships (line_matching/data/mh04/imgs/1..15.png, committed as pixels in tests/golden/mh04_frames.npz).
CPU: the fixture file, the stream generator, the oracle on all 15 real frames (gradient stage against the NumPy restatement,
structural properties, the 14 real consecutive pairs).
GPU: EDLines of all 64 frames and KLT matching of the 63 consecutive pairs in ONE batch through the C ABI, every frame and
every pair compared with the oracle (the oracle needs ~20 ms per frame); the 14 real pairs bit-exact down to the tracked key
points."""
import os

import numpy as np
import pytest

import oracle_api as o
import vplines_slam_amd as v
from test_edlines import canon, numpy_gradient

HERE = os.path.dirname(os.path.abspath(__file__))
MAXL = 256          # lines per frame handed to the matcher (the tracker's cap in the batch benchmark)


def test_fixture_file_holds_the_15_reference_frames():
    fx = v.workload.mh04_fixtures()
    assert fx.shape == (15, 480, 752) and fx.dtype == np.uint8
    for k in (1, 2):
        assert np.array_equal(fx[k - 1], np.load(os.path.join(HERE, "golden", "mh04_%d.npy" % k)))
    # 15 different frames of one sequence: consecutive frames differ, but not wildly
    for k in range(14):
        d = np.abs(fx[k].astype(int) - fx[k + 1].astype(int)).mean()
        assert 0.5 < d < 40, (k, d)


def test_frame_stream_is_deterministic_and_matchable():
    a = v.workload.frame_stream(20)
    b = v.workload.frame_stream(20)
    assert a.shape == (20, 480, 752) and a.dtype == np.uint8 and np.array_equal(a, b)
    fx = v.workload.mh04_fixtures()
    assert np.array_equal(a[:15], fx)
    # frame 15 is fixture 0 warped by < 4 px + noise sigma 2
    d = np.abs(a[15].astype(int) - fx[0].astype(int))
    assert 1.0 < d.mean() < 30.0
    # an identity homography reproduces the image exactly
    assert np.array_equal(np.rint(v.workload.warp_homography(fx[3], np.eye(3))).astype(np.uint8), fx[3])


def test_oracle_on_all_real_frames():
    fx = v.workload.mh04_fixtures()
    L = []
    for k in range(15):
        lines, st = o.edlines(fx[k], want_stages=True)
        gx, gy, g, d = numpy_gradient(fx[k])
        assert np.array_equal(st["dx"], gx) and np.array_equal(st["dy"], gy)
        assert np.array_equal(st["g"], g) and np.array_equal(st["dir"], d)
        assert 40 < len(lines) < 400, (k, len(lines))
        assert np.abs(np.hypot(lines[:, 4], lines[:, 5]) - 1).max() < 1e-12
        assert lines[:, 9].min() > 0 and np.median(lines[:, 9]) > 35   # no length filter in the reference (edline_detector.cpp:1079)
        L.append(lines[:MAXL])
    # the 14 real consecutive pairs: the matcher finds most of the lines again (two reference lines may share a current line)
    for k in range(14):
        ok, r2c, _ = o.line_match(fx[k], fx[k + 1], L[k], L[k + 1])
        assert ok
        hit = r2c[r2c >= 0]
        assert len(hit) >= 0.3 * min(len(L[k]), len(L[k + 1])), (k, len(hit))
        assert hit.max() < len(L[k + 1])


@pytest.mark.gpu
def test_gpu_config4_batch_64_frames_63_pairs():
    n = 64
    imgs = v.workload.frame_stream(n)
    fe = v.frontend.FrontendContext(device=0, max_images=n, width=752, height=480, max_lines=1024)
    fe.match_reserve(n - 1, 8192)
    det = fe.detect_batch(imgs)
    assert len(det) == n
    ref = [o.edlines(imgs[i]) for i in range(n)]
    for i in range(n):
        lg, lo = det[i], ref[i]
        assert len(lg) == len(lo), (i, len(lg), len(lo))
        assert len(lg) > 30
        a, b = canon(lg), canon(lo)
        assert np.abs(a[:, :4] - b[:, :4]).max() < 1e-3
        assert np.abs(a[:, 4:7] - b[:, 4:7]).max() < 1e-9
        # properties: unit normals, end points on the line, lengths
        assert np.abs(np.hypot(lg[:, 4], lg[:, 5]) - 1).max() < 1e-12
        assert lg[:, 9].min() > 0
    # the edge chains of the real frames, bit for bit
    for i in (0, 7, 14, 15, 40, 63):
        _, st = o.edlines(imgs[i], want_stages=True)
        sg = fe.debug_stage(i)
        for k in ("dx", "dy", "g", "dir", "anchors", "sid", "chain_x", "chain_y"):
            assert np.array_equal(sg[k], st[k]), (i, k)
    # the 63 consecutive pairs in one batch; the device's own detections feed the matcher, as in the tracker
    pairs = [(i, i + 1) for i in range(n - 1)]
    lref = [det[a][:MAXL] for a, _ in pairs]
    lcur = [det[b][:MAXL] for _, b in pairs]
    r2c, ok = fe.match_batch(imgs, pairs, lref, lcur)
    n_matched = []
    for i, (a, b) in enumerate(pairs):
        oko, ro, so = o.line_match(imgs[a], imgs[b], lref[i], lcur[i])
        assert bool(ok[i]) == bool(oko), i
        if not oko:
            continue
        assert np.array_equal(r2c[i], ro), i
        hit = ro[ro >= 0]
        n_matched.append(len(hit))
        if i < 14 or i in (15, 31, 62):      # key points bit-exact: all real pairs + some synthetic ones
            sg = fe.match_debug_kps(i)
            assert np.array_equal(sg["kps_ref"], so["kps_ref"])
            assert np.array_equal(sg["status"], so["status"])
            live = so["status"] > 0
            assert np.array_equal(sg["kps_cur"][live], so["kps_cur"][live])
            assert np.array_equal(sg["err"][live], so["err"][live])
    # the stream is trackable except at the wrap from fixture 15 back to fixture 1 (pairs 14, 29, 44, 59)
    good = [m for i, m in enumerate(n_matched)]
    assert np.median(good) > 40
    fe.close()
