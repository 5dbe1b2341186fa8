"""LineFeatureTracker::readImage around the device kernels.
CPU: the id / quota list handling -- oracle (oracle/preproc.cpp), the product's host function vpl_line_track_ids and an
independent Python restatement agree on random inputs, quirks included.
GPU: the C++ LineFeatureTracker mirror over a frame sequence, replayed step by step with the oracle."""
import math
import os
import subprocess

import numpy as np
import pytest

import oracle_api as o
import vplines_slam_amd as v

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def py_track_ids(ends, id_prev, tcnt_prev, p2n, max_h, max_v, cnt):
    n_new = len(ends)
    ids, tc = [-1] * n_new, [0] * n_new
    for k, mt in enumerate(p2n):
        if 0 < mt < n_new:
            ids[mt] = int(id_prev[k])
            tc[mt] = (int(tcnt_prev[mt]) if mt < len(tcnt_prev) else 0) + 1

    def is_h(e):
        e = [np.float32(x) for x in e]
        a = math.atan2(e[3] - e[1], e[2] - e[0]) if e[2] > e[0] else math.atan2(e[1] - e[3], e[0] - e[2])
        return (3.14 / 4 <= a <= 3 * 3.14 / 4) or (-3 * 3.14 / 4 <= a <= -3.14 / 4)
    tracked, new_h, new_v = [], [], []
    for i in range(n_new):
        if ids[i] == -1:
            ids[i] = cnt
            cnt += 1
            (new_h if is_h(ends[i]) else new_v).append(i)
        else:
            tracked.append(i)
    h = sum(is_h(ends[i]) for i in tracked)
    vv = len(tracked) - h
    keep = tracked + new_h[:max(max_h - h, 0)] + new_v[:max(max_v - vv, 0)]
    return np.array(keep, np.int32), np.array([ids[i] for i in keep], np.int32), np.array(tc, np.int32), cnt, np.array(new_v, np.int32)


def test_track_ids_oracle_product_and_python_agree():
    hip = v.load_hip_library()
    rng = np.random.default_rng(2)
    for trial in range(60):
        n_new, n_prev = int(rng.integers(0, 60)), int(rng.integers(0, 50))
        ends = rng.uniform(0, 700, (n_new, 4)).astype(np.float32)
        if n_new > 3:
            ends[1, 2] = ends[1, 0]                              # a vertical segment (x2 == x1)
            ends[2] = [10, 10, 20, 20.00001]                     # right at the 45 degree border
        id_prev = rng.integers(0, 1000, n_prev).astype(np.int32)
        tcnt_prev = rng.integers(0, 9, int(rng.integers(0, 70))).astype(np.int32)
        p2n = rng.integers(-1, max(n_new, 1) + 3, n_prev).astype(np.int32)      # includes 0 (ignored) and out-of-range
        max_h, max_v = int(rng.integers(0, 40)), int(rng.integers(0, 40))
        a = o.track_ids(ends, id_prev, tcnt_prev, p2n, max_h, max_v, 500 + trial, with_vertical=True)
        b = o.track_ids(ends, id_prev, tcnt_prev, p2n, max_h, max_v, 500 + trial, lib=hip, fn="vpl_line_track_ids", with_vertical=True)
        c = py_track_ids(ends, id_prev, tcnt_prev, p2n, max_h, max_v, 500 + trial)
        for x, y, z in zip(a, b, c):
            assert np.array_equal(x, y) and np.array_equal(x, z)
    # the documented quirks: a match to detection 0 is dropped, tracked lines are never cut by the quota
    ends = np.array([[0, 0, 100, 0]] * 4, np.float32)
    keep, ids, tc, cnt = o.track_ids(ends, [7, 8, 9], [5, 5, 5, 5], [0, 1, 3], 0, 0, 100, lib=hip, fn="vpl_line_track_ids")
    assert list(keep) == [1, 3] and list(ids) == [8, 9] and list(tc) == [0, 6, 0, 6] and cnt == 102


def poly_hash(img):
    k = np.full(img.size, 1315423911, np.uint64)
    k[0] = 1
    with np.errstate(over="ignore"):
        powers = np.cumprod(k)
        return int((img.ravel()[::-1].astype(np.uint64) * powers).sum())


@pytest.mark.gpu
def test_line_feature_tracker_mirror_replays_with_oracle(tmp_path):
    from test_preproc import euroc_maps, oracle_clahe, oracle_remap
    exe = str(tmp_path / "line_tracker_check")
    libdir = os.path.join(ROOT, "vplines-slam_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "native", "line_tracker_check.cpp"),
                           "-L", libdir, "-lvplines_hip", "-Wl,-rpath," + libdir, "-o", exe])
    a, b = (np.load(os.path.join(ROOT, "tests", "golden", "mh04_%d.npy" % i)) for i in (1, 2))
    frames = np.stack([a, b, np.roll(b, (2, -3), (0, 1)), np.roll(b, (4, -7), (0, 1))])
    H, W = a.shape
    mx, my = euroc_maps(W, H)
    frames.tofile(str(tmp_path / "frames.raw"))
    mx.tofile(str(tmp_path / "mx.f32"))
    my.tofile(str(tmp_path / "my.f32"))
    MAXH, MAXV = 25, 40
    out = subprocess.check_output([exe, str(tmp_path / "frames.raw"), str(len(frames)), str(W), str(H), str(tmp_path / "mx.f32"),
                                   str(tmp_path / "my.f32"), str(MAXH), str(MAXV)], text=True)
    vals = {}
    for ln in out.strip().splitlines():
        parts = ln.split()
        vals[parts[0]] = parts[1:]
    key = lambda q: q[np.lexsort((np.round(q[:, 3], 2), np.round(q[:, 2], 2), np.round(q[:, 1], 2), np.round(q[:, 0], 2)))]
    prev = None
    cnt = 0
    fx, fy, cx, cy = np.float32(458.654), np.float32(457.296), np.float32(W // 2), np.float32(H // 2)
    n_tracked_total = n_vp = n_classified = 0
    for f in range(len(frames)):
        img = oracle_clahe(oracle_remap(frames[f], mx, my))
        assert int(vals["frame%d_img" % f][0]) == poly_hash(img)                       # prepared frame, bit for bit
        det_full = np.array(vals["frame%d_det" % f], np.float64).reshape(-1, 10)
        det = det_full[:, :4]
        ref_lines = o.edlines(img)
        assert len(det) == len(ref_lines) and np.abs(key(det) - key(ref_lines)[:, :4]).max() < 1e-3
        lines = np.array(vals["frame%d_lines" % f], np.float64).reshape(-1, 4)
        ids = np.array(vals["frame%d_ids" % f], np.int64)
        tcnt = np.array(vals["frame%d_tcnt" % f], np.int64)
        match = np.array(vals["frame%d_match" % f], np.int64)
        if prev is None:
            assert np.array_equal(lines, det) and np.array_equal(ids, np.arange(len(det))) and len(match) == 0
            assert not tcnt.any()
            cnt = len(det)
            keep = np.arange(len(det))
        else:
            # the matcher is replayed on the device's own line records (as test_gpu_host_adapter does)
            ok, r2c, _ = o.line_match(prev["img"], img, prev["full"], det_full)
            assert ok and np.array_equal(match, r2c)
            keep, want_ids, want_tc, cnt, vert = o.track_ids(det, prev["ids"], prev["tcnt"], r2c, MAXH, MAXV, cnt, with_vertical=True)
            assert np.array_equal(ids, want_ids) and np.array_equal(tcnt, want_tc)
            assert np.array_equal(lines, det[keep])
            # vanishing points: hypotheses from the new v-class lines when there are more than two, else from the kept lines
            hyp = det[vert] if len(vert) > 2 else lines
            seed = int(vals["frame%d_vps" % f][0])
            assert seed == 1000 + n_vp
            wv, wi, it = o.vp_detect(hyp, lines, fx, cx, cy, seed, n_vp == 0)
            n_vp += 1
            assert it == 105
            assert np.array_equal(np.array(vals["frame%d_vps" % f][1:], np.float64).reshape(3, 3), wv)
            assert np.array_equal(np.array(vals["frame%d_vpids" % f], np.int64), wi)
            vp4 = np.array(vals["frame%d_vp4" % f], np.float64).reshape(-1, 4)
            want4 = np.array([[0, 0, 0, 0] if k == 3 else list(wv[k]) + [1.0] for k in wi])
            assert np.array_equal(vp4, want4)
            n_classified += int((wi < 3).sum())
            n_tracked_total += int(np.isin(ids, prev["ids"]).sum())
        assert [int(x) for x in vals["frame%d_cnt" % f]] == [cnt, 1]
        obs = np.array(vals["frame%d_obs" % f], np.float64).reshape(-1, 9)
        assert np.array_equal(obs[:, 0].astype(int), ids)
        if prev is None:
            assert not obs[:, 5:].any()                                                 # first frame: no VP stage ran
        else:
            assert np.allclose(obs[:, 5:], vp4[0], rtol=1e-8)                           # every line carries vps[0] (node.cpp:104)
        l32 = lines.astype(np.float32)
        want = np.stack([(l32[:, 0] - cx) / fx, (l32[:, 1] - cy) / fy, (l32[:, 2] - cx) / fx, (l32[:, 3] - cy) / fy], 1)
        assert np.abs(obs[:, 1:5] - want).max() < 1e-6
        prev = dict(img=img, full=det_full[keep], ids=ids, tcnt=tcnt)
    assert n_tracked_total > 30                                                         # ids really propagate
    assert n_vp == len(frames) - 1 and n_classified > 10
