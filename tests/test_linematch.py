"""Line front-end, KLT line matching.
CPU: the oracle's pyramid / Scharr stages against independent NumPy restatements (integer-exact), behaviour of
LineMatching::Matching on the MH_04 fixture pair, an identity pair, an exactly translated pair and the empty cases.
GPU: every stage of the HIP matcher against the oracle -- pyramid and derivative planes bit-exact, tracked key points /
status / error bit-exact (the kernel accumulates in the reference's pixel order), matches identical."""
import os

import numpy as np
import pytest

import oracle_api as o
import vplines_slam_amd as v

HERE = os.path.dirname(os.path.abspath(__file__))
IMGS = [np.load(os.path.join(HERE, "golden", "mh04_%d.npy" % i)) for i in (1, 2)]


def numpy_pyr_down(img):
    """cv::pyrDown for CV_8U: separable [1 4 6 4 1], reflect-101, (sum + 128) >> 8, size (n+1)//2"""
    a = np.pad(img.astype(np.int64), 2, mode="reflect")
    k = np.array([1, 4, 6, 4, 1])
    H, W = img.shape
    rows = sum(k[i] * a[:, i:i + W] for i in range(5))
    full = sum(k[j] * rows[j:j + H, :] for j in range(5))
    return ((full[::2, ::2] + 128) >> 8).astype(np.uint8)


def numpy_scharr(img):
    a = np.pad(img.astype(np.int64), 1, mode="reflect")
    sm = 3 * (a[:-2, :] + a[2:, :]) + 10 * a[1:-1, :]
    df = a[2:, :] - a[:-2, :]
    dx = sm[:, 2:] - sm[:, :-2]
    dy = 3 * (df[:, 2:] + df[:, :-2]) + 10 * df[:, 1:-1]
    return np.stack([dx, dy], axis=-1).astype(np.int16)


def shifted(img, sx, sy):
    """content moved by (+sx, +sy) pixels, borders replicated"""
    H, W = img.shape
    ys = np.clip(np.arange(H) - sy, 0, H - 1)
    xs = np.clip(np.arange(W) - sx, 0, W - 1)
    return np.ascontiguousarray(img[ys][:, xs])


def translate_lines(lines, sx, sy):
    out = lines.copy()
    out[:, [0, 2, 7]] += sx
    out[:, [1, 3, 8]] += sy
    out[:, 6] -= lines[:, 4] * sx + lines[:, 5] * sy
    return out


def test_oracle_pyramid_and_scharr_are_integer_exact():
    rng = np.random.default_rng(5)
    for img in (IMGS[0], rng.integers(0, 256, (61, 95), dtype=np.uint8), rng.integers(0, 256, (30, 40), dtype=np.uint8)):
        assert np.array_equal(o.pyr_down(img), numpy_pyr_down(img))
        assert np.array_equal(o.scharr(img), numpy_scharr(img))
    assert o.pyr_down(IMGS[0]).shape == (240, 376)


def test_oracle_matching_on_fixture_pair():
    la, lb = o.edlines(IMGS[0]), o.edlines(IMGS[1])
    ok, r2c, st = o.line_match(IMGS[0], IMGS[1], la, lb)
    assert ok and len(r2c) == len(la)
    m = r2c >= 0
    assert m.sum() >= 60
    # Anchors(): len/step + 2 points per line, first/last are the end points
    assert len(st["kps_ref"]) == int(sum(int(np.float32(l[9]) / np.float32(10)) + 2 for l in la))
    assert np.allclose(st["kps_ref"][0], la[0][0:2])
    # tracked points follow one dominant image motion; matched lines are displaced consistently with it
    good = st["status"] > 0
    flow = np.median((st["kps_cur"] - st["kps_ref"])[good], axis=0)
    assert 3 < np.hypot(*flow) < 20
    ca, cb = la[m][:, 7:9], lb[r2c[m]][:, 7:9]
    # mid points may slide along a line whose end points moved; the component across the line must follow the flow
    nrm = la[m][:, 4:6]
    across = np.abs(np.sum((cb - ca - flow) * nrm, axis=1))
    assert np.median(across) < 2.0 and np.mean(across < 6.0) > 0.9
    # a matched current line is used by few reference lines (duplicates only for split segments)
    assert len(set(r2c[m].tolist())) >= 0.9 * m.sum()
    # the switches only remove / add matches in the documented direction
    ok2, r2c_nofilter, _ = o.line_match(IMGS[0], IMGS[1], la, lb, o.lm_default_param(True, False))
    assert np.all((r2c == r2c_nofilter) | (r2c == -1))


def test_oracle_identity_and_translation():
    la = o.edlines(IMGS[0])
    ok, r2c, st = o.line_match(IMGS[0], IMGS[0], la, la)
    assert ok
    good = st["status"] > 0
    assert np.abs(st["kps_cur"][good] - st["kps_ref"][good]).max() < 1e-3     # zero motion is a fixed point of LK
    assert np.all(st["err"][good] < 1e-6)
    m = r2c >= 0
    assert m.mean() > 0.9 and np.all(r2c[m] == np.nonzero(m)[0])
    # exact integer translation of image content and of the lines
    sx, sy = 6, -4
    ok, r2c, st = o.line_match(IMGS[0], shifted(IMGS[0], sx, sy), la, translate_lines(la, sx, sy))
    inner = (st["status"] > 0) & (np.abs(st["kps_ref"][:, 0] - 376) < 300) & (np.abs(st["kps_ref"][:, 1] - 240) < 180)
    d = st["kps_cur"][inner] - st["kps_ref"][inner] - np.array([sx, sy], np.float32)
    assert np.mean(np.hypot(d[:, 0], d[:, 1]) < 0.05) > 0.95
    m = r2c >= 0
    assert m.mean() > 0.8 and np.all(r2c[m] == np.nonzero(m)[0])


def test_oracle_empty_and_small_inputs():
    la = o.edlines(IMGS[0])
    ok, r2c, _ = o.line_match(IMGS[0], IMGS[1], la[:0], la)
    assert not ok
    ok, r2c, _ = o.line_match(IMGS[0], IMGS[1], la, la[:0])
    assert not ok and np.all(r2c == -2)                         # Matching() returns before touching its output
    # 40 x 30 frame: buildOpticalFlowPyramid stops after level 1 (20 x 15), level 2 would be 10 x 8 <= 13
    rng = np.random.default_rng(1)
    small = (rng.integers(0, 64, (30, 40)) + np.linspace(0, 150, 40)[None, :]).astype(np.uint8)
    line = np.array([[5.0, 6.0, 33.0, 22.0, 0, 0, 0, 19.0, 14.0, np.hypot(28, 16)]])
    n = np.array([16.0, -28.0]) / np.hypot(28, 16)
    line[0, 4:6] = n
    line[0, 6] = -(n[0] * 5 + n[1] * 6)
    ok, r2c, st = o.line_match(small, small, line, line)
    assert ok and r2c[0] == 0 and len(st["status"]) == int(np.hypot(28, 16) / 10) + 2


# ---------------------------------------------------------------------------------------------------------------
def _gpu_vs_oracle(fe, imgs, pairs, lref, lcur, prm_gpu, prm_orc):
    r2c, ok = fe.match_batch(imgs, pairs, lref, lcur, prm_gpu)
    for i, (a, b) in enumerate(pairs):
        oko, ro, so = o.line_match(imgs[a], imgs[b], lref[i], lcur[i], prm_orc)
        assert ok[i] == int(oko)
        if not oko:
            assert np.all(r2c[i] == -2)
            continue
        sg = fe.match_debug_kps(i)
        assert np.array_equal(sg["kps_ref"], so["kps_ref"])
        assert np.array_equal(sg["status"], so["status"])
        live = so["status"] > 0
        assert np.array_equal(sg["kps_cur"][live], so["kps_cur"][live])          # bit-exact float32
        assert np.array_equal(sg["err"][live], so["err"][live])
        assert np.array_equal(sg["kp2line_cur"], so["kp2line_cur"])
        assert np.array_equal(r2c[i], ro)
    return r2c, ok


@pytest.mark.gpu
def test_gpu_pyramid_levels_match_oracle():
    imgs = np.stack(IMGS)
    fe = v.frontend.FrontendContext(device=0, max_images=2, width=752, height=480, max_lines=256)
    fe.match_reserve(1, 4096)
    la, lb = o.edlines(IMGS[0]), o.edlines(IMGS[1])
    fe.match_batch(imgs, [(0, 1)], [la], [lb])
    for n in range(2):
        cur = imgs[n]
        for level in range(4):
            px, d = fe.match_debug_level(n, level)
            assert np.array_equal(px, cur), (n, level)
            assert np.array_equal(d, o.scharr(cur)), (n, level)
            cur = o.pyr_down(cur)
    fe.close()


@pytest.mark.gpu
def test_gpu_matching_matches_oracle_bit_exact():
    imgs = np.stack(IMGS + [shifted(IMGS[0], 6, -4), IMGS[0][::-1].copy()])
    L = [o.edlines(im) for im in imgs]
    pairs = [(0, 1), (1, 0), (0, 0), (0, 2), (3, 3), (0, 3)]
    lref = [L[a] for a, _ in pairs]
    lcur = [L[b] for _, b in pairs]
    fe = v.frontend.FrontendContext(device=0, max_images=len(imgs), width=752, height=480, max_lines=256)
    fe.match_reserve(len(pairs), 4096)
    r2c, ok = _gpu_vs_oracle(fe, imgs, pairs, lref, lcur, None, None)
    assert all(ok) and (r2c[0] >= 0).sum() >= 60
    # without illumination adaptation / without the topological filter
    for ill, topo in ((False, True), (True, False), (False, False)):
        _gpu_vs_oracle(fe, imgs, pairs[:2], lref[:2], lcur[:2], v.frontend.default_match_param(ill, topo),
                       o.lm_default_param(ill, topo))
    # other LineMatching constructor arguments
    pg, po = v.frontend.default_match_param(), o.lm_default_param()
    for p in (pg, po):
        p.step, p.closest_line_threshold, p.line_matching_ratio, p.klt_error_threshold = 7, 1.5, 0.3, 10.0
    _gpu_vs_oracle(fe, imgs, pairs[:2], lref[:2], lcur[:2], pg, po)
    fe.close()


@pytest.mark.gpu
def test_gpu_matching_edge_cases():
    imgs = np.stack(IMGS)
    la, lb = o.edlines(IMGS[0]), o.edlines(IMGS[1])
    fe = v.frontend.FrontendContext(device=0, max_images=2, width=752, height=480, max_lines=256)
    fe.match_reserve(3, 2048)
    # empty reference / current list: Matching() returns false and leaves its output alone
    r2c, ok = _gpu_vs_oracle(fe, imgs, [(0, 1), (0, 1), (0, 1)], [la[:0], la, la[:5]], [lb, lb[:0], lb], None, None)
    assert ok == [0, 0, 1]
    # lines that leave the image: key points outside are dropped by the window test, not by a fault
    far = la[:4].copy()
    far[:, [0, 2]] += 740
    _gpu_vs_oracle(fe, imgs, [(0, 1)], [far], [lb], None, None)
    # capacity: more key points than reserved -> VPL_E_CAPACITY, no silent truncation
    fe2 = v.frontend.FrontendContext(device=0, max_images=2, width=752, height=480, max_lines=256)
    fe2.match_reserve(1, 64)
    with pytest.raises(RuntimeError, match="key points"):
        fe2.match_batch(imgs, [(0, 1)], [la], [lb])
    with pytest.raises(RuntimeError):
        fe2.match_upload([(0, 5)], [la], [lb])          # image index out of range
    fe2.close()
    fe.close()
    # small frames: the pyramid stops at level 1 (40 x 30 -> 20 x 15)
    rng = np.random.default_rng(1)
    small = (rng.integers(0, 64, (30, 40)) + np.linspace(0, 150, 40)[None, :]).astype(np.uint8)
    small2 = shifted(small, 1, 0)
    line = np.array([[5.0, 6.0, 33.0, 22.0, 0, 0, 0, 19.0, 14.0, np.hypot(28, 16)]])
    n = np.array([16.0, -28.0]) / np.hypot(28, 16)
    line[0, 4:6] = n
    line[0, 6] = -(n[0] * 5 + n[1] * 6)
    fs = v.frontend.FrontendContext(device=0, max_images=2, width=40, height=30, max_lines=8)
    fs.match_reserve(2, 64)
    ims = np.stack([small, small2])
    _gpu_vs_oracle(fs, ims, [(0, 0), (0, 1)], [line, line], [line, translate_lines(line, 1, 0)], None, None)
    with pytest.raises(RuntimeError):
        fs.match_debug_level(0, 2)
    fs.close()


@pytest.mark.gpu
def test_matcher_fed_from_the_detector_on_the_device_equals_the_host_hand_over():
    """vpl_match_from_detected: the lines of the last detect() go to the matcher device to device (sorted into the reference's
    order and converted to the wire layout by k_ed_sort_lines) -- same matches, same lines as download + vpl_match_upload."""
    import vplines_slam_amd as v
    imgs = v.workload.frame_stream(8)
    fe = v.frontend.FrontendContext(device=0, max_images=8, width=752, height=480, max_lines=256)
    fe.match_reserve(7, 8192)
    fe.upload(imgs)
    pairs = [(i, i + 1) for i in range(7)]
    fe.detect()
    fe.synchronize()
    lines = [l[:200] for l in fe.download()]
    fe.match_upload(pairs, [lines[a] for a, _ in pairs], [lines[b] for _, b in pairs])
    fe.match_run()
    fe.synchronize()
    r_host, ok_host = fe.match_download()
    fe.detect()
    fe.match_from_detected(pairs, 200)
    fe.match_run()
    fe.synchronize()
    r_dev, ok_dev = fe.match_download()
    lines2 = [l[:200] for l in fe.download()]
    assert ok_host == ok_dev and sum(ok_dev) >= 5
    for a, b in zip(r_host, r_dev):
        assert np.array_equal(a, b)
    for a, b in zip(lines, lines2):
        assert a.tobytes() == b.tobytes()
    nr, nc = fe.match_counts()
    assert nr == [len(lines[a]) for a, _ in pairs] and nc == [len(lines[b]) for _, b in pairs]
    fe.close()
