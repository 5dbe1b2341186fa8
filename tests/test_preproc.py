"""Image preparation of LineFeatureTracker::readImage (line_feature_tracker.cpp:62-68): cv::remap + CLAHE.
CPU: the C++ oracle against an independent vectorised NumPy restatement (bit-exact) and against properties.
GPU: the HIP kernels against the oracle (bit-exact), and the prepared batch feeding EDLines without a host round trip."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_api as o

HERE = os.path.dirname(os.path.abspath(__file__))
IMGS = [np.load(os.path.join(HERE, "golden", "mh04_%d.npy" % i)) for i in (1, 2)]
u8p, f32p = C.POINTER(C.c_uint8), C.POINTER(C.c_float)


def euroc_maps(W=752, H=480, scale=1.0):
    """maps as PinholeCamera::initUndistortRectifyMap builds them (PinholeCamera.cc:729-789) for the EuRoC cam0 model"""
    fx, fy, cx, cy = 458.654 * scale, 457.296 * scale, 367.215 * scale, 248.375 * scale
    k1, k2, p1, p2 = -0.28340811, 0.07395907, 0.00019359, 1.76187114e-05
    u, v = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    x, y = (u - W // 2) / fx, (v - H // 2) / fy
    r2 = x * x + y * y
    rad = k1 * r2 + k2 * r2 * r2
    dx = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    dy = y * rad + 2 * p2 * x * y + p1 * (r2 + 2 * y * y)
    return (fx * (x + dx) + cx).astype(np.float32), (fy * (y + dy) + cy).astype(np.float32)


def oracle_remap(img, mx, my):
    lib = o.load()
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty_like(img)
    H, W = img.shape
    mx, my = np.ascontiguousarray(mx, np.float32), np.ascontiguousarray(my, np.float32)
    lib.orc_remap_linear(img.ctypes.data_as(u8p), W, H, mx.ctypes.data_as(f32p), my.ctypes.data_as(f32p), out.ctypes.data_as(u8p))
    return out


def oracle_clahe(img, clip=3.0, tiles=(8, 8)):
    lib = o.load()
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty_like(img)
    H, W = img.shape
    lib.orc_clahe.argtypes = [u8p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, u8p]
    lib.orc_clahe(img.ctypes.data_as(u8p), W, H, clip, tiles[0], tiles[1], out.ctypes.data_as(u8p))
    return out


def numpy_remap(img, mx, my):
    H, W = img.shape
    sx = np.rint(mx * np.float32(32)).astype(np.int64)
    sy = np.rint(my * np.float32(32)).astype(np.int64)
    ix, iy, fx, fy = sx >> 5, sy >> 5, sx & 31, sy & 31
    pad = np.zeros((H + 2, W + 2), np.int64)
    pad[1:-1, 1:-1] = img

    def tap(dx, dy):
        px, py = np.clip(ix + dx + 1, 0, W + 1), np.clip(iy + dy + 1, 0, H + 1)
        return pad[py, px]
    acc = tap(0, 0) * (32 - fx) * (32 - fy) + tap(1, 0) * fx * (32 - fy) + tap(0, 1) * (32 - fx) * fy + tap(1, 1) * fx * fy
    return np.clip((acc * 32 + 16384) >> 15, 0, 255).astype(np.uint8)


def numpy_clahe(img, clip=3.0, tiles=(8, 8)):
    H, W = img.shape
    tX, tY = tiles
    ext = img
    if W % tX or H % tY:
        ext = np.pad(img, ((0, tY - H % tY), (0, tX - W % tX)), mode="reflect")
    tw, th = ext.shape[1] // tX, ext.shape[0] // tY
    area = tw * th
    lim = max(int(clip * area / 256), 1) if clip > 0 else 0
    scale = np.float32(255) / np.float32(area)
    luts = np.zeros((tY, tX, 256), np.float32)
    for ty in range(tY):
        for tx in range(tX):
            h = np.bincount(ext[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw].ravel(), minlength=256).astype(np.int64)
            if lim > 0:
                clipped = int(np.maximum(h - lim, 0).sum())
                h = np.minimum(h, lim) + clipped // 256
                res = clipped % 256
                if res:
                    step = max(256 // res, 1)
                    idx = np.arange(0, 256, step)[:res]
                    h[idx] += 1
            luts[ty, tx] = np.clip(np.rint(np.cumsum(h).astype(np.float32) * scale), 0, 255)
    f = np.float32
    txf = np.arange(W, dtype=f) * (f(1) / f(tw)) - f(0.5)
    tyf = np.arange(H, dtype=f) * (f(1) / f(th)) - f(0.5)
    tx1, ty1 = np.floor(txf).astype(int), np.floor(tyf).astype(int)
    xa, ya = (txf - tx1.astype(f))[None, :], (tyf - ty1.astype(f))[:, None]
    xa1, ya1 = f(1) - xa, f(1) - ya
    tx2, ty2 = np.minimum(tx1 + 1, tX - 1)[None, :], np.minimum(ty1 + 1, tY - 1)[:, None]
    tx1, ty1 = np.maximum(tx1, 0)[None, :], np.maximum(ty1, 0)[:, None]
    v = img.astype(int)
    res = (luts[ty1, tx1, v] * xa1 + luts[ty1, tx2, v] * xa) * ya1 + (luts[ty2, tx1, v] * xa1 + luts[ty2, tx2, v] * xa) * ya
    return np.clip(np.rint(res), 0, 255).astype(np.uint8)


def test_oracle_remap_matches_numpy_and_properties():
    img = IMGS[0]
    H, W = img.shape
    mx, my = euroc_maps(W, H)
    assert np.array_equal(oracle_remap(img, mx, my), numpy_remap(img, mx, my))
    u, v = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
    assert np.array_equal(oracle_remap(img, u, v), img)                              # identity
    sh = oracle_remap(img, u + 7, v - 3)                                             # integer shift, zero border
    assert np.array_equal(sh[3:, :W - 7], img[:H - 3, 7:]) and not sh[:3].any() and not sh[:, W - 7:].any()
    half = oracle_remap(img, u + 0.5, v)                                             # half-pixel: rounded mean of neighbours
    want = (img[:, :-1].astype(int) + img[:, 1:] + 1) >> 1
    assert np.array_equal(half[:, :-1], want)
    rng = np.random.default_rng(0)                                                   # wild maps incl. far outside
    wx, wy = (rng.uniform(-40, W + 40, (H, W))).astype(np.float32), (rng.uniform(-40, H + 40, (H, W))).astype(np.float32)
    wx[0, :4] = [-1e6, 1e6, -0.5, W - 0.5]
    assert np.array_equal(oracle_remap(img, wx, wy), numpy_remap(img, wx, wy))


def test_oracle_clahe_matches_numpy_and_properties():
    for img in (IMGS[0], IMGS[1][:, ::-1]):
        assert np.array_equal(oracle_clahe(img), numpy_clahe(img))
    small = np.ascontiguousarray(IMGS[0][:70, :100])                                 # not a multiple of the grid: reflect-101 pad
    assert np.array_equal(oracle_clahe(small), numpy_clahe(small))
    assert np.array_equal(oracle_clahe(small, 40.0, (4, 3)), numpy_clahe(small, 40.0, (4, 3)))
    assert np.array_equal(oracle_clahe(IMGS[0], 0.0), numpy_clahe(IMGS[0], 0.0))     # no clipping = plain AHE
    flat = np.full((480, 752), 77, np.uint8)
    out = oracle_clahe(flat)
    assert len(np.unique(out)) == 1                                                  # constant in, constant out
    eq = oracle_clahe(IMGS[0])
    assert eq.std() > IMGS[0].std()                                                  # contrast went up
    # monotone inside one tile centre row: a brighter input pixel never maps below a darker one at the same place
    assert np.array_equal(oracle_clahe(eq), numpy_clahe(eq))


@pytest.fixture(scope="module")
def fe():
    import vplines_slam_amd as v
    f = v.frontend.FrontendContext(device=0, max_images=6, width=752, height=480, max_lines=1024)
    yield f
    f.close()


def synth_frames(n):
    rng = np.random.default_rng(3)
    fr = []
    for i in range(n):
        base = IMGS[i % 2].astype(np.int32)
        fr.append(np.clip(np.roll(base, (3 * i, -5 * i), (0, 1)) + rng.integers(-6, 7, base.shape), 0, 255).astype(np.uint8))
    return np.stack(fr)


@pytest.mark.gpu
def test_gpu_preproc_matches_oracle(fe):
    raw = synth_frames(5)
    mx, my = euroc_maps()
    fe.set_maps(mx, my)
    fe.pre_upload(raw)
    fe.pre_run(True, 3.0, (8, 8))
    got = fe.pre_download()
    for i in range(len(raw)):
        assert np.array_equal(got[i], oracle_clahe(oracle_remap(raw[i], mx, my)))
    fe.pre_run(False)                                                                # remap only
    got = fe.pre_download()
    assert np.array_equal(got[2], oracle_remap(raw[2], mx, my))
    fe.pre_run(True, 40.0, (4, 3))
    assert np.array_equal(fe.pre_download()[1], oracle_clahe(oracle_remap(raw[1], mx, my), 40.0, (4, 3)))
    fe.set_maps(None, None)                                                          # CLAHE only / plain copy
    fe.pre_run(True, 3.0, (8, 8))
    assert np.array_equal(fe.pre_download()[3], oracle_clahe(raw[3]))
    fe.pre_run(False)
    assert np.array_equal(fe.pre_download(), raw)
    rng = np.random.default_rng(1)                                                   # maps that leave the frame
    wx = rng.uniform(-40, 792, (480, 752)).astype(np.float32)
    wy = rng.uniform(-40, 520, (480, 752)).astype(np.float32)
    fe.set_maps(wx, wy)
    fe.pre_run(False)
    assert np.array_equal(fe.pre_download()[0], oracle_remap(raw[0], wx, wy))


@pytest.mark.gpu
def test_gpu_preproc_odd_size_and_chain_into_edlines(fe):
    import vplines_slam_amd as v
    # a frame size that is neither a multiple of 4 nor of the tile grid
    W, H = 301, 203
    f2 = v.frontend.FrontendContext(device=0, max_images=2, width=W, height=H, max_lines=256)
    raw = np.ascontiguousarray(synth_frames(2)[:, 100:100 + H, 200:200 + W])
    u, vv = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
    mx, my = u * 0.97 + 3.3, vv * 1.02 - 1.7
    f2.set_maps(mx, my)
    f2.pre_upload(raw)
    f2.pre_run(True, 3.0, (8, 8))
    got = f2.pre_download()
    for i in range(2):
        assert np.array_equal(got[i], oracle_clahe(oracle_remap(raw[i], mx, my)))
    f2.close()
    # prepared frames feed the detector directly: same lines as uploading the oracle-prepared frames
    raw = synth_frames(3)
    mx, my = euroc_maps()
    fe.set_maps(mx, my)
    fe.pre_upload(raw)
    fe.pre_run(True, 3.0, (8, 8))
    fe.detect()
    fe.synchronize()
    la = fe.download()
    prepared = np.stack([oracle_clahe(oracle_remap(r, mx, my)) for r in raw])
    fe.upload(prepared)
    fe.detect()
    fe.synchronize()
    lb = fe.download()
    assert min(len(a) for a in la) > 20
    assert all(np.array_equal(a, b) for a, b in zip(la, lb))
