"""Synthetic sliding windows via the host-only generator workload/synth.cpp."""
import ctypes as C

import numpy as np

from . import _build
from .capi import NF, Window

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class Config(C.Structure):
    _fields_ = [("n_points", C.c_int), ("n_lines", C.c_int), ("track_len", C.c_int), ("with_vp", C.c_int),
                ("imu_rate_div", C.c_int), ("kf_dt", C.c_double), ("pose_sigma_p", C.c_double), ("pose_sigma_theta_deg", C.c_double),
                ("vel_sigma", C.c_double), ("frame_sigma_scale", C.c_double), ("pix_sigma", C.c_double), ("depth_rel_sigma", C.c_double),
                ("orth_sigma", C.c_double), ("acc_n", C.c_double), ("gyr_n", C.c_double), ("ba_sigma", C.c_double),
                ("bg_sigma", C.c_double), ("triad_roll_deg", C.c_double), ("line_dir_sign", C.c_int)]


_lib = None


def _load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_build.build_workload())
        _lib.vplw_default_config.argtypes = [C.POINTER(Config), C.c_int, C.c_int, C.c_int]
        _lib.vplw_generate.argtypes = [C.c_uint64, C.POINTER(Config), C.c_double] + [_dp] * 5 + [_ip, _ip, _dp, _dp,
                                                                                             _ip, _ip, _dp, _dp,
                                                                                             _dp, _dp, _dp]
    return _lib


def config(n_points=200, n_lines=80, with_vp=True):
    c = Config()
    _load().vplw_default_config(C.byref(c), n_points, n_lines, 1 if with_vp else 0)
    return c


def seed_for(config_id, window_index):
    """SURVEY.md 8d seeds one window as 0x5EED0000 + config*1000 + window_index; with more than 1000 windows per config
    (the 512-window batch uses two seeds per window) those ranges overlap, so the config stride is 2**20 here."""
    assert 0 <= window_index < (1 << 20)
    return 0x5EED0000 + (config_id << 20) + window_index


def generate(seed, cfg, t_start):
    """Returns a Window without pre-integrations (raw IMU samples are in .extra)."""
    lib = _load()
    P, L, TL, ns = cfg.n_points, cfg.n_lines, cfg.track_len, cfg.imu_rate_div
    f8 = np.float64
    pose, sb, ex = np.zeros((NF, 7), f8), np.zeros((NF, 9), f8), np.zeros(7, f8)
    pose_t, sb_t = np.zeros((NF, 7), f8), np.zeros((NF, 9), f8)
    ps, pn = np.zeros(max(P, 1), np.int32), np.zeros(max(P, 1), np.int32)
    pobs, invd = np.zeros((max(P, 1) * TL, 3), f8), np.zeros(max(P, 1), f8)
    ls, ln = np.zeros(max(L, 1), np.int32), np.zeros(max(L, 1), np.int32)
    lobs, lplk = np.zeros((max(L, 1) * TL, 8), f8), np.zeros((max(L, 1), 6), f8)
    imu = np.zeros((NF, ns, 7), f8)
    acc0, gyr0 = np.zeros((NF, 3), f8), np.zeros((NF, 3), f8)
    d = lambda a: a.ctypes.data_as(_dp)
    i = lambda a: a.ctypes.data_as(_ip)
    rc = lib.vplw_generate(C.c_uint64(seed), C.byref(cfg), float(t_start), d(pose), d(sb), d(ex), d(pose_t), d(sb_t),
                           i(ps), i(pn), d(pobs), d(invd), i(ls), i(ln), d(lobs), d(lplk), d(imu), d(acc0), d(gyr0))
    if rc != 0:
        raise RuntimeError("vplw_generate failed")
    w = Window(pose, sb, ex, ps[:P], pn[:P], pobs[:P * TL], invd[:P], ls[:L], ln[:L], lobs[:L * TL], lplk[:L])
    w.extra = dict(pose_true=pose_t, speed_bias_true=sb_t, imu_samples=imu, imu_acc0=acc0, imu_gyr0=gyr0,
                   seed=seed, t_start=t_start)
    return w


def imu_batch_arrays(windows):
    """Flattens the raw IMU intervals (frames 1..10 of every window) for vpl_preintegrate_batch."""
    samples, offset, nsamples, acc0, gyr0, ba, bg = [], [], [], [], [], [], []
    off = 0
    for w in windows:
        imu = w.extra["imu_samples"]
        for j in range(1, NF):
            samples.append(imu[j])
            offset.append(off)
            nsamples.append(imu.shape[1])
            off += imu.shape[1]
            acc0.append(w.extra["imu_acc0"][j])
            gyr0.append(w.extra["imu_gyr0"][j])
            ba.append(w.speed_bias[j, 3:6])   # IntegrationBase{acc_0, gyr_0, Bas[frame_count], Bgs[frame_count]}
            bg.append(w.speed_bias[j, 6:9])
    return (np.array(offset, np.int32), np.array(nsamples, np.int32), np.concatenate(samples, 0), np.array(acc0),
            np.array(gyr0), np.array(ba), np.array(bg))


def set_preintegrations(windows, pre_array):
    """pre_array: ctypes array of Preintegration, 10 per window, in imu_batch_arrays order."""
    k = 0
    for w in windows:
        for j in range(1, NF):
            C.memmove(C.byref(w.preint[j]), C.byref(pre_array[k]), C.sizeof(pre_array[k]))
            k += 1


# ---- config 4: the 752x480 frame stream (SURVEY.md 8d "Frame stream") -------------------------------------------
def mh04_fixtures():
    """The 15 EuRoC MH_04 frames the reference ships as test data (tests/golden/mh04_frames.npz, pixels only)."""
    import os
    path = os.path.join(_build.ROOT, "tests", "golden", "mh04_frames.npz")
    return np.load(path)["frames"]


def warp_homography(img, Hm):
    """dst(x, y) = bilinear sample of img at Hm @ (x, y, 1); border replicated.  float64 arithmetic, uint8 in, float out."""
    H, W = img.shape
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    d = Hm[2, 0] * xs + Hm[2, 1] * ys + Hm[2, 2]
    sx = np.clip((Hm[0, 0] * xs + Hm[0, 1] * ys + Hm[0, 2]) / d, 0.0, W - 1.0)
    sy = np.clip((Hm[1, 0] * xs + Hm[1, 1] * ys + Hm[1, 2]) / d, 0.0, H - 1.0)
    x0 = np.minimum(np.floor(sx).astype(np.int64), W - 2)
    y0 = np.minimum(np.floor(sy).astype(np.int64), H - 2)
    fx, fy = sx - x0, sy - y0
    a = img.astype(np.float64)
    return ((1 - fy) * ((1 - fx) * a[y0, x0] + fx * a[y0, x0 + 1]) + fy * ((1 - fx) * a[y0 + 1, x0] + fx * a[y0 + 1, x0 + 1]))


def frame_stream(n=64, base=None):
    """n frames 752x480 uint8: frame i < 15 is MH_04 fixture i; later frames are fixture i % 15 warped by a small
    homography about the image centre (rotation <= 0.5 deg, scale +-0.5 %, shift <= 3 px, perspective <= 2e-6 / px) plus
    N(0, 2^2) noise, seeded per frame with seed_for(4, i) (numpy PCG64).  Consecutive frames stay matchable."""
    fx = mh04_fixtures() if base is None else base
    out = []
    for i in range(n):
        if i < len(fx):
            out.append(fx[i].copy())
            continue
        rng = np.random.Generator(np.random.PCG64(seed_for(4, i)))
        th = np.deg2rad(rng.uniform(-0.5, 0.5))
        sc = 1.0 + rng.uniform(-0.005, 0.005)
        tx, ty = rng.uniform(-3, 3, 2)
        px, py = rng.uniform(-2e-6, 2e-6, 2)
        cx, cy = 375.5, 239.5
        A = np.array([[sc * np.cos(th), -sc * np.sin(th), tx], [sc * np.sin(th), sc * np.cos(th), ty], [px, py, 1.0]])
        C = np.array([[1, 0, cx], [0, 1, cy], [0, 0, 1.0]])
        Ci = np.array([[1, 0, -cx], [0, 1, -cy], [0, 0, 1.0]])
        im = warp_homography(fx[i % len(fx)], C @ A @ Ci) + rng.normal(0.0, 2.0, fx[0].shape)
        out.append(np.clip(np.rint(im), 0, 255).astype(np.uint8))
    return np.stack(out)


# ---- the benchmark batch (configs 2, 3, 5): windows with the prior of a warm-up solve on the preceding window ------
def primed_batch(ctx, global_ids, cfg, opt, config_id=3):
    """Builds the windows `global_ids` of the benchmark batch exactly as bench.py times them: for every id the preceding
    window A (no prior) and the window B one keyframe later (shard.window_seeds), IntegrationBase of both on the device,
    one batched solve of the A windows whose MARGIN_OLD priors become the priors of the B windows.  Returns (B, priors)
    where priors is the ctypes array that owns the memory B[i].prior points to (keep it alive)."""
    from . import shard
    from .capi import Prior
    seeds = [shard.window_seeds(config_id, g) for g in global_ids]
    A = [generate(sa, cfg, t) for (sa, sb, t) in seeds]
    B = [generate(sb, cfg, t + cfg.kf_dt) for (sa, sb, t) in seeds]
    pre = ctx.preintegrate(*imu_batch_arrays(A + B), opt)
    set_preintegrations(A + B, pre)
    pri, _ = ctx.solve_windows(A, opt)
    keep = (Prior * len(B))()
    C.memmove(keep, pri, C.sizeof(keep))
    for i, b in enumerate(B):
        b.prior = keep[i]
    return B, keep


def graft_long_tracks(seed, cfg, t_start, every=10):
    """The window generate(seed, cfg, t_start) with every `every`-th point track replaced by a track that starts in frame 0
    and is observed in all 11 frames (the same seed generated with track_len = 11: same trajectory, IMU samples and initial
    states).  With 6-frame tracks only, the frame that leaves is tied to poses 1..5 and the prior stays at 45 dims however
    long the chain; a live tracker keeps some features over the whole window, every pose enters the prior and stays:
    n = 75 = 10 poses + speed/bias 1 + extrinsic, the reference's steady state."""
    w = generate(seed, cfg, t_start)
    c11 = Config()
    C.memmove(C.byref(c11), C.byref(cfg), C.sizeof(c11))
    c11.track_len = NF
    wl = generate(seed, c11, t_start)
    assert np.array_equal(w.pose, wl.pose) and np.array_equal(w.extra["imu_samples"], wl.extra["imu_samples"])
    P = len(w.point_nobs)
    off = np.concatenate([[0], np.cumsum(w.point_nobs)])
    offl = np.concatenate([[0], np.cumsum(wl.point_nobs)])
    start, nobs, invd, obs = w.point_start.copy(), w.point_nobs.copy(), w.inv_depth.copy(), []
    for k in range(P):
        if k % every == 0:
            start[k], nobs[k], invd[k] = wl.point_start[k], wl.point_nobs[k], wl.inv_depth[k]
            obs.append(wl.point_obs[offl[k]:offl[k + 1]])
        else:
            obs.append(w.point_obs[off[k]:off[k + 1]])
    r = Window(w.pose, w.speed_bias, w.ex_pose, start, nobs, np.concatenate(obs) if obs else w.point_obs[:0], invd,
               w.line_start, w.line_nobs, w.line_obs, w.line_plk)
    r.extra = dict(w.extra)
    return r


def steady_point_obs(cfg, every=10):
    """point observations of one graft_long_tracks window"""
    P = cfg.n_points
    nlong = (P + every - 1) // every
    return (P - nlong) * cfg.track_len + nlong * NF


def steady_batch(ctx, global_ids, cfg, opt, config_id=3, chain=7, every=10):
    """The timed windows of `global_ids` (seeds and times of primed_batch's B windows) with a tenth of their point tracks
    living through the whole window (graft_long_tracks), behind a chain of `chain` preceding windows of the same kind one
    keyframe apart, the prior handed from solve to solve on the device (vpl_ba_upload_chained).  Prior in AND prior out then
    have the reference's steady-state size n = 75 (marginalization_factor.cpp:177-363) instead of the 45 dims that 6-frame
    tracks leave.  The context needs max_point_obs >= steady_point_obs(cfg).  B is uploaded (chained) when this returns:
    reset_state + solve re-run it.  Returns (B, n_prior)."""
    from . import shard
    seeds = [shard.window_seeds(config_id, g) for g in global_ids]
    B = [graft_long_tracks(sb, cfg, t + cfg.kf_dt, every) for (sa, sb, t) in seeds]
    chains = [[graft_long_tracks(seed_for(config_id, (1 << 18) + chain * g + k), cfg, t + cfg.kf_dt - (chain - k) * cfg.kf_dt, every)
               for (g, (sa, sb, t)) in zip(global_ids, seeds)] for k in range(chain)]
    allw = [w for step in chains for w in step] + B
    # IntegrationBase of every interval on the device, in pieces
    for i in range(0, len(allw), 2048):
        piece = allw[i:i + 2048]
        set_preintegrations(piece, ctx.preintegrate(*imu_batch_arrays(piece), opt))
    for k, step in enumerate(chains):
        ctx.upload(step, opt, chained=k > 0)
        ctx.solve()
        ctx.synchronize()
    _, rep = ctx.download()
    n_prior = int(round(sum(rep[i].prior_n for i in range(len(B))) / max(1, len(B))))
    ctx.upload(B, opt, chained=True)
    return B, n_prior
