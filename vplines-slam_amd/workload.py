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
                ("vel_sigma", C.c_double), ("pix_sigma", C.c_double), ("depth_rel_sigma", C.c_double),
                ("orth_sigma", C.c_double), ("acc_n", C.c_double), ("gyr_n", C.c_double), ("ba_sigma", C.c_double),
                ("bg_sigma", C.c_double)]


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
    """seed = 0x5EED0000 + config*1000 + window_index"""
    return 0x5EED0000 + config_id * 1000 + window_index


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
