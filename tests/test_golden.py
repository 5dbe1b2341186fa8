"""Golden-vector tests: tests/golden/ba_factors.npz (independent NumPy restatement, see
tests/golden/make_golden.py) against (1) the CPU oracle and (2) the device math header compiled for
the host (tests/native).  The GPU kernels are checked against the same file in test_gpu_golden."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_api as o
import vplines_slam_amd as v

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "ba_factors.npz"))
TOL = 1e-11


def _pre_from_pack(pack):
    p = v.Preintegration()
    p.sum_dt = pack[0]
    for k in range(3):
        p.delta_p[k] = pack[1 + k]
        p.delta_v[k] = pack[8 + k]
        p.linearized_ba[k] = pack[11 + k]
        p.linearized_bg[k] = pack[14 + k]
    for k in range(4):
        p.delta_q[k] = pack[4 + k]
    for k in range(225):
        p.jacobian[k] = pack[17 + k]
        p.covariance[k] = 1.0 if k // 15 == k % 15 else 0.0   # identity => whitening is the identity
    return p


def check_all(proj, line, vp, pplus, oplus, o2p, imu_raw):
    r, _ = proj(G["proj_params"], G["proj_pts"])
    assert np.abs(r - G["proj_res"]).max() < TOL * max(1, np.abs(G["proj_res"]).max())
    r, _ = line(G["line_params"], G["line_obs"])
    assert np.abs(r - G["line_res"]).max() < TOL * max(1, np.abs(G["line_res"]).max())
    r, _ = vp(G["line_params"], G["vp_obs"])
    assert np.abs(r - G["vp_res"]).max() < TOL * max(1, np.abs(G["vp_res"]).max())
    assert np.abs(pplus(G["pose_x"], G["pose_delta"]) - G["pose_plus"]).max() < 1e-14
    assert np.abs(oplus(G["line_params"][:, 14:], G["orth_delta"]) - G["orth_plus"]).max() < 1e-12
    if o2p is not None:
        for i in range(len(G["orth_plk"])):
            assert np.abs(o2p(G["line_params"][i, 14:]) - G["orth_plk"][i]).max() < 1e-14
    r = imu_raw(G["imu_params"], G["imu_pre"])
    assert np.abs(r - G["imu_res_raw"]).max() < 1e-12


def test_oracle_against_golden():
    def imu(params, packs):
        pre = (v.Preintegration * len(packs))()
        for i, pk in enumerate(packs):
            C.memmove(C.byref(pre[i]), C.byref(_pre_from_pack(pk)), C.sizeof(pre[i]))
        return o.imu_factor(params, pre, want_jac=False)[0]
    check_all(lambda p, c: o.projection_factor(p, c, want_jac=False), lambda p, c: o.line_factor(p, c, want_jac=False),
              lambda p, c: o.vp_factor(p, c, want_jac=False), o.pose_plus, o.line_orth_plus, o.orth_to_plk, imu)


@pytest.fixture(scope="module")
def hostcheck():
    """device math header compiled with g++ (test-only; the product library has no CPU path)"""
    src = os.path.join(HERE, "native", "devmath_hostcheck.cpp")
    lib = os.path.join(HERE, "native", "libdevmath_hostcheck.so")
    deps = [src, os.path.join(HERE, "..", "vplines-slam_amd", "csrc", "vpl_math.h"),
            os.path.join(HERE, "..", "vplines-slam_amd", "csrc", "vpl_preint.h")]
    if not os.path.exists(lib) or any(os.path.getmtime(d) > os.path.getmtime(lib) for d in deps):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-o", lib, src])
    return C.CDLL(lib)


def _P(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def test_device_math_on_host_against_golden_and_oracle(hostcheck):
    hc = hostcheck

    def fac(fn, nres, njac):
        def f(params, consts, sqrt_info):
            params = np.ascontiguousarray(params); consts = np.ascontiguousarray(consts)
            r = np.zeros((len(params), nres)); j = np.zeros((len(params), njac))
            for i in range(len(params)):
                fn(_P(params[i]), _P(consts[i]), C.c_double(sqrt_info), _P(r[i]), _P(j[i]))
            return r, j
        return f
    proj = fac(hc.hc_projection_factor, 2, 44)
    line = fac(hc.hc_line_factor, 2, 36)
    vp = fac(hc.hc_vp_factor, 2, 36)

    def plus(fn, nx):
        def f(x, d):
            x = np.ascontiguousarray(x); d = np.ascontiguousarray(d); out = np.zeros_like(x)
            for i in range(len(x)):
                fn(_P(x[i]), _P(d[i]), _P(out[i]))
            return out
        return f

    def imu(params, packs):
        out = np.zeros((len(packs), 15))
        for i, pk in enumerate(packs):
            pre = _pre_from_pack(pk)
            hc.hc_imu_factor_raw(_P(np.ascontiguousarray(params[i])), C.byref(pre), C.c_double(9.81007), _P(out[i]), None)
        return out
    check_all(lambda p, c: proj(p, c, 460 / 1.5), lambda p, c: line(p, c, 306.666666667), lambda p, c: vp(p, c, 10.0),
              plus(hc.hc_pose_plus, 7), plus(hc.hc_line_orth_plus, 4), None, imu)
    # Jacobians: device math vs oracle (the oracle's are finite-difference checked in test_oracle_factors)
    _, jd = proj(G["proj_params"], G["proj_pts"], 460 / 1.5)
    _, jo = o.projection_factor(G["proj_params"], G["proj_pts"])
    assert np.abs(jd - jo).max() < 1e-10 * np.abs(jo).max()
    _, jd = line(G["line_params"], G["line_obs"], 306.666666667)
    _, jo = o.line_factor(G["line_params"], G["line_obs"])
    assert np.abs(jd - jo).max() < 1e-10 * np.abs(jo).max()
    _, jd = vp(G["line_params"], G["vp_obs"], 10.0)
    _, jo = o.vp_factor(G["line_params"], G["vp_obs"])
    assert np.abs(jd - jo).max() < 1e-10 * max(1.0, np.abs(jo).max())


def test_device_preintegration_on_host_matches_oracle(hostcheck):
    opt = v.default_options()
    cfg = v.workload.config(4, 0, False)
    w = v.workload.generate(77, cfg, 0.3)
    o.preintegrate_windows([w], opt)
    imu = w.extra["imu_samples"]
    for j in range(1, 6):
        out = v.Preintegration()
        s = np.ascontiguousarray(imu[j])
        a0 = np.ascontiguousarray(w.extra["imu_acc0"][j]); g0 = np.ascontiguousarray(w.extra["imu_gyr0"][j])
        z = np.zeros(3)
        hostcheck.hc_preintegrate(s.shape[0], _P(s), _P(a0), _P(g0), _P(z), _P(z), C.byref(opt), C.byref(out))
        for f in ("delta_p", "delta_q", "delta_v", "jacobian", "covariance"):
            a, b = np.array(getattr(out, f)), np.array(getattr(w.preint[j], f))
            assert np.abs(a - b).max() <= 1e-13 * max(1e-30, np.abs(b).max())


@pytest.mark.gpu
def test_gpu_against_golden(gpu_ctx):
    def imu(params, packs):
        # whitening with covariance = I is the identity
        pre = (v.Preintegration * len(packs))()
        for i, pk in enumerate(packs):
            C.memmove(C.byref(pre[i]), C.byref(_pre_from_pack(pk)), C.sizeof(pre[i]))
        return gpu_ctx.imu_factor(params, pre, want_jac=False)[0]
    check_all(lambda p, c: gpu_ctx.projection_factor(p, c, want_jac=False), lambda p, c: gpu_ctx.line_factor(p, c, want_jac=False),
              lambda p, c: gpu_ctx.vp_factor(p, c, want_jac=False), gpu_ctx.pose_plus, gpu_ctx.line_orth_plus, None, imu)
