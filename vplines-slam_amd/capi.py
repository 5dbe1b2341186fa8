"""ctypes bindings of include/vplines_ba.h (the C ABI of the HIP library)."""
import ctypes as C
import time
import os

import numpy as np

from . import _build

NF = 11
MAX_PRIOR_BLOCKS = 23
MAX_PRIOR_DIM = 171
MARGIN_OLD, MARGIN_SECOND_NEW, MARGIN_NONE = 0, 1, -1
BLOCK_POSE, BLOCK_SPEEDBIAS, BLOCK_EXPOSE = 0, 1, 2

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class BaOptions(C.Structure):
    _fields_ = [("num_iterations", C.c_int), ("estimate_extrinsic", C.c_int), ("marginalization_flag", C.c_int),
                ("remove_line_outliers", C.c_int), ("focal_length", C.c_double), ("line_factor", C.c_double),
                ("vp_factor", C.c_double), ("g_norm", C.c_double), ("acc_n", C.c_double), ("gyr_n", C.c_double),
                ("acc_w", C.c_double), ("gyr_w", C.c_double), ("huber_delta", C.c_double)]


def default_options():
    """EuRoC values (config/euroc/euroc_config.yaml) with the BASELINE metric's 5 iterations."""
    o = BaOptions()
    o.num_iterations = 5
    o.estimate_extrinsic = 1
    o.marginalization_flag = MARGIN_OLD
    o.remove_line_outliers = 0
    o.focal_length = 460.0
    o.line_factor = 306.666666667
    o.vp_factor = 10.0
    o.g_norm = 9.81007
    o.acc_n, o.gyr_n, o.acc_w, o.gyr_w = 0.08, 0.004, 0.00004, 2.0e-6
    o.huber_delta = 1.0
    return o


class Preintegration(C.Structure):
    _fields_ = [("sum_dt", C.c_double), ("delta_p", C.c_double * 3), ("delta_q", C.c_double * 4),
                ("delta_v", C.c_double * 3), ("linearized_ba", C.c_double * 3), ("linearized_bg", C.c_double * 3),
                ("jacobian", C.c_double * 225), ("covariance", C.c_double * 225)]


class Prior(C.Structure):
    _fields_ = [("n", C.c_int), ("n_blocks", C.c_int), ("block_kind", C.c_int * MAX_PRIOR_BLOCKS),
                ("block_frame", C.c_int * MAX_PRIOR_BLOCKS), ("block_idx", C.c_int * MAX_PRIOR_BLOCKS),
                ("x0", (C.c_double * 9) * MAX_PRIOR_BLOCKS), ("J0", C.c_double * (MAX_PRIOR_DIM * MAX_PRIOR_DIM)),
                ("r0", C.c_double * MAX_PRIOR_DIM)]

    def J(self):
        n = self.n
        return np.ctypeslib.as_array(self.J0)[: n * n].reshape(n, n).copy()

    def r(self):
        return np.ctypeslib.as_array(self.r0)[: self.n].copy()


class CWindow(C.Structure):
    _fields_ = [("pose", (C.c_double * 7) * NF), ("speed_bias", (C.c_double * 9) * NF), ("ex_pose", C.c_double * 7),
                ("n_points", C.c_int), ("point_start", _ip), ("point_nobs", _ip), ("point_obs", _dp),
                ("inv_depth", _dp),
                ("n_lines", C.c_int), ("line_start", _ip), ("line_nobs", _ip), ("line_obs", _dp), ("line_plk", _dp),
                ("line_removed", _ip), ("line_triangulated", _ip),
                ("preint", Preintegration * NF), ("has_prior", C.c_int), ("prior", C.POINTER(Prior)),
                ("failure_occur", C.c_int), ("last_P0", C.c_double * 3), ("last_R0", C.c_double * 9),
                ("line_orth", C.POINTER(C.c_double))]


class SolveReport(C.Structure):
    _fields_ = [("iterations", C.c_int), ("num_successful_steps", C.c_int), ("termination", C.c_int),
                ("initial_cost", C.c_double), ("final_cost", C.c_double), ("n_lines_removed", C.c_int),
                ("prior_m", C.c_int), ("prior_n", C.c_int)]


def _arr(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


class Window:
    """numpy-backed owner of one vpl_window (keeps the arrays alive)."""

    def __init__(self, pose, speed_bias, ex_pose, point_start, point_nobs, point_obs, inv_depth, line_start,
                 line_nobs, line_obs, line_plk, preint=None, prior=None):
        self.pose = _arr(pose, np.float64).reshape(NF, 7).copy()
        self.speed_bias = _arr(speed_bias, np.float64).reshape(NF, 9).copy()
        self.ex_pose = _arr(ex_pose, np.float64).reshape(7).copy()
        self.point_start = _arr(point_start, np.int32).copy()
        self.point_nobs = _arr(point_nobs, np.int32).copy()
        self.point_obs = _arr(point_obs, np.float64).reshape(-1, 3).copy()
        self.inv_depth = _arr(inv_depth, np.float64).copy()
        self.line_start = _arr(line_start, np.int32).copy()
        self.line_nobs = _arr(line_nobs, np.int32).copy()
        self.line_obs = _arr(line_obs, np.float64).reshape(-1, 8).copy()
        self.line_plk = _arr(line_plk, np.float64).reshape(-1, 6).copy()
        self.line_removed = np.zeros(max(len(self.line_start), 1), np.int32)   # out: removeLineOutlier flags
        self.line_triangulated = np.ones(max(len(self.line_start), 1), np.int32)   # is_triangulation (in/out)
        self.preint = preint if preint is not None else (Preintegration * NF)()
        self.prior = prior
        self.failure = None       # (last_P0 [3], last_R0 [3,3]) => failure_occur = 1 (estimator.cpp:818-823)
        self.extra = {}

    def copy(self):
        pre = (Preintegration * NF)()
        C.memmove(pre, self.preint, C.sizeof(pre))
        w = Window(self.pose, self.speed_bias, self.ex_pose, self.point_start, self.point_nobs, self.point_obs,
                   self.inv_depth, self.line_start, self.line_nobs, self.line_obs, self.line_plk, pre, self.prior)
        w.extra = dict(self.extra)
        w.failure = self.failure
        w.line_triangulated[:] = self.line_triangulated
        return w

    def to_c(self, cw=None):
        cw = cw if cw is not None else CWindow()
        C.memmove(cw.pose, self.pose.ctypes.data, NF * 7 * 8)
        C.memmove(cw.speed_bias, self.speed_bias.ctypes.data, NF * 9 * 8)
        C.memmove(cw.ex_pose, self.ex_pose.ctypes.data, 7 * 8)
        cw.n_points = len(self.point_start)
        cw.point_start = self.point_start.ctypes.data_as(_ip)
        cw.point_nobs = self.point_nobs.ctypes.data_as(_ip)
        cw.point_obs = self.point_obs.ctypes.data_as(_dp)
        cw.inv_depth = self.inv_depth.ctypes.data_as(_dp)
        cw.n_lines = len(self.line_start)
        cw.line_start = self.line_start.ctypes.data_as(_ip)
        cw.line_nobs = self.line_nobs.ctypes.data_as(_ip)
        cw.line_obs = self.line_obs.ctypes.data_as(_dp)
        cw.line_plk = self.line_plk.ctypes.data_as(_dp)
        cw.line_removed = self.line_removed.ctypes.data_as(_ip)
        cw.line_triangulated = self.line_triangulated.ctypes.data_as(_ip)
        C.memmove(cw.preint, self.preint, C.sizeof(Preintegration) * NF)
        cw.has_prior = 1 if self.prior is not None else 0
        cw.prior = C.pointer(self.prior) if self.prior is not None else C.POINTER(Prior)()
        cw.line_orth = C.POINTER(C.c_double)()
        cw.failure_occur = 0
        if self.failure is not None:
            cw.failure_occur = 1
            p0 = np.ascontiguousarray(self.failure[0], np.float64).reshape(3)
            r0 = np.ascontiguousarray(self.failure[1], np.float64).reshape(9)
            for k in range(3):
                cw.last_P0[k] = p0[k]
            for k in range(9):
                cw.last_R0[k] = r0[k]
        return cw

    def from_c(self, cw):
        """copy back the in/out fields after a solve"""
        self.pose[:] = np.ctypeslib.as_array(cw.pose).reshape(NF, 7)
        self.speed_bias[:] = np.ctypeslib.as_array(cw.speed_bias).reshape(NF, 9)
        self.ex_pose[:] = np.ctypeslib.as_array(cw.ex_pose)


class CSlideTracks(C.Structure):
    _fields_ = [("point_start", _ip), ("point_nobs", _ip), ("point_drop", _ip),
                ("line_start", _ip), ("line_nobs", _ip), ("line_drop", _ip)]


class SlideTracks:
    """vpl_slide_tracks of one window (include/vplines_ba.h)"""

    def __init__(self, n_points, n_lines):
        self.point_start, self.point_nobs, self.point_drop = (np.zeros(n_points, np.int32) for _ in range(3))
        self.line_start, self.line_nobs, self.line_drop = (np.zeros(n_lines, np.int32) for _ in range(3))

    def to_c(self, ct=None):
        ct = ct if ct is not None else CSlideTracks()
        for f, _ in CSlideTracks._fields_:
            setattr(ct, f, getattr(self, f).ctypes.data_as(_ip))
        return ct


_hip = None


def load_hip_library():
    """Loads libvplines_hip.so; raises (never falls back) when it is missing."""
    global _hip
    if _hip is not None:
        return _hip
    path = _build.HIP_LIB
    if not os.path.exists(path):
        raise RuntimeError("HIP extension %s is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950); "
                           "there is no CPU fallback" % path)
    # PyTorch ships a HIP runtime of its own: in a process that uses both, torch's must be the one that is loaded first (with
    # /opt/rocm's loaded and initialised first, torch.cuda afterwards reports "No HIP GPUs are available" -- seen on the GPU box)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    vp = C.c_void_p
    lib.vpl_ba_default_options.argtypes = [C.POINTER(BaOptions)]
    lib.vpl_ctx_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.vpl_ctx_destroy.argtypes = [vp]
    lib.vpl_ctx_destroy.restype = None
    lib.vpl_ctx_set_stream.argtypes = [vp, vp]
    lib.vpl_last_error.argtypes = [vp]
    lib.vpl_last_error.restype = C.c_char_p
    lib.vpl_preintegrate_batch.argtypes = [vp, C.c_int, _ip, _ip, _dp, _dp, _dp, _dp, _dp, C.POINTER(BaOptions),
                                           C.POINTER(Preintegration)]
    for name in ("vpl_projection_factor_evaluate", "vpl_line_factor_evaluate", "vpl_vp_factor_evaluate"):
        getattr(lib, name).argtypes = [vp, C.c_int, _dp, _dp, C.c_double, _dp, _dp]
    lib.vpl_imu_factor_evaluate.argtypes = [vp, C.c_int, _dp, C.POINTER(Preintegration), C.c_double, _dp, _dp]
    lib.vpl_prior_factor_evaluate.argtypes = [vp, C.POINTER(Prior), _dp, _dp, _dp]
    lib.vpl_pose_plus.argtypes = [vp, C.c_int, _dp, _dp, _dp]
    lib.vpl_line_orth_plus.argtypes = [vp, C.c_int, _dp, _dp, _dp]
    lib.vpl_ba_upload.argtypes = [vp, C.c_int, C.POINTER(CWindow), C.POINTER(BaOptions)]
    lib.vpl_ba_solve.argtypes = [vp]
    lib.vpl_ba_upload_chained.argtypes = [vp, C.c_int, C.POINTER(CWindow), C.POINTER(BaOptions)]
    lib.vpl_ba_reset_state.argtypes = [vp]
    lib.vpl_ba_download.argtypes = [vp, C.c_int, C.POINTER(CWindow), C.POINTER(Prior), C.POINTER(SolveReport)]
    lib.vpl_ctx_synchronize.argtypes = [vp]
    lib.vpl_ba_pack_states_device.argtypes = [vp, C.c_int, vp]
    lib.vpl_ba_solve_windows.argtypes = [vp, C.c_int, C.POINTER(CWindow), C.POINTER(BaOptions), C.POINTER(Prior),
                                         C.POINTER(SolveReport)]
    lib.vpl_ba_triangulate_lines.argtypes = [vp, C.c_int, C.POINTER(CWindow)]
    lib.vpl_ba_marginalize.argtypes = [vp, C.c_int, C.POINTER(CWindow), C.POINTER(BaOptions), C.c_int, C.POINTER(Prior), _ip, _ip]
    lib.vpl_ba_slide_window.argtypes = [vp, C.c_int, C.POINTER(CWindow), C.c_int, C.c_double, C.POINTER(CSlideTracks)]
    lib.vpl_ba_triangulate_points.argtypes = [vp, C.c_int, C.POINTER(CWindow), C.c_double]
    lib.vpl_ba_only_line_opt.argtypes = [vp, C.c_int, C.POINTER(CWindow), C.POINTER(BaOptions), C.POINTER(SolveReport)]
    for name in ("vpl_ba_triangulate_lines", "vpl_ba_marginalize", "vpl_ba_slide_window", "vpl_ba_triangulate_points",
                 "vpl_ba_only_line_opt"):
        getattr(lib, name + "_async").argtypes = getattr(lib, name).argtypes
    lib.vpl_ba_collect.argtypes = [vp]
    lib.vpl_ba_solve_odometry.argtypes = [vp, C.c_int, C.POINTER(CWindow), C.POINTER(BaOptions), C.c_double, C.POINTER(Prior),
                                          C.POINTER(SolveReport), C.POINTER(SolveReport)]
    lib.vpl_ba_enable_kernel_timing.argtypes = [vp, C.c_int]
    lib.vpl_ba_kernel_times.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_char_p), _dp, _ip]
    lib.vpl_ba_launch_profile.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_char_p), _dp, _ip]
    _hip = lib
    return lib


def _p(a):
    return a.ctypes.data_as(_dp)


class Context:
    """One vpl_ctx (device buffers + stream of one GPU)."""

    def __init__(self, device=0, max_windows=1, max_points=256, max_point_obs=2816, max_lines=128,
                 max_line_obs=1408, stream=None):
        self.lib = load_hip_library()
        self.h = C.c_void_p()
        rc = self.lib.vpl_ctx_create(C.byref(self.h), device, max_windows, max_points, max_point_obs, max_lines,
                                     max_line_obs)
        if rc != 0:
            raise RuntimeError("vpl_ctx_create failed: %d" % rc)
        if stream is not None:
            self.set_stream(stream)
        self._cw = None
        self._windows = []

    def debug_guards(self):
        """VPL_DEBUG_GUARDS=1 (set before the context is made): number of device arrays with a write behind their end"""
        return int(self.lib.vpl_ba_debug_guards(self.h))

    def close(self):
        if self.h:
            bad = self.debug_guards() if os.environ.get("VPL_DEBUG_GUARDS") == "1" else 0
            msg = self.lib.vpl_last_error(self.h).decode() if bad else ""
            self.lib.vpl_ctx_destroy(self.h)
            self.h = C.c_void_p()
            if bad:
                import sys
                print("vpl_ba_debug_guards: %d arrays overrun; %s" % (bad, msg), file=sys.stderr)    # (also when closed by __del__)
                raise RuntimeError("vpl_ba_debug_guards: %d arrays overrun; %s" % (bad, msg))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            msg = self.lib.vpl_last_error(self.h)
            raise RuntimeError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))

    def set_stream(self, stream_ptr):
        self._settle()
        self._check(self.lib.vpl_ctx_set_stream(self.h, C.c_void_p(stream_ptr)), "vpl_ctx_set_stream")

    def synchronize(self):
        self._check(self.lib.vpl_ctx_synchronize(self.h), "vpl_ctx_synchronize")
        self._after_collect()

    def collect(self):
        """vpl_ba_collect: completes the call that was enqueued with async_=True (its results are in the Windows / the returned
        objects afterwards)"""
        self._check(self.lib.vpl_ba_collect(self.h), "vpl_ba_collect")
        self._after_collect()

    def _settle(self):
        """an enqueued call is completed before anything else touches the context (the library does the same on its side)"""
        if getattr(self, "_pending_refs", None) is not None:
            self.collect()

    def _after_collect(self):
        fin, self._pending_fin = getattr(self, "_pending_fin", None), None
        self._pending_refs = None
        if fin:
            fin()

    def _call5(self, name, async_, refs, fin, *args):
        """one of the five line-map entry points, synchronous or enqueued (vpl_ba_<name>[_async])"""
        self._settle()
        fn = getattr(self.lib, name + ("_async" if async_ else ""))
        self._check(fn(self.h, *args), name)
        if async_:
            self._pending_refs, self._pending_fin = refs, fin      # keep the C arrays alive until the call is collected
        elif fin:
            fin()

    # ---- single factor evaluators -------------------------------------------------
    def _factor(self, fn, params, consts, sqrt_info, nres, njac, want_jac=True):
        params = _arr(params, np.float64)
        consts = _arr(consts, np.float64)
        n = params.shape[0]
        res = np.zeros((n, nres))
        jac = np.zeros((n, njac)) if want_jac else None
        rc = fn(self.h, n, _p(params), _p(consts), sqrt_info, _p(res), _p(jac) if want_jac else None)
        self._check(rc, fn.__name__)
        return res, jac

    def projection_factor(self, params, pts, sqrt_info=460.0 / 1.5, want_jac=True):
        return self._factor(self.lib.vpl_projection_factor_evaluate, params, pts, sqrt_info, 2, 44, want_jac)

    def line_factor(self, params, obs, sqrt_info=306.666666667, want_jac=True):
        return self._factor(self.lib.vpl_line_factor_evaluate, params, obs, sqrt_info, 2, 36, want_jac)

    def vp_factor(self, params, vp, sqrt_info=10.0, want_jac=True):
        return self._factor(self.lib.vpl_vp_factor_evaluate, params, vp, sqrt_info, 2, 36, want_jac)

    def imu_factor(self, params, pre_array, g_norm=9.81007, want_jac=True):
        params = _arr(params, np.float64)
        n = params.shape[0]
        res = np.zeros((n, 15))
        jac = np.zeros((n, 480)) if want_jac else None
        rc = self.lib.vpl_imu_factor_evaluate(self.h, n, _p(params), pre_array, g_norm, _p(res),
                                              _p(jac) if want_jac else None)
        self._check(rc, "vpl_imu_factor_evaluate")
        return res, jac

    def prior_factor(self, prior, params, want_jac=True):
        params = _arr(params, np.float64)
        n = prior.n
        res = np.zeros(n)
        jac = np.zeros(n * params.size) if want_jac else None
        rc = self.lib.vpl_prior_factor_evaluate(self.h, C.byref(prior), _p(params), _p(res),
                                                _p(jac) if want_jac else None)
        self._check(rc, "vpl_prior_factor_evaluate")
        return res, jac

    def pose_plus(self, x, delta):
        x = _arr(x, np.float64)
        delta = _arr(delta, np.float64)
        out = np.zeros_like(x)
        self._check(self.lib.vpl_pose_plus(self.h, x.shape[0], _p(x), _p(delta), _p(out)), "vpl_pose_plus")
        return out

    def line_orth_plus(self, x, delta):
        x = _arr(x, np.float64)
        delta = _arr(delta, np.float64)
        out = np.zeros_like(x)
        self._check(self.lib.vpl_line_orth_plus(self.h, x.shape[0], _p(x), _p(delta), _p(out)), "vpl_line_orth_plus")
        return out

    def preintegrate(self, offset, nsamples, samples, acc0, gyr0, ba, bg, opt):
        offset = _arr(offset, np.int32)
        nsamples = _arr(nsamples, np.int32)
        n = len(offset)
        out = (Preintegration * n)()
        samples, acc0, gyr0, ba, bg = (_arr(a, np.float64) for a in (samples, acc0, gyr0, ba, bg))
        rc = self.lib.vpl_preintegrate_batch(self.h, n, offset.ctypes.data_as(_ip), nsamples.ctypes.data_as(_ip),
                                             _p(samples), _p(acc0), _p(gyr0), _p(ba), _p(bg), C.byref(opt), out)
        self._check(rc, "vpl_preintegrate_batch")
        return out

    # ---- window solve ---------------------------------------------------------------
    def upload(self, windows, opt, chained=False):
        """chained=True: vpl_ba_upload_chained -- every window takes the prior the previous solve of this context left for it
        (device resident), the Windows' own .prior is ignored"""
        self._settle()
        n = len(windows)
        cw = (CWindow * n)()
        for i, w in enumerate(windows):
            w.to_c(cw[i])
        self._cw = cw
        self._windows = windows
        if chained:
            self._check(self.lib.vpl_ba_upload_chained(self.h, n, cw, C.byref(opt)), "vpl_ba_upload_chained")
        else:
            self._check(self.lib.vpl_ba_upload(self.h, n, cw, C.byref(opt)), "vpl_ba_upload")

    def solve(self):
        self._check(self.lib.vpl_ba_solve(self.h), "vpl_ba_solve")

    def reset_state(self):
        self._check(self.lib.vpl_ba_reset_state(self.h), "vpl_ba_reset_state")

    def download(self):
        n = len(self._windows)
        if n == 0 or self._cw is None:
            self._check(self.lib.vpl_ba_download(self.h, 0, None, None, None), "vpl_ba_download")   # (refused: nothing uploaded)
        priors = (Prior * n)()
        reports = (SolveReport * n)()
        self._check(self.lib.vpl_ba_download(self.h, n, self._cw, priors, reports), "vpl_ba_download")
        for i, w in enumerate(self._windows):
            w.from_c(self._cw[i])
        return priors, reports

    def solve_windows(self, windows, opt):
        self._settle()
        self.upload(windows, opt)
        self.solve()
        self.synchronize()
        return self.download()

    def solve_odometry(self, windows, opt, init_depth=5.0):
        """vpl_ba_solve_odometry: triangulate || (triangulateLine -> onlyLineOpt) -> solve, in place; (priors, line reports, reports)"""
        self._settle()
        n = len(windows)
        cw = (CWindow * n)()
        for i, w in enumerate(windows):
            w.to_c(cw[i])
        priors = (Prior * n)()
        lreps = (SolveReport * n)()
        reps = (SolveReport * n)()
        t0 = time.perf_counter()
        rc = self.lib.vpl_ba_solve_odometry(self.h, n, cw, C.byref(opt), init_depth, priors, lreps, reps)
        self.last_call_s = time.perf_counter() - t0      # the C call alone (tools/time_odometry.py)
        self._check(rc, "vpl_ba_solve_odometry")
        for i, w in enumerate(windows):
            w.from_c(cw[i])
        return priors, lreps, reps

    def pack_states_device(self, n, data_ptr):
        """vpl_ba_pack_states_device: the [n][183] states of the solved batch into a device buffer (raw pointer)"""
        self._check(self.lib.vpl_ba_pack_states_device(self.h, n, C.c_void_p(data_ptr)), "vpl_ba_pack_states_device")

    def marginalize(self, windows, opt, flag, async_=False):
        """vpl_ba_marginalize: (priors, m, n) of the windows' current states, no solve"""
        n = len(windows)
        cw = (CWindow * n)()
        for i, w in enumerate(windows):
            w.to_c(cw[i])
        priors = (Prior * n)()
        m = np.zeros(n, np.int32)
        nn = np.zeros(n, np.int32)
        self._call5("vpl_ba_marginalize", async_, (cw, windows, priors, m, nn, opt), None, n, cw, C.byref(opt), flag, priors,
                    m.ctypes.data_as(_ip), nn.ctypes.data_as(_ip))
        return priors, m, nn

    def triangulate_lines(self, windows, async_=False):
        """FeatureManager::triangulateLine on the device; updates line_plk / line_triangulated of the Windows in place"""
        n = len(windows)
        cw = (CWindow * n)()
        for i, w in enumerate(windows):
            w.to_c(cw[i])
        self._call5("vpl_ba_triangulate_lines", async_, (cw, windows), None, n, cw)

    def slide_window(self, windows, marginalization_flag, init_depth=5.0, async_=False):
        """Estimator::slideWindow; pose / speed_bias / inv_depth / line_plk of the Windows change in place,
        returns one SlideTracks (new start / nobs / dropped observation per track) per window"""
        n = len(windows)
        cw = (CWindow * n)()
        ct = (CSlideTracks * n)()
        res = []
        for i, w in enumerate(windows):
            w.to_c(cw[i])
            res.append(SlideTracks(len(w.point_start), len(w.line_start)))
            res[-1].to_c(ct[i])
        def fin():
            for i, w in enumerate(windows):
                w.from_c(cw[i])
        self._call5("vpl_ba_slide_window", async_, (cw, ct, windows, res), fin, n, cw, marginalization_flag, init_depth, ct)
        return res

    def triangulate_points(self, windows, init_depth=5.0, async_=False):
        """FeatureManager::triangulate on the device; updates inv_depth of the Windows in place"""
        n = len(windows)
        cw = (CWindow * n)()
        for i, w in enumerate(windows):
            w.to_c(cw[i])
        self._call5("vpl_ba_triangulate_points", async_, (cw, windows), None, n, cw, init_depth)

    def only_line_opt(self, windows, opt, async_=False):
        """Estimator::onlyLineOpt on the device; updates line_plk / line_removed in place, returns the reports"""
        n = len(windows)
        cw = (CWindow * n)()
        for i, w in enumerate(windows):
            w.to_c(cw[i])
        reps = (SolveReport * n)()
        self._call5("vpl_ba_only_line_opt", async_, (cw, windows, reps, opt), None, n, cw, C.byref(opt), reps)
        return reps

    def enable_kernel_timing(self, on=True):
        self._check(self.lib.vpl_ba_enable_kernel_timing(self.h, 1 if on else 0), "vpl_ba_enable_kernel_timing")

    def kernel_times(self):
        cnt = C.c_int(64)
        names = (C.c_char_p * 64)()
        ms = (C.c_double * 64)()
        launches = (C.c_int * 64)()
        self._check(self.lib.vpl_ba_kernel_times(self.h, C.byref(cnt), names, ms, launches), "vpl_ba_kernel_times")
        return {names[i].decode(): (ms[i], launches[i]) for i in range(cnt.value)}

    def launch_profile(self):
        """Launches of the last timed solve, in order: (kernel, ms, (linearised, new step, re-used step, evaluated))."""
        cnt = C.c_int(64)
        names = (C.c_char_p * 64)()
        ms = (C.c_double * 64)()
        act = (C.c_int * 256)()
        self._check(self.lib.vpl_ba_launch_profile(self.h, C.byref(cnt), names, ms, act), "vpl_ba_launch_profile")
        return [(names[i].decode(), ms[i], tuple(act[4 * i + k] for k in range(4))) for i in range(cnt.value)]
