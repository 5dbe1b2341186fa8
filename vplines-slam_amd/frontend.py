"""ctypes binding of include/vplines_frontend.h (EDLines extractor + KLT line matcher on the GPU)."""
import ctypes as C
import os

import numpy as np

from .capi import load_hip_library


class EdlineParam(C.Structure):
    _fields_ = [("ksize", C.c_int), ("sigma", C.c_float), ("gradientThreshold", C.c_float),
                ("anchorThreshold", C.c_float), ("scanIntervals", C.c_int), ("minLineLen", C.c_int),
                ("lineFitErrThreshold", C.c_double)]


class Line(C.Structure):
    _fields_ = [("line_endpoint", C.c_float * 4), ("line_equation", C.c_double * 3), ("center", C.c_float * 2),
                ("length", C.c_float)]


class MatchParam(C.Structure):
    _fields_ = [("step", C.c_int), ("closest_line_threshold", C.c_float), ("line_matching_ratio", C.c_float),
                ("line_distance_error_ratio", C.c_float), ("klt_error_threshold", C.c_float),
                ("illumination_adapt", C.c_int), ("topological_filter", C.c_int),
                ("topo_distance_threshold", C.c_float), ("topo_length_tolerate_ratio", C.c_float),
                ("topo_violation_ratio", C.c_float)]


def default_match_param(illumination_adapt=True, topological_filter=True):
    """LineMatching() defaults (line_matching.h:14-18,45-47) with the tracker's flags (line_feature_tracker.cpp:307-308)"""
    return MatchParam(10, 0.5, 0.4, 3.0, 40.0, int(illumination_adapt), int(topological_filter), 15.0, 0.2, 0.05)


# numpy view of vpl_line (56 bytes) for bulk packing / unpacking without per-line Python loops
LINE_DTYPE = np.dtype({"names": ["line_endpoint", "line_equation", "center", "length"],
                       "formats": [("<f4", 4), ("<f8", 3), ("<f4", 2), "<f4"],
                       "offsets": [0, 16, 40, 48], "itemsize": 56})
assert LINE_DTYPE.itemsize == C.sizeof(Line)


def lines_to_records(a, out):
    """[n,10] doubles (x1,y1,x2,y2,eq0..2,cx,cy,len) -> out[:n] (LINE_DTYPE records)"""
    n = len(a)
    if n:
        a = np.asarray(a, np.float64)
        out["line_endpoint"][:n] = a[:, 0:4]
        out["line_equation"][:n] = a[:, 4:7]
        out["center"][:n] = a[:, 7:9]
        out["length"][:n] = a[:, 9]


def records_to_lines(rec):
    out = np.empty((len(rec), 10))
    out[:, 0:4] = rec["line_endpoint"]
    out[:, 4:7] = rec["line_equation"]
    out[:, 7:9] = rec["center"]
    out[:, 9] = rec["length"]
    return out


BLUR_NORMALISED, BLUR_OPENCV_341 = 0, 1
LINE_FILTER_PARALLEL_DEFAULT = 0.0348994967   # sin(3 degrees), line_matching.h:37


def default_param():
    """production values: line_feature_tracker_node.cpp:203 / config/euroc/euroc_config.yaml:84-87"""
    p = EdlineParam()
    p.ksize, p.sigma, p.gradientThreshold, p.anchorThreshold = 5, 1.0, 30.0, 5.0
    p.scanIntervals, p.minLineLen, p.lineFitErrThreshold = 2, 35, 1.8
    return p


_bound = False


def _bind(lib):
    global _bound
    if _bound:
        return
    vp = C.c_void_p
    lib.vpl_edline_default_param.argtypes = [C.POINTER(EdlineParam)]
    lib.vpl_fe_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.vpl_fe_destroy.argtypes = [vp]
    lib.vpl_fe_destroy.restype = None
    lib.vpl_fe_set_stream.argtypes = [vp, vp]
    lib.vpl_fe_synchronize.argtypes = [vp]
    lib.vpl_fe_last_error.argtypes = [vp]
    lib.vpl_fe_last_error.restype = C.c_char_p
    lib.vpl_edlines_upload.argtypes = [vp, C.c_int, C.POINTER(C.c_uint8)]
    lib.vpl_vp_detect_batch.argtypes = [vp, C.c_int, C.POINTER(Line), C.POINTER(C.c_int), C.POINTER(Line), C.POINTER(C.c_int),
                                        C.c_float, C.c_float, C.c_float, C.POINTER(C.c_uint32), C.POINTER(C.c_int),
                                        C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.vpl_vp_debug.argtypes = [vp, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.vpl_pre_set_maps.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.vpl_pre_upload.argtypes = [vp, C.c_int, C.POINTER(C.c_uint8)]
    lib.vpl_pre_run.argtypes = [vp, C.c_int, C.c_double, C.c_int, C.c_int]
    lib.vpl_pre_download.argtypes = [vp, C.c_int, C.POINTER(C.c_uint8)]
    lib.vpl_pre_batch.argtypes = [vp, C.c_int, C.POINTER(C.c_uint8), C.c_int, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_uint8)]
    lib.vpl_edlines_detect.argtypes = [vp, C.POINTER(EdlineParam)]
    lib.vpl_edlines_detect_ex.argtypes = [vp, C.POINTER(EdlineParam), C.c_int]
    lib.vpl_edlines_detect_batch_ex.argtypes = [vp, C.c_int, C.POINTER(C.c_uint8), C.POINTER(EdlineParam), C.c_int,
                                                C.POINTER(Line), C.POINTER(C.c_int)]
    lib.vpl_fe_set_blur_kernel.argtypes = [vp, C.c_int]
    lib.vpl_fe_keep_blurred.argtypes = [vp, C.c_int]
    lib.vpl_edlines_debug_blurred.argtypes = [vp, C.c_int, C.POINTER(C.c_uint8)]
    lib.vpl_line_filter_detected.argtypes = [vp, C.c_float, C.c_float]
    lib.vpl_line_filter_batch.argtypes = [vp, C.c_int, C.POINTER(Line), C.POINTER(C.c_int), C.c_float, C.c_float]
    lib.vpl_edlines_download.argtypes = [vp, C.c_int, C.POINTER(Line), C.POINTER(C.c_int)]
    lib.vpl_edlines_detect_batch.argtypes = [vp, C.c_int, C.POINTER(C.c_uint8), C.POINTER(EdlineParam), C.POINTER(Line),
                                             C.POINTER(C.c_int)]
    lib.vpl_edlines_debug_stage.argtypes = [vp, C.c_int] + [C.c_void_p] * 5 + [C.POINTER(C.c_int)] + [C.c_void_p] * 3 + \
                                           [C.POINTER(C.c_int)]
    ip = C.POINTER(C.c_int)
    lib.vpl_edlines_debug_route_stats.argtypes = [vp, C.c_int, C.POINTER(C.c_ulonglong)]
    lib.vpl_match_default_param.argtypes = [C.POINTER(MatchParam)]
    lib.vpl_match_reserve.argtypes = [vp, C.c_int, C.c_int]
    lib.vpl_match_upload.argtypes = [vp, C.c_int, ip, ip, C.POINTER(Line), ip, C.POINTER(Line), ip]
    lib.vpl_match_run.argtypes = [vp, C.POINTER(MatchParam)]
    lib.vpl_match_from_detected.argtypes = [vp, C.c_int, ip, ip, C.c_int]
    lib.vpl_match_counts.argtypes = [vp, C.c_int, ip, ip]
    lib.vpl_match_download.argtypes = [vp, C.c_int, ip, ip]
    lib.vpl_line_match_batch.argtypes = [vp, C.c_int, C.POINTER(C.c_uint8), C.c_int, ip, ip, C.POINTER(Line), ip,
                                         C.POINTER(Line), ip, C.POINTER(MatchParam), ip, ip]
    lib.vpl_match_debug_kps.argtypes = [vp, C.c_int, C.c_int] + [C.c_void_p] * 5 + [ip]
    lib.vpl_match_debug_level.argtypes = [vp, C.c_int, C.c_int, C.c_void_p, C.c_void_p, ip, ip]
    lib.vpl_fe_enable_kernel_timing.argtypes = [vp, C.c_int]
    lib.vpl_fe_kernel_times.argtypes = [vp, ip, C.POINTER(C.c_char_p), C.POINTER(C.c_double)]
    _bound = True


class FrontendContext:
    def __init__(self, device=0, max_images=1, width=752, height=480, max_lines=1024, stream=None):
        self.lib = load_hip_library()
        _bind(self.lib)
        self.h = C.c_void_p()
        self.W, self.H, self.max_lines, self.max_images = width, height, max_lines, max_images
        rc = self.lib.vpl_fe_create(C.byref(self.h), device, max_images, width, height, max_lines)
        if rc != 0:
            raise RuntimeError("vpl_fe_create failed: %d (no HIP device? there is no CPU fallback)" % rc)
        if stream is not None:
            self._check(self.lib.vpl_fe_set_stream(self.h, C.c_void_p(stream)), "vpl_fe_set_stream")
        self.n = 0

    def debug_guards(self):
        """VPL_DEBUG_GUARDS=1 (set before the context is made): number of device arrays with a write behind their end"""
        return int(self.lib.vpl_fe_debug_guards(self.h))

    def close(self):
        if self.h:
            bad = self.debug_guards() if os.environ.get("VPL_DEBUG_GUARDS") == "1" else 0
            msg = self.lib.vpl_fe_last_error(self.h).decode() if bad else ""
            self.lib.vpl_fe_destroy(self.h)
            self.h = C.c_void_p()
            if bad:
                import sys
                print("vpl_fe_debug_guards: %d arrays overrun; %s" % (bad, msg), file=sys.stderr)    # (also when closed by __del__)
                raise RuntimeError("vpl_fe_debug_guards: %d arrays overrun; %s" % (bad, msg))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, self.lib.vpl_fe_last_error(self.h).decode()))

    def upload(self, images):
        images = np.ascontiguousarray(images, np.uint8)
        assert images.ndim == 3 and images.shape[1:] == (self.H, self.W)
        self.n = images.shape[0]
        self._img = images
        self._check(self.lib.vpl_edlines_upload(self.h, self.n, images.ctypes.data_as(C.POINTER(C.c_uint8))), "upload")

    # ---- image preparation (remap + CLAHE), line_feature_tracker.cpp:62-68 ----
    def set_maps(self, map_x, map_y):
        """float32 [H][W] undistortion maps; None, None switches the remap off"""
        fp = C.POINTER(C.c_float)
        if map_x is None:
            self._check(self.lib.vpl_pre_set_maps(self.h, None, None), "vpl_pre_set_maps")
            return
        mx, my = np.ascontiguousarray(map_x, np.float32), np.ascontiguousarray(map_y, np.float32)
        assert mx.shape == (self.H, self.W) and my.shape == (self.H, self.W)
        self._check(self.lib.vpl_pre_set_maps(self.h, mx.ctypes.data_as(fp), my.ctypes.data_as(fp)), "vpl_pre_set_maps")

    def pre_upload(self, raw):
        raw = np.ascontiguousarray(raw, np.uint8)
        assert raw.ndim == 3 and raw.shape[1:] == (self.H, self.W)
        self.n = raw.shape[0]
        self._img = raw
        self._check(self.lib.vpl_pre_upload(self.h, self.n, raw.ctypes.data_as(C.POINTER(C.c_uint8))), "vpl_pre_upload")

    def pre_run(self, equalize=True, clip_limit=3.0, tiles=(8, 8)):
        self._check(self.lib.vpl_pre_run(self.h, int(equalize), clip_limit, tiles[0], tiles[1]), "vpl_pre_run")

    def pre_download(self):
        out = np.empty((self.n, self.H, self.W), np.uint8)
        self._check(self.lib.vpl_pre_download(self.h, self.n, out.ctypes.data_as(C.POINTER(C.c_uint8))), "vpl_pre_download")
        return out

    # ---- vanishing points (vanishing_point_detection.cpp) ----
    def vp_detect(self, hyp_lines, all_lines, f, cx, cy, seeds, first_frame):
        """hyp_lines / all_lines: per frame an [k][>=4] array whose first 4 columns are the end points.
        Returns vps [n][3][3], ids (list of int arrays), status [n]."""
        n = len(hyp_lines)
        ML = self.max_lines

        def pack(lists):
            rec = np.zeros(n * ML, LINE_DTYPE)
            cnt = np.zeros(n, np.int32)
            for i, l in enumerate(lists):
                l = np.asarray(l, np.float64).reshape(-1, np.asarray(l).shape[-1] if len(l) else 4)
                cnt[i] = len(l)
                rec["line_endpoint"][i * ML:i * ML + len(l)] = l[:, :4]
            return rec, cnt
        rh, nh = pack(hyp_lines)
        ra, na = pack(all_lines)
        seeds = np.ascontiguousarray(seeds, np.uint32)
        first = np.ascontiguousarray(first_frame, np.int32)
        vps = np.zeros((n, 3, 3))
        ids = np.zeros((n, ML), np.int32)
        status = np.zeros(n, np.int32)
        ip = C.POINTER(C.c_int)
        self._check(self.lib.vpl_vp_detect_batch(self.h, n, rh.ctypes.data_as(C.POINTER(Line)), nh.ctypes.data_as(ip),
                                                 ra.ctypes.data_as(C.POINTER(Line)), na.ctypes.data_as(ip), f, cx, cy,
                                                 seeds.ctypes.data_as(C.POINTER(C.c_uint32)), first.ctypes.data_as(ip),
                                                 vps.ctypes.data_as(C.POINTER(C.c_double)), ids.ctypes.data_as(ip),
                                                 status.ctypes.data_as(ip)), "vpl_vp_detect_batch")
        return vps, [ids[i, :na[i]].copy() for i in range(n)], status

    def vp_debug(self, frame):
        grid = np.zeros((90, 360))
        pairs = np.zeros((105, 2), np.int32)
        best, drawn = C.c_int(), C.c_int()
        self._check(self.lib.vpl_vp_debug(self.h, frame, grid.ctypes.data, pairs.ctypes.data, C.byref(best), C.byref(drawn)),
                    "vpl_vp_debug")
        return grid, pairs, best.value, drawn.value

    def detect(self, param=None, smoothed=True):
        """EDline(image, lines, smoothed) for the uploaded batch; smoothed=False (the reference's default) runs the Gaussian
        pre-blur of EdgeDrawing first (edline_detector.cpp:82-84)"""
        self._param = param or default_param()
        self._check(self.lib.vpl_edlines_detect_ex(self.h, C.byref(self._param), 1 if smoothed else 0), "vpl_edlines_detect_ex")

    def set_blur_kernel(self, mode):
        """BLUR_NORMALISED (default, OpenCV 3.4.9+ / 4.2+) or BLUR_OPENCV_341"""
        self._check(self.lib.vpl_fe_set_blur_kernel(self.h, int(mode)), "vpl_fe_set_blur_kernel")

    def keep_blurred(self, on=True):
        self._check(self.lib.vpl_fe_keep_blurred(self.h, 1 if on else 0), "vpl_fe_keep_blurred")

    def debug_blurred(self, img):
        out = np.empty((self.H, self.W), np.uint8)
        self._check(self.lib.vpl_edlines_debug_blurred(self.h, img, out.ctypes.data_as(C.POINTER(C.c_uint8))),
                    "vpl_edlines_debug_blurred")
        return out

    def line_filter_detected(self, distance_threshold, parallel_threshold=LINE_FILTER_PARALLEL_DEFAULT):
        """LineMatching::LineFilter (line_matching.cpp:167-264) on the lines of the last detect, in HBM"""
        self._check(self.lib.vpl_line_filter_detected(self.h, distance_threshold, parallel_threshold), "vpl_line_filter_detected")

    def line_filter_batch(self, lines, distance_threshold, parallel_threshold=LINE_FILTER_PARALLEL_DEFAULT):
        """LineFilter on caller-owned lists: lines = list of [k,10] arrays; returns the filtered lists"""
        n, ML = len(lines), self.max_lines
        rec = np.zeros(n * ML, LINE_DTYPE)
        cnt = np.zeros(n, np.int32)
        for i, l in enumerate(lines):
            cnt[i] = len(l)
            lines_to_records(l, rec[i * ML:i * ML + len(l)])
        self._check(self.lib.vpl_line_filter_batch(self.h, n, rec.ctypes.data_as(C.POINTER(Line)),
                                                   cnt.ctypes.data_as(C.POINTER(C.c_int)), distance_threshold,
                                                   parallel_threshold), "vpl_line_filter_batch")
        return [records_to_lines(rec[i * ML:i * ML + cnt[i]]) for i in range(n)]

    def enable_kernel_timing(self, on=True):
        self._check(self.lib.vpl_fe_enable_kernel_timing(self.h, 1 if on else 0), "vpl_fe_enable_kernel_timing")

    def kernel_times(self):
        """{kernel: total ms} of the launches since timing was enabled (hipEvents on the context's stream)"""
        cnt = C.c_int(256)
        names = (C.c_char_p * 256)()
        ms = (C.c_double * 256)()
        self._check(self.lib.vpl_fe_kernel_times(self.h, C.byref(cnt), names, ms), "vpl_fe_kernel_times")
        out = {}
        for i in range(cnt.value):
            out[names[i].decode()] = out.get(names[i].decode(), 0.0) + ms[i]
        return out

    def synchronize(self):
        self._check(self.lib.vpl_fe_synchronize(self.h), "vpl_fe_synchronize")

    def download(self):
        rec = np.zeros(self.n * self.max_lines, LINE_DTYPE)
        counts = (C.c_int * self.n)()
        self._check(self.lib.vpl_edlines_download(self.h, self.n, rec.ctypes.data_as(C.POINTER(Line)), counts),
                    "vpl_edlines_download")
        return [records_to_lines(rec[i * self.max_lines:i * self.max_lines + counts[i]]) for i in range(self.n)]

    def detect_batch(self, images, param=None, smoothed=True):
        self.upload(images)
        self.detect(param, smoothed)
        self.synchronize()
        return self.download()

    def debug_stage(self, img):
        W, H = self.W, self.H
        N = W * H
        cap = N // 5
        dx = np.zeros(N, np.int16); dy = np.zeros(N, np.int16); g = np.zeros(N, np.int16); d = np.zeros(N, np.uint8)
        anchors = np.zeros((cap, 2), np.uint32); nA = C.c_int(0)
        cx = np.zeros(2 * cap, np.uint32); cy = np.zeros(2 * cap, np.uint32); sid = np.zeros(cap // 20 + 2, np.uint32)
        nE = C.c_int(0)
        p = lambda a: C.c_void_p(a.ctypes.data)
        self._check(self.lib.vpl_edlines_debug_stage(self.h, img, p(dx), p(dy), p(g), p(d), p(anchors), C.byref(nA), p(cx),
                                                     p(cy), p(sid), C.byref(nE)), "vpl_edlines_debug_stage")
        ne = nE.value
        npx = int(sid[ne])
        return dict(dx=dx.reshape(H, W), dy=dy.reshape(H, W), g=g.reshape(H, W), dir=d.reshape(H, W),
                    anchors=anchors[:nA.value], chain_x=cx[:npx], chain_y=cy[:npx], sid=sid[:ne + 1])

    def route_stats(self, img):
        out = (C.c_ulonglong * 4)()
        self._check(self.lib.vpl_edlines_debug_route_stats(self.h, img, out), "vpl_edlines_debug_route_stats")
        return dict(steps=out[0], tile_loads=out[1], walks=out[2], cycles=out[3])

    # ---- KLT line matching ---------------------------------------------------------------------------------
    def match_reserve(self, max_pairs, max_kps=4096):
        self.max_pairs, self.max_kps = max_pairs, max_kps
        self._check(self.lib.vpl_match_reserve(self.h, max_pairs, max_kps), "vpl_match_reserve")

    def _pack_lines(self, per_pair):
        rec = np.zeros(len(per_pair) * self.max_lines, LINE_DTYPE)
        cnt = (C.c_int * len(per_pair))()
        for i, L in enumerate(per_pair):
            cnt[i] = len(L)
            if len(L) <= self.max_lines:   # otherwise the library reports VPL_E_CAPACITY
                lines_to_records(L, rec[i * self.max_lines:(i + 1) * self.max_lines])
        return rec, cnt

    def match_upload(self, pairs, lines_ref, lines_cur):
        """pairs: list of (ref image index, cur image index); lines_*: per pair [n,10] arrays"""
        self.n_pairs = len(pairs)
        ri = (C.c_int * self.n_pairs)(*[p[0] for p in pairs])
        ci = (C.c_int * self.n_pairs)(*[p[1] for p in pairs])
        lr, nr = self._pack_lines(lines_ref)
        lc, nc = self._pack_lines(lines_cur)
        self._nref = [len(L) for L in lines_ref]
        self._check(self.lib.vpl_match_upload(self.h, self.n_pairs, ri, ci, lr.ctypes.data_as(C.POINTER(Line)), nr,
                                              lc.ctypes.data_as(C.POINTER(Line)), nc), "vpl_match_upload")

    def match_from_detected(self, pairs, max_lines=None):
        """vpl_match_from_detected: the lines of the last detect() as the matcher's input, device to device"""
        self.n_pairs = len(pairs)
        ri = (C.c_int * self.n_pairs)(*[p[0] for p in pairs])
        ci = (C.c_int * self.n_pairs)(*[p[1] for p in pairs])
        self._check(self.lib.vpl_match_from_detected(self.h, self.n_pairs, ri, ci, int(max_lines or self.max_lines)),
                    "vpl_match_from_detected")
        self._nref = None

    def match_counts(self):
        nr, nc = (C.c_int * self.n_pairs)(), (C.c_int * self.n_pairs)()
        self._check(self.lib.vpl_match_counts(self.h, self.n_pairs, nr, nc), "vpl_match_counts")
        return [int(x) for x in nr], [int(x) for x in nc]

    def match_run(self, param=None):
        self._mparam = param or default_match_param()
        self._check(self.lib.vpl_match_run(self.h, C.byref(self._mparam)), "vpl_match_run")

    def match_download(self):
        a = np.full((self.n_pairs, self.max_lines), -2, np.int32)
        ok = (C.c_int * self.n_pairs)()
        self._check(self.lib.vpl_match_download(self.h, self.n_pairs, a.ctypes.data_as(C.POINTER(C.c_int)), ok),
                    "vpl_match_download")
        if self._nref is None:
            self._nref = self.match_counts()[0]
        return [a[i, :self._nref[i]].copy() for i in range(self.n_pairs)], [int(x) for x in ok]

    def match_batch(self, images, pairs, lines_ref, lines_cur, param=None):
        self.upload(images)
        self.match_upload(pairs, lines_ref, lines_cur)
        self.match_run(param)
        self.synchronize()
        return self.match_download()

    def match_debug_kps(self, pair):
        cap = self.max_kps
        kr = np.zeros((cap, 2), np.float32); kc = np.zeros((cap, 2), np.float32)
        st = np.zeros(cap, np.uint8); er = np.zeros(cap, np.float32); k2l = np.zeros(cap, np.int32)
        nk = C.c_int(0)
        p = lambda a: C.c_void_p(a.ctypes.data)
        self._check(self.lib.vpl_match_debug_kps(self.h, pair, cap, p(kr), p(kc), p(st), p(er), p(k2l), C.byref(nk)),
                    "vpl_match_debug_kps")
        n = nk.value
        return dict(kps_ref=kr[:n], kps_cur=kc[:n], status=st[:n], err=er[:n], kp2line_cur=k2l[:n])

    def match_debug_level(self, img, level):
        w, h = C.c_int(0), C.c_int(0)
        self._check(self.lib.vpl_match_debug_level(self.h, img, level, None, None, C.byref(w), C.byref(h)),
                    "vpl_match_debug_level")
        px = np.zeros((h.value, w.value), np.uint8); d = np.zeros((h.value, w.value, 2), np.int16)
        self._check(self.lib.vpl_match_debug_level(self.h, img, level, C.c_void_p(px.ctypes.data),
                                                   C.c_void_p(d.ctypes.data), C.byref(w), C.byref(h)),
                    "vpl_match_debug_level")
        return px, d
