"""ctypes binding of include/vplines_frontend.h (EDLines extractor on the GPU)."""
import ctypes as C

import numpy as np

from .capi import load_hip_library


class EdlineParam(C.Structure):
    _fields_ = [("ksize", C.c_int), ("sigma", C.c_float), ("gradientThreshold", C.c_float),
                ("anchorThreshold", C.c_float), ("scanIntervals", C.c_int), ("minLineLen", C.c_int),
                ("lineFitErrThreshold", C.c_double)]


class Line(C.Structure):
    _fields_ = [("line_endpoint", C.c_float * 4), ("line_equation", C.c_double * 3), ("center", C.c_float * 2),
                ("length", C.c_float)]


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
    lib.vpl_edlines_detect.argtypes = [vp, C.POINTER(EdlineParam)]
    lib.vpl_edlines_download.argtypes = [vp, C.c_int, C.POINTER(Line), C.POINTER(C.c_int)]
    lib.vpl_edlines_detect_batch.argtypes = [vp, C.c_int, C.POINTER(C.c_uint8), C.POINTER(EdlineParam), C.POINTER(Line),
                                             C.POINTER(C.c_int)]
    lib.vpl_edlines_debug_stage.argtypes = [vp, C.c_int] + [C.c_void_p] * 5 + [C.POINTER(C.c_int)] + [C.c_void_p] * 3 + \
                                           [C.POINTER(C.c_int)]
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

    def close(self):
        if self.h:
            self.lib.vpl_fe_destroy(self.h)
            self.h = C.c_void_p()

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

    def detect(self, param=None):
        self._param = param or default_param()
        self._check(self.lib.vpl_edlines_detect(self.h, C.byref(self._param)), "vpl_edlines_detect")

    def synchronize(self):
        self._check(self.lib.vpl_fe_synchronize(self.h), "vpl_fe_synchronize")

    def download(self):
        lines = (Line * (self.n * self.max_lines))()
        counts = (C.c_int * self.n)()
        self._check(self.lib.vpl_edlines_download(self.h, self.n, lines, counts), "vpl_edlines_download")
        out = []
        for i in range(self.n):
            m = counts[i]
            arr = np.zeros((m, 10))
            for k in range(m):
                ln = lines[i * self.max_lines + k]
                arr[k, 0:4] = ln.line_endpoint[:]
                arr[k, 4:7] = ln.line_equation[:]
                arr[k, 7:9] = ln.center[:]
                arr[k, 9] = ln.length
            out.append(arr)
        return out

    def detect_batch(self, images, param=None):
        self.upload(images)
        self.detect(param)
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
