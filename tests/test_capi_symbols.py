"""The C-ABI shared library loads and exports every function include/vplines_ba.h declares
and include/vplines_frontend.h declare (no compute calls: this runs without a GPU)."""
import ctypes as C
import os
import re

import vplines_slam_amd as v

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vpl_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = v.load_hip_library()
    names = declared_functions(os.path.join(ROOT, "include", "vplines_ba.h"))
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "missing export: " + n
    fe = declared_functions(os.path.join(ROOT, "include", "vplines_frontend.h"))
    assert len(fe) >= 18
    for n in fe:
        assert hasattr(lib, n), "missing export: " + n


def test_struct_layouts_match_header_sizes():
    # sizes the C header implies (doubles/ints only, natural alignment)
    assert C.sizeof(v.capi.Preintegration) == 8 * (1 + 3 + 4 + 3 + 3 + 3 + 225 + 225)
    assert C.sizeof(v.capi.Prior) == 4 * (2 + 3 * 23) + 4 + 8 * (23 * 9 + 171 * 171 + 171)
    assert C.sizeof(v.capi.BaOptions) == 16 + 8 * 9
    opt = v.capi.BaOptions()
    v.load_hip_library().vpl_ba_default_options(C.byref(opt))
    ref = v.default_options()
    for f, _ in v.capi.BaOptions._fields_:
        assert getattr(opt, f) == getattr(ref, f), f


def test_no_device_reports_error_not_fallback():
    """Without a GPU the context constructor must fail loudly (VPL_E_NODEVICE), never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        return
    lib = v.load_hip_library()
    h = C.c_void_p()
    rc = lib.vpl_ctx_create(C.byref(h), 0, 1, 8, 64, 8, 64)
    assert rc in (-2, -3) and not h.value
