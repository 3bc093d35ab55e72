"""ctypes wrapper of oracle/libsvgf_oracle.so — the CPU parity oracle.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The product
(cuda-path-tracer-denoising_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsvgf_oracle.so")

VARIANCE_SNAPSHOT, VARIANCE_INPLACE = 0, 1
_lib = None


def load(camera_cls, params_cls):
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} missing: run __graft_entry__.build()")
    lib = C.CDLL(LIB_PATH)
    vp, ip, fp = C.c_void_p, C.c_int, C.c_float
    lib.svgf_oracle_create.argtypes = [ip, ip]
    lib.svgf_oracle_create.restype = vp
    lib.svgf_oracle_destroy.argtypes = [vp]
    lib.svgf_oracle_reset.argtypes = [vp]
    lib.svgf_oracle_set_threads.argtypes = [vp, ip]
    lib.svgf_oracle_set_variance_mode.argtypes = [vp, ip]
    lib.svgf_oracle_denoise.argtypes = [vp, vp, vp, vp, C.POINTER(camera_cls), C.POINTER(params_cls)]
    lib.svgf_oracle_read_state.argtypes = [vp, ip, vp, C.c_ulonglong]
    lib.svgf_oracle_atrous.argtypes = [vp, vp, vp, vp, vp, ip, ip, ip, ip, fp, fp, fp, ip, ip, ip, ip]
    lib.svgf_oracle_backproject.argtypes = [vp] * 11 + [ip, ip, fp, fp, ip]
    lib.svgf_oracle_view_matrix.argtypes = [C.POINTER(camera_cls), vp]
    _lib = lib
    return lib


class Oracle:
    """CPU restatement of denoiseInit / denoise / denoiseFree (reference src/denoise.cu)."""

    def __init__(self, pkg, width, height, threads=1, variance_mode=VARIANCE_SNAPSHOT):
        self.pkg = pkg
        self.lib = load(pkg.SvgfCamera, pkg.SvgfParams)
        self.W, self.H = int(width), int(height)
        self.h = self.lib.svgf_oracle_create(self.W, self.H)
        self.lib.svgf_oracle_set_threads(self.h, int(threads))
        self.lib.svgf_oracle_set_variance_mode(self.h, int(variance_mode))

    def free(self):
        if self.h:
            self.lib.svgf_oracle_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def reset(self):
        self.lib.svgf_oracle_reset(self.h)

    def denoise(self, color, gbuffer, camera, params):
        color = np.ascontiguousarray(color, dtype=np.float32)
        gbuffer = np.ascontiguousarray(gbuffer)
        assert color.size == 3 * self.W * self.H and gbuffer.nbytes == 52 * self.W * self.H
        out = np.empty_like(color)
        cam = camera if isinstance(camera, self.pkg.SvgfCamera) else self.pkg.SvgfCamera.from_dict(camera)
        self.lib.svgf_oracle_denoise(self.h, out.ctypes.data, color.ctypes.data, gbuffer.ctypes.data,
                                     C.byref(cam), C.byref(params))
        return out

    def read_state(self, which):
        shape, dt = {0: ((self.H, self.W), np.int32), 1: ((self.H, self.W, 2), np.float32),
                     2: ((self.H, self.W, 3), np.float32), 3: ((self.H, self.W), np.float32),
                     4: ((self.H, self.W, 3), np.float32)}[which]
        a = np.empty(shape, dtype=dt)
        rc = self.lib.svgf_oracle_read_state(self.h, which, a.ctypes.data, a.nbytes)
        assert rc == 0
        return a


def view_matrix(pkg, camera):
    lib = load(pkg.SvgfCamera, pkg.SvgfParams)
    cam = camera if isinstance(camera, pkg.SvgfCamera) else pkg.SvgfCamera.from_dict(camera)
    m = np.empty(16, dtype=np.float32)
    lib.svgf_oracle_view_matrix(C.byref(cam), m.ctypes.data)
    return m
