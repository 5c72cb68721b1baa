"""ctypes front-end of the CPU oracle (oracle/libsg_oracle.so).  TEST INFRASTRUCTURE ONLY.

May be imported from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never from
the product package.  Builds the library with gcc on first use if it is missing.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_DIR, "libsg_oracle.so")

POLYNOMIAL, REFLECT, PERIODIC, CONSTANT = 0, 1, 2, 3
B2D_VALID, B2D_CONSTANT, B2D_REFLECT = 0, 1, 2
SEED = 0x5A17601A


def build(force=False):
    src = [os.path.join(_DIR, f) for f in ("sg_oracle.c", "sg_oracle.h")]
    stale = (not os.path.exists(_LIB)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _DIR, "libsg_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


class _Stream(C.Structure):
    _fields_ = [("ring", C.c_float * 65), ("wp", C.c_int), ("received", C.c_uint64), ("emitted", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        fp, dp, vp, sz = C.POINTER(C.c_float), C.POINTER(C.c_double), C.c_void_p, C.c_size_t
        L.sgo_weights.argtypes = [C.c_int] * 3 + [fp, fp]
        L.sgo_dt_scale.restype = C.c_float; L.sgo_dt_scale.argtypes = [C.c_float, C.c_int]
        L.sgo_dt_inv.restype = C.c_float; L.sgo_dt_inv.argtypes = [C.c_float, C.c_int]
        L.sgo_apply_f32.argtypes = [fp, fp, C.c_int, C.c_float, C.c_int, fp, fp, sz]
        L.sgo_apply_valid_f32.restype = sz
        L.sgo_apply_valid_f32.argtypes = [fp, C.c_int, C.c_float, fp, sz, fp]
        L.sgo_apply_strided_f32.argtypes = [fp, fp, C.c_int, C.c_float, vp, sz, sz, vp, sz, sz, sz]
        L.sgo_apply_batch_f32.argtypes = [fp, fp, C.c_int, C.c_float, C.c_int, fp, fp, sz, sz, sz, C.c_int]
        L.sgo_apply_f64.argtypes = [fp, fp, C.c_int, C.c_float, C.c_int, dp, dp, sz]
        L.sgo_apply_batch_f64.argtypes = [fp, fp, C.c_int, C.c_float, C.c_int, dp, dp, sz, sz, sz, C.c_int]
        L.sgo_stream_reset.argtypes = [C.POINTER(_Stream)]
        L.sgo_stream_push.argtypes = [C.POINTER(_Stream), fp, C.c_int, C.c_float, C.c_float, fp]
        L.sgo_stream_push_full.argtypes = [C.POINTER(_Stream), fp, fp, C.c_int, C.c_float, C.c_float, fp, C.c_int]
        L.sgo_stream_flush.argtypes = [C.POINTER(_Stream), fp, C.c_int, C.c_float, fp, C.c_int]
        L.sgo_stream_flush_leading.argtypes = [C.POINTER(_Stream), fp, C.c_int, C.c_float, fp, C.c_int]
        L.sgo2d_weights.argtypes = [C.c_int] * 5 + [fp]
        L.sgo2d_scale.restype = C.c_float; L.sgo2d_scale.argtypes = [C.c_float, C.c_float, C.c_int, C.c_int]
        L.sgo2d_apply_valid_f32.argtypes = [fp, C.c_int, C.c_int, C.c_float, fp, C.c_int, C.c_int, C.c_int, fp, C.c_int]
        L.sgo2d_apply_f32.argtypes = [fp, C.c_int, C.c_int, C.c_float, fp, C.c_int, C.c_int, C.c_int, fp, C.c_int, C.c_int]
        L.sgo2d_apply_f64acc.argtypes = [fp, C.c_int, C.c_int, C.c_float, fp, C.c_int, C.c_int, C.c_int, dp, C.c_int, C.c_int]
        L.sgo_synth_f32.argtypes = [fp, sz, sz, sz, sz, C.c_uint64]
        L.sgo_synth_f64.argtypes = [dp, sz, sz, sz, sz, C.c_uint64]
        _lib = L
    return _lib


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def weights(n, m, d):
    """(center[2n+1], edges[n][2n+1]) fp32, or None when the config is invalid."""
    ws = 2 * n + 1
    cw = np.zeros(max(ws, 1), np.float32)
    ew = np.zeros((max(n, 1), max(ws, 1)), np.float32)
    if lib().sgo_weights(n, m, d, _f(cw), _f(ew)) != 0:
        return None
    return cw, ew


def dt_scale(time_step, d):
    return np.float32(lib().sgo_dt_scale(float(np.float32(time_step)), d))


def dt_inv(time_step, d):
    return np.float32(lib().sgo_dt_inv(float(np.float32(time_step)), d))


class Filter:
    """Oracle-side filter: the fp32 tables + the constants the apply loops need."""

    def __init__(self, n, m, d=0, time_step=1.0, mode=POLYNOMIAL):
        w = weights(n, m, d)
        if w is None or not (np.float32(time_step) > 0):
            raise ValueError("invalid Savitzky-Golay configuration")
        self.n, self.m, self.d, self.mode = n, m, d, mode
        self.ws = 2 * n + 1
        self.center, self.edges = w
        self.dt_scale = dt_scale(time_step, d)
        self.dt_inv = dt_inv(time_step, d)

    # ---- fp32, reference order ----
    def apply(self, x, mode=None):
        x = np.ascontiguousarray(x, np.float32)
        mode = self.mode if mode is None else mode
        if x.ndim == 1:
            y = np.empty_like(x)
            rc = lib().sgo_apply_f32(_f(self.center), _f(self.edges), self.n, float(self.dt_inv), mode, _f(x), _f(y), x.size)
        else:
            y = np.empty_like(x)
            rc = lib().sgo_apply_batch_f32(_f(self.center), _f(self.edges), self.n, float(self.dt_inv), mode,
                                           _f(x), _f(y), x.shape[0], x.shape[1], x.shape[1], _threads(x.shape[0]))
        if rc != 0:
            raise ValueError("oracle apply failed (length < window?)")
        return y

    def apply_valid(self, x):
        x = np.ascontiguousarray(x, np.float32)
        y = np.empty(max(x.size - 2 * self.n, 0), np.float32)
        got = lib().sgo_apply_valid_f32(_f(self.center), self.n, float(self.dt_inv), _f(x), x.size, _f(y))
        return y[:got]

    def apply_strided(self, src, in_stride, in_offset, dst, out_stride, out_offset, count):
        return lib().sgo_apply_strided_f32(_f(self.center), _f(self.edges), self.n, float(self.dt_inv),
                                           src.ctypes.data, in_stride, in_offset,
                                           dst.ctypes.data, out_stride, out_offset, count)

    # ---- fp64 oracle (fp32 tables promoted, double accumulation) ----
    def apply_f64(self, x, mode=None, threads=None):
        x = np.ascontiguousarray(x, np.float64)
        mode = self.mode if mode is None else mode
        y = np.empty_like(x)
        if x.ndim == 1:
            rc = lib().sgo_apply_f64(_f(self.center), _f(self.edges), self.n, float(self.dt_inv), mode, _d(x), _d(y), x.size)
        else:
            rc = lib().sgo_apply_batch_f64(_f(self.center), _f(self.edges), self.n, float(self.dt_inv), mode,
                                           _d(x), _d(y), x.shape[0], x.shape[1], x.shape[1],
                                           threads or _threads(x.shape[0]))
        if rc != 0:
            raise ValueError("oracle apply failed (length < window?)")
        return y


def _threads(channels):
    """one OpenMP thread per channel at most: a 256-thread team for a 5-channel batch costs ~120 ms per call on the GPU box's host"""
    return max(1, min(os.cpu_count() or 1, int(channels)))


class Stream:
    def __init__(self, filt):
        self.f = filt
        self.s = _Stream()
        lib().sgo_stream_reset(C.byref(self.s))

    def push(self, x):
        y = C.c_float(0.0)
        ok = lib().sgo_stream_push(C.byref(self.s), _f(self.f.center), self.f.n, float(self.f.dt_inv), float(x), C.byref(y))
        return (np.float32(y.value) if ok else np.float32(0.0)), bool(ok)

    def push_full(self, x, max_out=33):
        buf = np.zeros(max(max_out, 1), np.float32)
        c = lib().sgo_stream_push_full(C.byref(self.s), _f(self.f.center), _f(self.f.edges), self.f.n,
                                       float(self.f.dt_inv), float(x), _f(buf), max_out)
        return buf[:c].copy()

    def flush(self, max_out=32):
        buf = np.zeros(max(max_out, 1), np.float32)
        c = lib().sgo_stream_flush(C.byref(self.s), _f(self.f.edges), self.f.n, float(self.f.dt_inv), _f(buf), max_out)
        return c, buf[:max(c, 0)].copy()

    def flush_leading(self, max_out=32):
        buf = np.zeros(max(max_out, 1), np.float32)
        c = lib().sgo_stream_flush_leading(C.byref(self.s), _f(self.f.edges), self.f.n, float(self.f.dt_inv), _f(buf), max_out)
        return c, buf[:max(c, 0)].copy()

    @property
    def counters(self):
        return int(self.s.received), int(self.s.emitted), int(self.s.wp)


class Filter2D:
    def __init__(self, nx, ny, order, dx=0, dy=0, delta_x=1.0, delta_y=1.0):
        self.nx, self.ny, self.order, self.dx, self.dy = nx, ny, order, dx, dy
        W = np.zeros((2 * ny + 1, 2 * nx + 1), np.float32)
        if lib().sgo2d_weights(nx, ny, order, dx, dy, _f(W)) != 0 or not (delta_x > 0 and delta_y > 0):
            raise ValueError("invalid 2-D Savitzky-Golay configuration")
        self.W = W
        self.scale = np.float32(lib().sgo2d_scale(float(np.float32(delta_x)), float(np.float32(delta_y)), dx, dy))

    def apply(self, img, cols=None, boundary=B2D_VALID, out=None):
        """img: 2-D fp32 array whose row pitch is img.shape[1] (cols <= pitch)."""
        img = np.ascontiguousarray(img, np.float32)
        rows, stride = img.shape
        cols = stride if cols is None else cols
        o = np.array(out, np.float32, copy=True) if out is not None else np.zeros_like(img)
        rc = lib().sgo2d_apply_f32(_f(self.W), self.nx, self.ny, float(self.scale), _f(img), rows, cols, stride,
                                   _f(o), stride, boundary)
        if rc != 0:
            raise ValueError("oracle 2-D apply failed")
        return o

    def apply_f64acc(self, img, cols=None, boundary=B2D_VALID):
        img = np.ascontiguousarray(img, np.float32)
        rows, stride = img.shape
        cols = stride if cols is None else cols
        o = np.zeros(img.shape, np.float64)
        rc = lib().sgo2d_apply_f64acc(_f(self.W), self.nx, self.ny, float(self.scale), _f(img), rows, cols, stride,
                                      _d(o), stride, boundary)
        if rc != 0:
            raise ValueError("oracle 2-D apply failed")
        return o


def synth_f32(channel0, channels, length, seed=SEED):
    a = np.empty((channels, length), np.float32)
    lib().sgo_synth_f32(_f(a), channel0, channels, length, length, seed)
    return a


def synth_f64(channel0, channels, length, seed=SEED):
    a = np.empty((channels, length), np.float64)
    lib().sgo_synth_f64(_d(a), channel0, channels, length, length, seed)
    return a
