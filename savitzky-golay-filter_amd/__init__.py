"""savgol_amd -- Python mirror (ctypes) of the C ABI of libsavgol_hip.so.

The product is the shared library: host C/C++ that keeps the reference's savgolFilter.h /
savgol_stream.h / savgol2d.h API, and hand-written gfx950 HIP kernels for the hot path.  This module
only binds it: same function names, same argument meaning and error behaviour as the C headers in
../include (which cite the reference lines they replace).  There is no Python or CPU compute path
in here -- if the library is missing, importing `lib()` fails loudly.

Host-pointer calls take numpy arrays; device-pointer calls take anything with `.data_ptr()`
(torch tensors: torch is used for device memory / streams only) or a raw integer address.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SAVGOL_HIP_LIB: another build of the same library (tools/ A/B runs); the product is lib/libsavgol_hip.so
LIB_PATH = os.environ.get("SAVGOL_HIP_LIB") or os.path.join(_HERE, "lib", "libsavgol_hip.so")

SAVGOL_MAX_HALF_WINDOW = 32
SAVGOL_MAX_WINDOW = 65
SAVGOL_MAX_POLY_ORDER = 10
SAVGOL_MAX_DERIVATIVE = 4
SAVGOL_BOUNDARY_POLYNOMIAL, SAVGOL_BOUNDARY_REFLECT, SAVGOL_BOUNDARY_PERIODIC, SAVGOL_BOUNDARY_CONSTANT = 0, 1, 2, 3
SAVGOL2D_BOUNDARY_VALID, SAVGOL2D_BOUNDARY_CONSTANT, SAVGOL2D_BOUNDARY_REFLECT = 0, 1, 2
SAVGOL_HIP_OPT_CORRECT_LEADING_EDGE = 1
SAVGOL_HIP_OPT_REFERENCE_SUMMATION = 2
SAVGOL_HIP_OPT_PLAIN_SUMMATION = 3
SAVGOL_HIP_OPT_BOUNDARY_AWARE = 4
SAVGOL_HIP_OPT_TILE_WIDTH = 5
SAVGOL_STREAMBANK_FMA = 1
SAVGOL_BATCH_REFERENCE_SUMMATION, SAVGOL_BATCH_PLAIN_SUMMATION, SAVGOL_BATCH_TILE_NARROW, SAVGOL_BATCH_TILE_WIDE = 1, 2, 4, 8
SAVGOL_BATCH_CORRECT_LEADING_EDGE, SAVGOL_BATCH_BOUNDARY_AWARE, SAVGOL_BATCH_MOMENT_F64 = 16, 32, 64


class SavgolConfig(C.Structure):
    _fields_ = [("half_window", C.c_uint8), ("poly_order", C.c_uint8), ("derivative", C.c_uint8),
                ("time_step", C.c_float), ("boundary", C.c_int)]


class SavgolFilter(C.Structure):
    _fields_ = [("config", SavgolConfig), ("window_size", C.c_int), ("dt_scale", C.c_float),
                ("center_weights", C.c_float * SAVGOL_MAX_WINDOW),
                ("edge_weights", (C.c_float * SAVGOL_MAX_WINDOW) * SAVGOL_MAX_HALF_WINDOW)]


class SavgolStream(C.Structure):
    _fields_ = [("filter", C.POINTER(SavgolFilter)), ("buffer", C.c_float * SAVGOL_MAX_WINDOW),
                ("write_pos", C.c_int), ("samples_received", C.c_size_t), ("samples_output", C.c_size_t),
                ("owns_filter", C.c_bool), ("dt_inv", C.c_float)]


class Savgol2DConfig(C.Structure):
    _fields_ = [("half_window_x", C.c_uint8), ("half_window_y", C.c_uint8), ("poly_order", C.c_uint8),
                ("deriv_x", C.c_uint8), ("deriv_y", C.c_uint8), ("delta_x", C.c_float), ("delta_y", C.c_float)]


class Savgol2DFilter(C.Structure):
    _fields_ = [("config", Savgol2DConfig), ("window_width", C.c_int), ("window_height", C.c_int),
                ("window_area", C.c_int), ("num_terms", C.c_int), ("scale", C.c_float),
                ("weights", C.POINTER(C.c_float))]


_fp, _dp, _vp, _sz = C.POINTER(C.c_float), C.POINTER(C.c_double), C.c_void_p, C.c_size_t
_F, _F2, _S = C.POINTER(SavgolFilter), C.POINTER(Savgol2DFilter), C.POINTER(SavgolStream)

# name -> (restype, argtypes); everything include/*.h declares
SIGNATURES = {
    # savgolFilter.h
    "savgol_create": (_F, [C.POINTER(SavgolConfig)]),
    "savgol_destroy": (None, [_F]),
    "savgol_apply": (C.c_int, [_F, _fp, _fp, _sz]),
    "savgol_apply_strided": (C.c_int, [_F, _vp, _sz, _sz, _vp, _sz, _sz, _sz]),
    "savgol_apply_valid": (_sz, [_F, _fp, _sz, _fp]),
    # savgol_stream.h
    "savgol_stream_create": (_S, [C.POINTER(SavgolConfig)]),
    "savgol_stream_init": (C.c_int, [_S, _F]),
    "savgol_stream_destroy": (None, [_S]),
    "savgol_stream_reset": (None, [_S]),
    "savgol_stream_push": (C.c_float, [_S, C.c_float, C.POINTER(C.c_bool)]),
    "savgol_stream_push_full": (C.c_int, [_S, C.c_float, _fp, C.c_int]),
    "savgol_stream_flush": (C.c_int, [_S, _fp, C.c_int]),
    "savgol_stream_flush_leading": (C.c_int, [_S, _fp, C.c_int]),
    "savgol_stream_ready": (C.c_bool, [_S]),
    "savgol_stream_latency": (_sz, [_S]),
    "savgol_stream_buffered": (_sz, [_S]),
    "savgol_stream_samples_received": (_sz, [_S]),
    "savgol_stream_samples_output": (_sz, [_S]),
    # savgol2d.h
    "savgol2d_create": (_F2, [C.POINTER(Savgol2DConfig)]),
    "savgol2d_destroy": (None, [_F2]),
    "savgol2d_config_valid": (C.c_bool, [C.POINTER(Savgol2DConfig)]),
    "savgol2d_apply_valid": (C.c_int, [_F2, _fp, C.c_int, C.c_int, C.c_int, _fp, C.c_int]),
    "savgol2d_apply": (C.c_int, [_F2, _fp, C.c_int, C.c_int, C.c_int, _fp, C.c_int, C.c_int]),
    "savgol2d_gradient": (C.c_int, [C.c_int] * 3 + [_fp, C.c_int, C.c_int, C.c_int, _fp, _fp, C.c_float, C.c_float, C.c_int]),
    "savgol2d_hessian": (C.c_int, [C.c_int] * 3 + [_fp, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, C.c_float, C.c_float, C.c_int]),
    "savgol2d_laplacian": (C.c_int, [C.c_int] * 3 + [_fp, C.c_int, C.c_int, C.c_int, _fp, C.c_float, C.c_float, C.c_int]),
    # savgol_hip.h: runtime
    "savgol_hip_device_count": (C.c_int, []),
    "savgol_hip_set_device": (C.c_int, [C.c_int]),
    "savgol_hip_get_device": (C.c_int, []),
    "savgol_hip_synchronize": (C.c_int, [_vp]),
    "savgol_hip_trim_scratch": (C.c_int, []),
    "savgol_hip_scratch_reserved": (C.c_size_t, []),
    "savgol_hip_shard_range": (C.c_int, [_sz, C.c_int, C.c_int, C.POINTER(_sz), C.POINTER(_sz)]),
    "savgol_hip_last_error": (C.c_char_p, []),
    "savgol_hip_version": (C.c_char_p, []),
    "savgol_hip_set_option": (C.c_int, [C.c_int, C.c_int]),
    "savgol_hip_momenth_table": (C.c_int, [_F, _fp]),
    "savgol_hip_stream_moment_table": (C.c_int, [C.c_int, _fp, _fp]),
    "savgol_export_header": (C.c_long, [_F, C.c_char_p, C.c_char_p, C.c_char_p, _sz]),
    # savgol_hip.h: 1-D batch
    "savgol_apply_batch_f32": (C.c_int, [_F, _vp, _vp, _sz, _sz, _sz, _sz, _vp]),
    "savgol_apply_batch_f64": (C.c_int, [_F, _vp, _vp, _sz, _sz, _sz, _sz, _vp]),
    "savgol_apply_valid_batch_f32": (C.c_int, [_F, _vp, _vp, _sz, _sz, _sz, _sz, _vp]),
    "savgol_apply_valid_batch_f64": (C.c_int, [_F, _vp, _vp, _sz, _sz, _sz, _sz, _vp]),
    "savgol_apply_strided_batch_f32": (C.c_int, [_F, _vp, _sz, _sz, _sz, _vp, _sz, _sz, _sz, _sz, _sz, _vp]),
    "savgol_apply_batch_f32_ex": (C.c_int, [_F, _vp, _vp, _sz, _sz, _sz, _sz, C.c_uint, _vp]),
    "savgol_apply_batch_f64_ex": (C.c_int, [_F, _vp, _vp, _sz, _sz, _sz, _sz, C.c_uint, _vp]),
    "savgol_apply_batch_f64_tol": (C.c_int, [_F, _vp, _vp, _sz, _sz, _sz, _sz, C.c_double, _vp]),
    "savgol_apply_valid_batch_f64_tol": (C.c_int, [_F, _vp, _vp, _sz, _sz, _sz, _sz, C.c_double, _vp]),
    "savgol_apply_valid_batch_f32_ex": (C.c_int, [_F, _vp, _vp, _sz, _sz, _sz, _sz, C.c_uint, _vp]),
    "savgol_apply_valid_batch_f64_ex": (C.c_int, [_F, _vp, _vp, _sz, _sz, _sz, _sz, C.c_uint, _vp]),
    "savgol_apply_strided_batch_f32_ex": (C.c_int, [_F, _vp, _sz, _sz, _sz, _vp, _sz, _sz, _sz, _sz, _sz, C.c_uint, _vp]),
    "savgol_hip_default_flags": (C.c_uint, []),
    # savgol_hip.h: stream bank
    "savgol_streambank_create": (_vp, [C.POINTER(SavgolConfig), _sz]),
    "savgol_streambank_create_ex": (_vp, [C.POINTER(SavgolConfig), _sz, C.c_uint]),
    "savgol_streambank_destroy": (None, [_vp]),
    "savgol_streambank_reset": (C.c_int, [_vp, _vp]),
    "savgol_streambank_push": (C.c_int, [_vp, _vp, _vp, _vp]),
    "savgol_streambank_push_wait": (C.c_int, [_vp, _vp, _vp, _vp]),
    "savgol_streambank_push_full": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp]),
    "savgol_streambank_push_block": (C.c_int, [_vp, _vp, _sz, _vp, _vp]),
    "savgol_streambank_flush": (C.c_int, [_vp, _vp, C.c_int, _vp]),
    "savgol_streambank_flush_leading": (C.c_int, [_vp, _vp, C.c_int, _vp]),
    "savgol_streambank_ready": (C.c_bool, [_vp]),
    "savgol_streambank_latency": (_sz, [_vp]),
    "savgol_streambank_streams": (_sz, [_vp]),
    "savgol_streambank_samples_received": (_sz, [_vp]),
    "savgol_streambank_samples_output": (_sz, [_vp]),
    "savgol_streambank_service_start": (C.c_int, [_vp, C.c_uint]),
    "savgol_streambank_service_tick": (C.c_int, [_vp, _vp, _vp]),
    "savgol_streambank_service_stop": (C.c_int, [_vp]),
    "savgol_streambank_service_running": (C.c_int, [_vp]),
    "savgol_streambank_state_bytes": (_sz, [_vp]),
    "savgol_streambank_save": (C.c_int, [_vp, _vp, _vp]),
    "savgol_streambank_load": (C.c_int, [_vp, _vp, _vp]),
    # savgol_hip.h: 2-D batch
    "savgol2d_apply_batch_f32": (C.c_int, [_F2, _vp, C.c_int, C.c_int, C.c_int, _sz, _vp, C.c_int, _sz, _sz, C.c_int, C.c_int, _vp]),
    "savgol2d_gradient_batch_f32": (C.c_int, [C.c_int] * 3 + [_vp, C.c_int, C.c_int, C.c_int, _sz, _vp, _vp, C.c_int, _sz, _sz, C.c_float, C.c_float, C.c_int, _vp]),
    "savgol2d_hessian_batch_f32": (C.c_int, [C.c_int] * 3 + [_vp, C.c_int, C.c_int, C.c_int, _sz, _vp, _vp, _vp, C.c_int, _sz, _sz, C.c_float, C.c_float, C.c_int, _vp]),
    "savgol2d_laplacian_batch_f32": (C.c_int, [C.c_int] * 3 + [_vp, C.c_int, C.c_int, C.c_int, _sz, _vp, C.c_int, _sz, _sz, C.c_float, C.c_float, C.c_int, _vp]),
    # savgol_hip.h: 2-D row bands
    "savgol2d_rowband_plan": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "savgol2d_apply_rowband_f32": (C.c_int, [_F2, _vp, C.c_int, C.c_int, C.c_int, _sz, _vp, _vp, C.c_int, _sz, _vp, C.c_int, _sz, _sz, C.c_int, C.c_int, _vp]),
    "savgol2d_apply_rowband_edges_f32": (C.c_int, [_F2, _vp, C.c_int, C.c_int, C.c_int, _sz, _vp, _vp, C.c_int, _sz, _vp, C.c_int, _sz, _sz, C.c_int, C.c_int, _vp]),
    "savgol2d_apply_rowband_edges_streams_f32": (C.c_int, [_F2, _vp, C.c_int, C.c_int, C.c_int, _sz, _vp, _vp, C.c_int, _sz, _vp, C.c_int, _sz, _sz, C.c_int, C.c_int, _vp, _vp]),
    # savgol_hip.h: bench utilities
    "savgol_hip_stream_copy": (C.c_int, [_vp, _vp, _sz, _vp]),
    "savgol_hip_stream_read": (C.c_int, [_vp, _sz, _vp, _vp]),
    "savgol_hip_synth_f32": (C.c_int, [_vp, _sz, _sz, _sz, _sz, C.c_uint64, _vp]),
    "savgol_hip_synth_f64": (C.c_int, [_vp, _sz, _sz, _sz, _sz, C.c_uint64, _vp]),
}

_lib = None


def lib():
    """The loaded C ABI.  Raises if libsavgol_hip.so was not built (no fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `make -C {_HERE} -j8` "
                              "(or __graft_entry__.build()); there is no CPU fallback")
        try:                # if torch is around, let it load ITS HIP runtime first so both share one copy
            import torch  # noqa: F401
        except Exception:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError here = header/library mismatch
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def last_error():
    return lib().savgol_hip_last_error().decode()


def device_count():
    return lib().savgol_hip_device_count()


def _addr(x):
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    if isinstance(x, np.ndarray):
        return x.ctypes.data
    return int(x)


def _stream(stream):
    """None -> torch's current stream if torch has a GPU, else the default stream."""
    if stream is not None:
        return getattr(stream, "cuda_stream", stream)
    try:
        import torch
        if torch.cuda.is_available():
            return torch.cuda.current_stream().cuda_stream
    except Exception:
        pass
    return None


def _f(a):
    return a.ctypes.data_as(_fp)


def shard_range(total, world_size, rank):
    """Contiguous slice [lo, hi) of `total` independent units (channels, streams, images) owned by `rank`.

    The hot path has no exchange step, so multi-GPU = every rank filters its own slice; the first
    `total % world_size` ranks take one unit more."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad world_size / rank")
    q, r = divmod(total, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


class Filter:
    """RAII wrapper around savgol_create / savgol_destroy with the apply entry points as methods."""

    def __init__(self, half_window, poly_order, derivative=0, time_step=1.0, boundary=SAVGOL_BOUNDARY_POLYNOMIAL):
        cfg = SavgolConfig(half_window, poly_order, derivative, time_step, boundary)
        self.ptr = lib().savgol_create(C.byref(cfg))
        if not self.ptr:
            raise ValueError("savgol_create rejected the configuration")
        self.n = half_window
        self.ws = 2 * half_window + 1

    def close(self):
        if getattr(self, "ptr", None) and _lib is not None:      # (_lib is gone at interpreter shutdown)
            _lib.savgol_destroy(self.ptr)
        self.ptr = None

    __del__ = close

    # tables, as numpy copies
    @property
    def center_weights(self):
        return np.array(self.ptr.contents.center_weights[:self.ws], dtype=np.float32)

    @property
    def edge_weights(self):
        ew = self.ptr.contents.edge_weights
        return np.array([list(ew[e][:self.ws]) for e in range(self.n)], dtype=np.float32).reshape(self.n, self.ws)

    @property
    def dt_scale(self):
        return np.float32(self.ptr.contents.dt_scale)

    # ---- host-pointer drop-in calls (numpy in, numpy out) ----
    def apply(self, x, out=None):
        x = np.ascontiguousarray(x, np.float32)
        y = np.empty_like(x) if out is None else out
        rc = lib().savgol_apply(self.ptr, _f(x), _f(y), x.size)
        if rc != 0:
            raise RuntimeError(f"savgol_apply returned {rc}: {last_error()}")
        return y

    def apply_valid(self, x):
        x = np.ascontiguousarray(x, np.float32)
        y = np.empty(max(x.size - 2 * self.n, 1), np.float32)
        got = lib().savgol_apply_valid(self.ptr, _f(x), x.size, _f(y))
        return y[:got]

    def apply_strided(self, src, in_stride, in_offset, dst, out_stride, out_offset, count):
        return lib().savgol_apply_strided(self.ptr, src.ctypes.data, in_stride, in_offset,
                                          dst.ctypes.data, out_stride, out_offset, count)

    # ---- device-pointer batch calls (torch tensors or raw addresses) ----
    def apply_batch(self, d_in, d_out, channels, length, in_ld=None, out_ld=None, dtype="f32", valid=False, stream=None, flags=None, rel_tol=None):
        """flags=None: the process-wide defaults (savgol_hip_set_option); an int: the *_ex entry point with exactly these SAVGOL_BATCH_* flags;
        rel_tol (fp64 only): savgol_apply[_valid]_batch_f64_tol -- the accuracy the caller accepts picks the kernel."""
        assert rel_tol is None or (dtype == "f64" and flags is None)
        name = f"savgol_apply_{'valid_' if valid else ''}batch_{dtype}" + ("_tol" if rel_tol is not None else ("" if flags is None else "_ex"))
        args = [self.ptr, _addr(d_in), _addr(d_out), channels, length, length if in_ld is None else in_ld,
                (length - 2 * self.n if valid else length) if out_ld is None else out_ld]
        rc = getattr(lib(), name)(*args, *([float(rel_tol)] if rel_tol is not None else ([] if flags is None else [flags])), _stream(stream))
        if rc != 0:
            raise RuntimeError(f"{name} returned {rc}: {last_error()}")

    def apply_tensor(self, x, valid=False, stream=None, flags=None):
        """x: contiguous 2-D torch tensor [channels, length] (float32 or float64) on the GPU."""
        import torch
        assert x.is_cuda and x.dim() == 2 and x.is_contiguous()
        dtype = {torch.float32: "f32", torch.float64: "f64"}[x.dtype]
        ch, length = x.shape
        out_len = length - 2 * self.n if valid else length
        y = torch.empty((ch, out_len), dtype=x.dtype, device=x.device)
        self.apply_batch(x, y, ch, length, length, out_len, dtype=dtype, valid=valid, stream=stream, flags=flags)
        return y


def synth(tensor, channel0=0, seed=0x5A17601A, stream=None):
    """Fill a contiguous [channels, length] GPU tensor with the SURVEY 8(d) synthetic workload."""
    import torch
    ch, length = tensor.shape
    fn = {torch.float32: lib().savgol_hip_synth_f32, torch.float64: lib().savgol_hip_synth_f64}[tensor.dtype]
    rc = fn(tensor.data_ptr(), channel0, ch, length, tensor.stride(0), seed, _stream(stream))
    if rc != 0:
        raise RuntimeError(f"savgol_hip_synth returned {rc}: {last_error()}")
    return tensor


class Stream:
    """savgol_stream_* on one host-side SavgolStream (the reference's streaming API, state in the POD)."""

    def __init__(self, half_window, poly_order, derivative=0, time_step=1.0):
        cfg = SavgolConfig(half_window, poly_order, derivative, time_step, 0)
        self.ptr = lib().savgol_stream_create(C.byref(cfg))
        if not self.ptr:
            raise ValueError("savgol_stream_create rejected the configuration")
        self.n = half_window

    def close(self):
        if getattr(self, "ptr", None) and _lib is not None:
            _lib.savgol_stream_destroy(self.ptr)
        self.ptr = None

    __del__ = close

    def push(self, x):
        ok = C.c_bool(False)
        y = lib().savgol_stream_push(self.ptr, float(x), C.byref(ok))
        return np.float32(y), bool(ok.value)

    def push_full(self, x, max_outputs=SAVGOL_MAX_HALF_WINDOW + 1):
        buf = np.zeros(max(max_outputs, 1), np.float32)
        c = lib().savgol_stream_push_full(self.ptr, float(x), _f(buf), max_outputs)
        return buf[:c].copy()

    def flush(self, max_count=SAVGOL_MAX_HALF_WINDOW):
        buf = np.zeros(max(max_count, 1), np.float32)
        c = lib().savgol_stream_flush(self.ptr, _f(buf), max_count)
        return c, buf[:max(c, 0)].copy()

    def flush_leading(self, max_count=SAVGOL_MAX_HALF_WINDOW):
        buf = np.zeros(max(max_count, 1), np.float32)
        c = lib().savgol_stream_flush_leading(self.ptr, _f(buf), max_count)
        return c, buf[:max(c, 0)].copy()

    @property
    def counters(self):
        s = self.ptr.contents
        return int(s.samples_received), int(s.samples_output), int(s.write_pos)


class StreamBank:
    """savgol_streambank_*: `streams` lock-step streams with their rings in HBM (torch tensors in/out)."""

    def __init__(self, streams, half_window, poly_order, derivative=0, time_step=1.0, fma=False):
        """fma=True: SAVGOL_STREAMBANK_FMA (fused multiply-adds: fast, not the reference's bits)."""
        cfg = SavgolConfig(half_window, poly_order, derivative, time_step, 0)
        self.ptr = lib().savgol_streambank_create_ex(C.byref(cfg), streams, SAVGOL_STREAMBANK_FMA if fma else 0)
        if not self.ptr:
            raise RuntimeError(f"savgol_streambank_create failed: {last_error()}")
        self.streams, self.n = streams, half_window

    def close(self):
        if getattr(self, "ptr", None) and _lib is not None:
            _lib.savgol_streambank_destroy(self.ptr)
        self.ptr = None

    __del__ = close

    def reset(self, stream=None):
        return lib().savgol_streambank_reset(self.ptr, _stream(stream))

    def push(self, samples, out, stream=None):
        return lib().savgol_streambank_push(self.ptr, _addr(samples), _addr(out), _stream(stream))

    def push_wait(self, samples, out, stream=None):
        """push + wait for the outputs through a stream-written completion word (savgol_streambank_push_wait)"""
        return lib().savgol_streambank_push_wait(self.ptr, _addr(samples), _addr(out), _stream(stream))

    def push_full(self, samples, out, max_rows, stream=None):
        return lib().savgol_streambank_push_full(self.ptr, _addr(samples), _addr(out), max_rows, _stream(stream))

    def push_block(self, samples, ticks, out, stream=None):
        return lib().savgol_streambank_push_block(self.ptr, _addr(samples), ticks, _addr(out), _stream(stream))

    def service_start(self, idle_ms=0):
        if lib().savgol_streambank_service_start(self.ptr, idle_ms) != 0:
            raise RuntimeError(last_error())

    def service_tick(self, samples, out):
        return lib().savgol_streambank_service_tick(self.ptr, _addr(samples), _addr(out))

    def service_stop(self):
        if lib().savgol_streambank_service_stop(self.ptr) != 0:
            raise RuntimeError(last_error())

    def flush(self, out, max_rows, stream=None):
        return lib().savgol_streambank_flush(self.ptr, _addr(out), max_rows, _stream(stream))

    def flush_leading(self, out, max_rows, stream=None):
        return lib().savgol_streambank_flush_leading(self.ptr, _addr(out), max_rows, _stream(stream))

    @property
    def counters(self):
        return (lib().savgol_streambank_samples_received(self.ptr), lib().savgol_streambank_samples_output(self.ptr))

    def save(self, stream=None):
        blob = np.zeros(lib().savgol_streambank_state_bytes(self.ptr), np.uint8)
        if lib().savgol_streambank_save(self.ptr, blob.ctypes.data, _stream(stream)) != 0:
            raise RuntimeError(last_error())
        return blob

    def load(self, blob, stream=None):
        if lib().savgol_streambank_load(self.ptr, blob.ctypes.data, _stream(stream)) != 0:
            raise RuntimeError(last_error())


class Filter2D:
    """savgol2d_create / savgol2d_destroy + the apply entry points."""

    def __init__(self, nx, ny, order, dx=0, dy=0, delta_x=1.0, delta_y=1.0):
        cfg = Savgol2DConfig(nx, ny, order, dx, dy, delta_x, delta_y)
        self.ptr = lib().savgol2d_create(C.byref(cfg))
        if not self.ptr:
            raise ValueError("savgol2d_create rejected the configuration")
        self.nx, self.ny = nx, ny

    def close(self):
        if getattr(self, "ptr", None) and _lib is not None:
            _lib.savgol2d_destroy(self.ptr)
        self.ptr = None

    __del__ = close

    @property
    def weights(self):
        f = self.ptr.contents
        return np.array(f.weights[:f.window_area], dtype=np.float32).reshape(f.window_height, f.window_width)

    @property
    def scale(self):
        return np.float32(self.ptr.contents.scale)

    def apply(self, img, cols=None, boundary=SAVGOL2D_BOUNDARY_VALID, out=None):
        """img: 2-D fp32 numpy array, row pitch = img.shape[1]; returns a same-shape frame (copy of `out` or zeros)."""
        img = np.ascontiguousarray(img, np.float32)
        rows, stride = img.shape
        cols = stride if cols is None else cols
        o = np.array(out, np.float32, copy=True) if out is not None else np.zeros_like(img)
        rc = lib().savgol2d_apply(self.ptr, _f(img), rows, cols, stride, _f(o), stride, boundary)
        if rc != 0:
            raise RuntimeError(f"savgol2d_apply returned {rc}: {last_error()}")
        return o

    def apply_valid(self, img, cols=None):
        img = np.ascontiguousarray(img, np.float32)
        rows, stride = img.shape
        cols = stride if cols is None else cols
        o = np.zeros((rows - 2 * self.ny, cols - 2 * self.nx), np.float32)
        rc = lib().savgol2d_apply_valid(self.ptr, _f(img), rows, cols, stride, _f(o), o.shape[1])
        if rc != 0:
            raise RuntimeError(f"savgol2d_apply_valid returned {rc}: {last_error()}")
        return o

    def apply_batch(self, d_in, d_out, rows, cols, images, in_stride=None, out_stride=None, in_pitch=None, out_pitch=None,
                    boundary=SAVGOL2D_BOUNDARY_VALID, method=0, stream=None):
        in_stride = cols if in_stride is None else in_stride
        out_stride = cols if out_stride is None else out_stride
        rc = lib().savgol2d_apply_batch_f32(self.ptr, _addr(d_in), rows, cols, in_stride,
                                            rows * in_stride if in_pitch is None else in_pitch, _addr(d_out), out_stride,
                                            rows * out_stride if out_pitch is None else out_pitch, images, boundary, method,
                                            _stream(stream))
        if rc != 0:
            raise RuntimeError(f"savgol2d_apply_batch_f32 returned {rc}: {last_error()}")
