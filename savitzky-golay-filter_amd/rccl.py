"""ctypes access to RCCL and to lib/libsavgol_hip_rccl.so (csrc/sg_rowband_rccl.hip): the C form of the two halo exchanges, for callers
that do not want `torch.distributed` on the data path -- and for bench.py / the tests, which drive the documented C entry points
rather than a Python re-statement of them.

    uid  = rccl.unique_id()                       # rank 0; 128 bytes, hand them to the other ranks (torch store, MPI, a file ...)
    comm = rccl.Comm(world, rank, uid)            # ncclCommInitRank on the current device
    comm.rowband_exchange(band, ny, up, down, scratch, stream=...)          # savgol2d_rowband_exchange_rccl
    comm.lengthsplit_exchange(seg, n, prev, nxt, halo_prev, halo_next, scratch, periodic=..., stream=...)

The reference has no multi-device path; what the exchanges serve is its whole-frame / whole-channel loops
(src/savgol2d.c:417-453, src/savgolFilter.c:763-766) on data that is cut across GPUs."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_rccl = None
_ext = None


class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def _libs():
    global _rccl, _ext
    if _ext is None:
        rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
        _rccl = C.CDLL(os.path.join(rocm, "lib", "librccl.so"), mode=C.RTLD_GLOBAL)
        _rccl.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
        _rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
        _rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        _rccl.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        _rccl.ncclGetErrorString.restype = C.c_char_p
        _rccl.ncclGetErrorString.argtypes = [C.c_int]
        _ext = C.CDLL(os.environ.get("SAVGOL_HIP_RCCL_LIB") or os.path.join(_HERE, "lib", "libsavgol_hip_rccl.so"))
        vp, sz, i = C.c_void_p, C.c_size_t, C.c_int
        _ext.savgol2d_rowband_exchange_rccl.argtypes = [vp, i, i, vp, i, i, i, sz, sz, i, vp, vp, vp, vp]
        _ext.savgol2d_rowband_exchange_rccl_peers.argtypes = [vp, i, i, vp, i, i, i, sz, sz, i, vp, vp, vp, vp]
        _ext.savgol_lengthsplit_exchange_rccl.argtypes = [vp, i, i, vp, sz, sz, sz, i, i, vp, vp, vp, vp]
    return _rccl, _ext


def available():
    try:
        _libs()
        return True
    except OSError:
        return False


def unique_id():
    """128 bytes identifying a new communicator (ncclGetUniqueId); every rank passes the same bytes to Comm()."""
    nccl, _ = _libs()
    uid = UniqueId()
    rc = nccl.ncclGetUniqueId(C.byref(uid))
    if rc != 0:
        raise RuntimeError(f"ncclGetUniqueId: {nccl.ncclGetErrorString(rc).decode()}")
    return bytes(C.string_at(C.addressof(uid), 128))


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(stream):
    import torch
    return C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)


class Comm:
    def __init__(self, world, rank, uid_bytes):
        nccl, _ = _libs()
        uid = UniqueId()
        C.memmove(C.addressof(uid), uid_bytes, 128)
        self.world, self.rank = int(world), int(rank)
        self.handle = C.c_void_p()
        rc = nccl.ncclCommInitRank(C.byref(self.handle), self.world, uid, self.rank)
        if rc != 0:
            raise RuntimeError(f"ncclCommInitRank: {nccl.ncclGetErrorString(rc).decode()}")

    def count(self):
        """ranks in the communicator as RCCL itself reports them (ncclCommCount): what bench.py prints as `rccl_ranks`"""
        n = C.c_int(-1)
        rc = _libs()[0].ncclCommCount(self.handle, C.byref(n))
        if rc != 0:
            raise RuntimeError(f"ncclCommCount: {_libs()[0].ncclGetErrorString(rc).decode()}")
        return n.value

    def close(self):
        if getattr(self, "handle", None) and self.handle.value:
            _libs()[0].ncclCommDestroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def rowband_exchange(self, band, ny, halo_up, halo_down, scratch, peers=None, stream=None):
        """band: [images, band_rows, cols] fp32 (rows `cols` apart or a view with a larger row stride); halo_up / halo_down: [images, ny,
        cols] receive buffers (None where there is no neighbour); scratch: 2 * images * ny * cols floats.  peers = (up, down) overrides
        rank - 1 / rank + 1 (-1 = none)."""
        images, rows, cols = band.shape
        assert band.stride(2) == 1
        _, ext = _libs()
        args = (_ptr(band), rows, cols, band.stride(1), band.stride(0), images, ny, _ptr(halo_up), _ptr(halo_down), _ptr(scratch), _stream(stream))
        if peers is None:
            rc = ext.savgol2d_rowband_exchange_rccl(self.handle, self.rank, self.world, *args)
        else:
            rc = ext.savgol2d_rowband_exchange_rccl_peers(self.handle, peers[0], peers[1], *args)
        if rc != 0:
            raise RuntimeError("savgol2d_rowband_exchange_rccl failed (bad arguments, or an RCCL / HIP error)")

    def lengthsplit_exchange(self, seg, n, halo_prev, halo_next, scratch, peers, stream=None):
        """seg: [channels, own] fp32 / fp64 segment of every channel; halo_prev / halo_next: [channels, n] receive buffers (None = no
        neighbour); scratch: 2 * channels * n samples; peers = (prev, next), -1 = none."""
        channels, own = seg.shape
        assert seg.stride(1) == 1
        _, ext = _libs()
        rc = ext.savgol_lengthsplit_exchange_rccl(self.handle, peers[0], peers[1], _ptr(seg), channels, own, seg.stride(0), n, seg.element_size(),
                                                  _ptr(halo_prev), _ptr(halo_next), _ptr(scratch), _stream(stream))
        if rc != 0:
            raise RuntimeError("savgol_lengthsplit_exchange_rccl failed (bad arguments, or an RCCL / HIP error)")
