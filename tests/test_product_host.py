"""CPU-side checks of the product (no GPU needed): the C-ABI library loads and exports every symbol
include/*.h declares, struct layouts match the reference's, the host weight generator is bit-identical
to the golden tables, config validation mirrors the reference, and the apply entry points refuse to
run (loudly) without a device instead of falling back to the CPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from tests._util import bits
from tests.golden.make_golden import WEIGHT_GRID

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = set()
    for h in ("savgolFilter.h", "savgol_stream.h", "savgol2d.h", "savgol_hip.h"):
        txt = open(os.path.join(ROOT, "include", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        txt = re.sub(r"static inline[^{]*\{.*?\n\}", "", txt, flags=re.S)
        names |= set(re.findall(r"\b(savgol\w*)\s*\(", txt))
    return {n for n in names if not n.isupper()}


def test_library_exports_every_declared_symbol(sg):
    L = C.CDLL(sg.LIB_PATH)
    missing = [n for n in sorted(declared_functions()) if not hasattr(L, n)]
    assert not missing, f"declared in include/*.h but not exported: {missing}"
    unbound = [n for n in sorted(declared_functions()) if n not in sg.SIGNATURES]
    assert not unbound, f"declared in include/*.h but not bound in the Python mirror: {unbound}"


def test_optional_rccl_library_exports_its_header(sg):
    """include/savgol_hip_rccl.h is served by lib/libsavgol_hip_rccl.so (the only object that links librccl); the main library
    must NOT depend on RCCL."""
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "savgol_hip_rccl.h")).read(), flags=re.S)
    names = set(re.findall(r"\b(savgol\w*)\s*\(", txt))
    assert names == {"savgol2d_rowband_exchange_rccl", "savgol2d_rowband_exchange_rccl_peers", "savgol_lengthsplit_exchange_rccl"}
    path = os.path.join(os.path.dirname(sg.LIB_PATH), "libsavgol_hip_rccl.so")
    assert os.path.exists(path), "make builds it next to libsavgol_hip.so"
    L = C.CDLL(path)
    for n in names:
        assert hasattr(L, n)
    import subprocess
    needed = subprocess.run(["readelf", "-d", sg.LIB_PATH], capture_output=True, text=True).stdout
    assert "rccl" not in needed and "nccl" not in needed


def test_struct_layouts_match_reference(sg):
    # SURVEY 8b [probed on the reference]: sizes and field offsets are part of the drop-in contract
    assert C.sizeof(sg.SavgolConfig) == 12
    assert [getattr(sg.SavgolConfig, f).offset for f in ("half_window", "poly_order", "derivative", "time_step", "boundary")] == [0, 1, 2, 4, 8]
    assert C.sizeof(sg.SavgolFilter) == 8600
    assert [getattr(sg.SavgolFilter, f).offset for f in ("config", "window_size", "dt_scale", "center_weights", "edge_weights")] == [0, 12, 16, 20, 280]
    assert C.sizeof(sg.SavgolStream) == 296
    assert [getattr(sg.SavgolStream, f).offset for f in ("filter", "buffer", "write_pos", "samples_received", "samples_output", "owns_filter", "dt_inv")] == [0, 8, 268, 272, 280, 288, 292]
    assert C.sizeof(sg.Savgol2DConfig) == 16
    assert C.sizeof(sg.Savgol2DFilter) == 48
    assert [getattr(sg.Savgol2DFilter, f).offset for f in ("config", "window_width", "window_height", "window_area", "num_terms", "scale", "weights")] == [0, 16, 20, 24, 28, 32, 40]


@pytest.mark.parametrize("n,m,d", WEIGHT_GRID)
def test_host_weight_tables_bit_identical_to_reference(sg, golden, n, m, d):
    g = golden("weights1d")
    key = f"n{n}_m{m}_d{d}"
    for dt in (1.0, 1e-3, 0.25):
        f = sg.Filter(n, m, d, dt)
        assert np.array_equal(bits(f.center_weights), bits(g[key + "_center"]))
        assert np.array_equal(bits(f.edge_weights), bits(g[key + "_edges"]))
        assert bits(f.dt_scale) == bits(g[key + f"_dtscale_{dt:g}"])
        assert f.ptr.contents.window_size == 2 * n + 1
        f.close()


def test_host_weight_tables_match_oracle_on_random_configs(sg, sgo):
    rng = np.random.default_rng(11)
    for _ in range(200):
        n = int(rng.integers(1, 33)); m = int(rng.integers(0, min(2 * n, 12) + 1)); d = int(rng.integers(0, min(m, 4) + 1))
        w = sgo.weights(n, m, d)
        if w is None:
            with pytest.raises(ValueError):
                sg.Filter(n, m, d)
            continue
        f = sg.Filter(n, m, d, 0.37)
        assert np.array_equal(bits(f.center_weights), bits(w[0])), (n, m, d)
        assert np.array_equal(bits(f.edge_weights), bits(w[1])), (n, m, d)
        assert bits(f.dt_scale) == bits(sgo.dt_scale(0.37, d))


def test_create_validation_like_reference_tests(sg):
    # reference test_savgol.c:37-85: NULL for n=0, m >= window, d > m; destroy(NULL) is a no-op
    L = sg.lib()
    for cfg in [(0, 2, 0, 1.0), (2, 5, 0, 1.0), (5, 2, 3, 1.0), (33, 2, 0, 1.0), (5, 6, 5, 1.0), (5, 3, 0, 0.0), (5, 3, 0, -1.0),
                (32, 11, 0, 1.0)]:
        c = sg.SavgolConfig(cfg[0], cfg[1], cfg[2], cfg[3], 0)
        assert not L.savgol_create(C.byref(c)), cfg
    assert not L.savgol_create(None)
    L.savgol_destroy(None)
    f = sg.Filter(5, 3)
    cw = f.center_weights
    assert abs(cw.sum() - 1.0) < 1e-5                       # test_savgol.c:91-105
    assert np.allclose(cw, cw[::-1], atol=1e-6)             # :107-121
    d1 = sg.Filter(5, 3, 1).center_weights
    assert np.allclose(d1, -d1[::-1], atol=1e-6) and abs(d1[5]) < 1e-6   # :123-140


def test_apply_argument_errors_need_no_device(sg):
    L = sg.lib()
    f = sg.Filter(5, 3)
    x = np.zeros(10, np.float32)
    fp = C.POINTER(C.c_float)
    assert L.savgol_apply(None, x.ctypes.data_as(fp), x.ctypes.data_as(fp), 10) == -1
    assert L.savgol_apply(f.ptr, None, x.ctypes.data_as(fp), 10) == -1
    assert L.savgol_apply(f.ptr, x.ctypes.data_as(fp), x.ctypes.data_as(fp), 10) == -1       # shorter than the window
    assert L.savgol_apply_valid(f.ptr, x.ctypes.data_as(fp), 10, x.ctypes.data_as(fp)) == 0
    assert L.savgol_apply_strided(f.ptr, None, 4, 0, x.ctypes.data, 4, 0, 20) == -1
    assert L.savgol_apply_batch_f32(f.ptr, None, None, 1, 100, 100, 100, None) == -1
    assert "NULL" in sg.last_error()


def test_no_cpu_fallback_without_device(sg):
    """On a box without a GPU every compute entry point must FAIL, not quietly compute on the host."""
    if sg.device_count() > 0:
        pytest.skip("a GPU is present")
    f = sg.Filter(5, 3)
    x = np.arange(100, dtype=np.float32)
    with pytest.raises(RuntimeError):
        f.apply(x)
    assert f.apply_valid(x).size == 0
    assert "no usable HIP device" in sg.last_error()


def test_rccl_helper_module_imports_without_a_gpu(sg):
    """savitzky-golay-filter_amd/rccl.py (ctypes access to librccl + lib/libsavgol_hip_rccl.so) must import on a box without a GPU and say
    whether the two libraries load; the exchange library exports the three entry points its header declares (checked above) and the
    argument lists the Python side binds match the header's parameter counts."""
    import importlib
    rccl = importlib.import_module("savgol_amd.rccl")
    assert isinstance(rccl.available(), bool)
    if rccl.available():
        _, ext = rccl._libs()
        assert len(ext.savgol2d_rowband_exchange_rccl.argtypes) == 14 and len(ext.savgol2d_rowband_exchange_rccl_peers.argtypes) == 14
        assert len(ext.savgol_lengthsplit_exchange_rccl.argtypes) == 13
        hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "savgol_hip_rccl.h")).read(), flags=re.S)
        for name, count in (("savgol2d_rowband_exchange_rccl", 14), ("savgol2d_rowband_exchange_rccl_peers", 14), ("savgol_lengthsplit_exchange_rccl", 13)):
            m = re.search(r"\b" + name + r"\s*\((.*?)\)\s*;", hdr, flags=re.S)
            assert m and len(m.group(1).split(",")) == count, name
