#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the COMPILED, UNMODIFIED reference.

Run in the build container only (needs /root/reference and `make -C oracle ref`):

    python tests/golden/make_golden.py

It drives oracle/_ref/libsavgol_ref.so (gcc -O2 -ffp-contract=off build of the reference's
src/savgolFilter.c, src/savgol_stream.c, src/savgol2d.c) through ctypes on seeded inputs and
stores inputs + outputs as small .npz files.  Fixtures are DATA: numbers only, no reference
source text.  Two of them carry data the reference's own files hold:

  * matlab_pair.npz  -- the 301-point `rawData` / `yourSavgolData` arrays embedded in
                        "tool for matlab comparisons/savgolComparison.m" (lines 2 and 5), the only
                        golden vector in the reference repository (window 13, degree 3);
  * demo360.npz      -- the 360-point dataset embedded in test/iterative/test_savgol_main.c:55-92.
"""
import ctypes as C
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("SAVGOL_REFERENCE", "/root/reference")
LIB = os.path.join(ROOT, "oracle", "_ref", "libsavgol_ref.so")

MAXN, MAXWS = 32, 65


class Cfg(C.Structure):          # savgolFilter.h:92-98
    _fields_ = [("half_window", C.c_uint8), ("poly_order", C.c_uint8), ("derivative", C.c_uint8),
                ("time_step", C.c_float), ("boundary", C.c_int)]


class Filt(C.Structure):         # savgolFilter.h:107-113
    _fields_ = [("config", Cfg), ("window_size", C.c_int), ("dt_scale", C.c_float),
                ("center_weights", C.c_float * MAXWS), ("edge_weights", (C.c_float * MAXWS) * MAXN)]


class Stream(C.Structure):       # savgol_stream.h:29-37
    _fields_ = [("filter", C.POINTER(Filt)), ("buffer", C.c_float * MAXWS), ("write_pos", C.c_int),
                ("samples_received", C.c_size_t), ("samples_output", C.c_size_t),
                ("owns_filter", C.c_bool), ("dt_inv", C.c_float)]


class Cfg2(C.Structure):         # savgol2d.h:82-90
    _fields_ = [("half_window_x", C.c_uint8), ("half_window_y", C.c_uint8), ("poly_order", C.c_uint8),
                ("deriv_x", C.c_uint8), ("deriv_y", C.c_uint8), ("delta_x", C.c_float), ("delta_y", C.c_float)]


class Filt2(C.Structure):        # savgol2d.h:95-103
    _fields_ = [("config", Cfg2), ("window_width", C.c_int), ("window_height", C.c_int),
                ("window_area", C.c_int), ("num_terms", C.c_int), ("scale", C.c_float),
                ("weights", C.POINTER(C.c_float))]


def load():
    assert C.sizeof(Cfg) == 12 and C.sizeof(Filt) == 8600 and C.sizeof(Stream) == 296
    assert C.sizeof(Cfg2) == 16 and C.sizeof(Filt2) == 48
    L = C.CDLL(LIB)
    fp, vp = C.POINTER(C.c_float), C.c_void_p
    L.savgol_create.restype = C.POINTER(Filt); L.savgol_create.argtypes = [C.POINTER(Cfg)]
    L.savgol_destroy.argtypes = [C.POINTER(Filt)]
    L.savgol_apply.argtypes = [C.POINTER(Filt), fp, fp, C.c_size_t]
    L.savgol_apply_valid.restype = C.c_size_t
    L.savgol_apply_valid.argtypes = [C.POINTER(Filt), fp, C.c_size_t, fp]
    L.savgol_apply_strided.argtypes = [C.POINTER(Filt), vp, C.c_size_t, C.c_size_t, vp, C.c_size_t, C.c_size_t, C.c_size_t]
    L.savgol_stream_create.restype = C.POINTER(Stream); L.savgol_stream_create.argtypes = [C.POINTER(Cfg)]
    L.savgol_stream_destroy.argtypes = [C.POINTER(Stream)]
    L.savgol_stream_push.restype = C.c_float
    L.savgol_stream_push.argtypes = [C.POINTER(Stream), C.c_float, C.POINTER(C.c_bool)]
    L.savgol_stream_push_full.argtypes = [C.POINTER(Stream), C.c_float, fp, C.c_int]
    L.savgol_stream_flush.argtypes = [C.POINTER(Stream), fp, C.c_int]
    L.savgol_stream_flush_leading.argtypes = [C.POINTER(Stream), fp, C.c_int]
    L.savgol2d_create.restype = C.POINTER(Filt2); L.savgol2d_create.argtypes = [C.POINTER(Cfg2)]
    L.savgol2d_destroy.argtypes = [C.POINTER(Filt2)]
    L.savgol2d_apply.argtypes = [C.POINTER(Filt2), fp, C.c_int, C.c_int, C.c_int, fp, C.c_int, C.c_int]
    L.savgol2d_apply_valid.argtypes = [C.POINTER(Filt2), fp, C.c_int, C.c_int, C.c_int, fp, C.c_int]
    L.savgol2d_gradient.argtypes = [C.c_int] * 3 + [fp, C.c_int, C.c_int, C.c_int, fp, fp, C.c_float, C.c_float, C.c_int]
    L.savgol2d_hessian.argtypes = [C.c_int] * 3 + [fp, C.c_int, C.c_int, C.c_int, fp, fp, fp, C.c_float, C.c_float, C.c_int]
    L.savgol2d_laplacian.argtypes = [C.c_int] * 3 + [fp, C.c_int, C.c_int, C.c_int, fp, C.c_float, C.c_float, C.c_int]
    return L


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def signal(rng, length):
    """smooth-ish random test signal with structure at several scales, fp32"""
    t = np.arange(length, dtype=np.float64)
    x = (np.sin(0.031 * t + rng.uniform(0, 6.28)) * rng.uniform(0.5, 3.0)
         + 0.4 * np.sin(0.37 * t + rng.uniform(0, 6.28))
         + 0.002 * t + rng.normal(0, 0.25, length))
    return x.astype(np.float32)


def filt_tables(f):
    n = f.contents.config.half_window
    ws = f.contents.window_size
    cw = np.array(f.contents.center_weights[:ws], dtype=np.float32)
    ew = np.array([list(f.contents.edge_weights[e][:ws]) for e in range(n)], dtype=np.float32).reshape(n, ws)
    return cw, ew


WEIGHT_GRID = [(5, 3, 0), (32, 4, 0), (32, 4, 2), (16, 2, 1), (6, 3, 0), (32, 10, 4), (1, 0, 0),
               (2, 2, 2), (10, 3, 1), (3, 2, 0), (12, 4, 0), (32, 4, 1), (8, 5, 3), (25, 6, 2),
               (32, 2, 0), (4, 4, 4), (1, 2, 1), (20, 10, 0)]


def gen_weights(L, out):
    for (n, m, d) in WEIGHT_GRID:
        for dt in (1.0, 1e-3, 0.25):
            cfg = Cfg(n, m, d, dt, 0)
            f = L.savgol_create(C.byref(cfg))
            assert f, (n, m, d)
            cw, ew = filt_tables(f)
            key = f"n{n}_m{m}_d{d}"
            out[key + "_center"] = cw
            out[key + "_edges"] = ew
            out[key + f"_dtscale_{dt:g}"] = np.float32(f.contents.dt_scale)
            L.savgol_destroy(f)


APPLY_CASES = [  # (n, m, d, time_step, length)
    (5, 3, 0, 1.0, 257), (5, 3, 1, 0.01, 257), (5, 3, 2, 0.5, 300),
    (32, 4, 0, 1.0, 1000), (32, 4, 1, 1.0, 1000), (32, 4, 2, 1.0, 1000),
    (16, 2, 1, 1e-3, 500), (6, 3, 0, 1.0, 13), (2, 2, 0, 1.0, 5), (32, 4, 0, 1.0, 65),
    (10, 3, 1, 2.0, 64), (3, 2, 0, 1.0, 4099), (32, 10, 4, 0.1, 400),
]


def gen_apply(L, out, rng):
    for ci, (n, m, d, dt, length) in enumerate(APPLY_CASES):
        x = signal(rng, length)
        out[f"c{ci}_in"] = x
        out[f"c{ci}_cfg"] = np.array([n, m, d, length], dtype=np.int64)
        out[f"c{ci}_dt"] = np.float32(dt)
        for mode in range(4):
            cfg = Cfg(n, m, d, dt, mode)
            f = L.savgol_create(C.byref(cfg))
            y = np.full(length, np.nan, dtype=np.float32)
            assert L.savgol_apply(f, fptr(x), fptr(y), length) == 0
            out[f"c{ci}_mode{mode}_out"] = y
            if mode == 0:
                v = np.full(length - 2 * n, np.nan, dtype=np.float32)
                got = L.savgol_apply_valid(f, fptr(x), length, fptr(v))
                assert got == length - 2 * n
                out[f"c{ci}_valid_out"] = v
                # strided: array of {float a; float value; float b}
                aos = rng.normal(0, 1, (length, 3)).astype(np.float32)
                aos[:, 1] = x
                dst = aos.copy()
                assert L.savgol_apply_strided(f, aos.ctypes.data, 12, 4, dst.ctypes.data, 12, 4, length) == 0
                out[f"c{ci}_strided_in"] = aos
                out[f"c{ci}_strided_out"] = dst
            L.savgol_destroy(f)
    # error behaviour: too-short input
    cfg = Cfg(5, 3, 0, 1.0, 0)
    f = L.savgol_create(C.byref(cfg))
    x = np.zeros(10, np.float32); y = np.zeros(10, np.float32)
    out["short_rc"] = np.int64(L.savgol_apply(f, fptr(x), fptr(y), 10))
    out["short_valid_rc"] = np.int64(L.savgol_apply_valid(f, fptr(x), 10, fptr(y)))
    L.savgol_destroy(f)
    # leading-edge sign quirk: d=1 on y = 3x + 7, n=5, m=2
    cfg = Cfg(5, 2, 1, 1.0, 0)
    f = L.savgol_create(C.byref(cfg))
    x = (3.0 * np.arange(50) + 7.0).astype(np.float32); y = np.zeros(50, np.float32)
    L.savgol_apply(f, fptr(x), fptr(y), 50)
    out["quirk_in"] = x; out["quirk_out"] = y
    L.savgol_destroy(f)


STREAM_CASES = [(5, 3, 0, 1.0, 100), (16, 2, 1, 1e-3, 200), (32, 4, 0, 1.0, 150), (3, 2, 2, 0.5, 7), (4, 2, 0, 1.0, 5)]


def gen_stream(L, out, rng):
    for ci, (n, m, d, dt, count) in enumerate(STREAM_CASES):
        x = signal(rng, count)
        out[f"s{ci}_in"] = x
        out[f"s{ci}_cfg"] = np.array([n, m, d, count], dtype=np.int64)
        out[f"s{ci}_dt"] = np.float32(dt)
        cfg = Cfg(n, m, d, dt, 0)
        # plain push
        s = L.savgol_stream_create(C.byref(cfg))
        vals, valid = [], []
        ok = C.c_bool(False)
        for v in x:
            r = L.savgol_stream_push(s, float(v), C.byref(ok))
            vals.append(r); valid.append(bool(ok.value))
        out[f"s{ci}_push_val"] = np.array(vals, dtype=np.float32)
        out[f"s{ci}_push_valid"] = np.array(valid, dtype=np.bool_)
        out[f"s{ci}_push_counters"] = np.array([s.contents.samples_received, s.contents.samples_output,
                                                 s.contents.write_pos], dtype=np.int64)
        L.savgol_stream_destroy(s)
        # push_full + flush (+ flush_leading)
        s = L.savgol_stream_create(C.byref(cfg))
        buf = np.zeros(MAXN + 1, np.float32)
        seq, counts = [], []
        for v in x:
            c = L.savgol_stream_push_full(s, float(v), fptr(buf), MAXN + 1)
            counts.append(c); seq.extend(buf[:c].tolist())
        lead = np.zeros(MAXN, np.float32)
        nl = L.savgol_stream_flush_leading(s, fptr(lead), MAXN)
        fl = np.zeros(MAXN, np.float32)
        nf = L.savgol_stream_flush(s, fptr(fl), MAXN)
        out[f"s{ci}_full_seq"] = np.array(seq, dtype=np.float32)
        out[f"s{ci}_full_counts"] = np.array(counts, dtype=np.int64)
        out[f"s{ci}_flush_leading"] = lead[:max(nl, 0)].copy()
        out[f"s{ci}_flush_leading_rc"] = np.int64(nl)
        out[f"s{ci}_flush"] = fl[:max(nf, 0)].copy()
        out[f"s{ci}_flush_rc"] = np.int64(nf)
        out[f"s{ci}_full_counters"] = np.array([s.contents.samples_received, s.contents.samples_output,
                                                 s.contents.write_pos], dtype=np.int64)
        # truncated burst: max_outputs smaller than n+1 silently drops (savgol_stream.c:209-218)
        L.savgol_stream_destroy(s)
        s = L.savgol_stream_create(C.byref(cfg))
        tr = []
        for v in x:
            c = L.savgol_stream_push_full(s, float(v), fptr(buf), 2)
            tr.extend(buf[:c].tolist())
        out[f"s{ci}_full_trunc2"] = np.array(tr, dtype=np.float32)
        L.savgol_stream_destroy(s)


CASES_2D = [  # nx, ny, order, deltas
    (3, 3, 2, 1.0, 1.0), (7, 7, 3, 1.0, 1.0), (2, 1, 2, 1.0, 1.0), (7, 7, 4, 0.5, 2.0), (16, 16, 6, 1.0, 1.0),
    (4, 6, 3, 0.1, 0.2),
]
DERIVS = [(0, 0), (1, 0), (0, 1), (2, 0), (1, 1), (0, 2)]


def gen_2d(L, out, rng):
    rows, cols, stride = 44, 57, 64
    img = np.zeros((rows, stride), np.float32)
    yy, xx = np.mgrid[0:rows, 0:cols]
    img[:, :cols] = (np.sin(0.21 * xx) * np.cos(0.17 * yy) + 0.01 * xx * yy / 10.0
                     + rng.normal(0, 0.1, (rows, cols))).astype(np.float32)
    out["img"] = img
    out["img_dims"] = np.array([rows, cols, stride], dtype=np.int64)
    for ci, (nx, ny, order, ddx, ddy) in enumerate(CASES_2D):
        out[f"k{ci}_cfg"] = np.array([nx, ny, order], dtype=np.int64)
        out[f"k{ci}_delta"] = np.array([ddx, ddy], dtype=np.float32)
        for (dx, dy) in DERIVS:
            if dx + dy > order:
                continue
            cfg = Cfg2(nx, ny, order, dx, dy, ddx, ddy)
            f = L.savgol2d_create(C.byref(cfg))
            assert f, (nx, ny, order, dx, dy)
            area = f.contents.window_area
            W = np.array(f.contents.weights[:area], dtype=np.float32).reshape(2 * ny + 1, 2 * nx + 1)
            out[f"k{ci}_d{dx}{dy}_W"] = W
            out[f"k{ci}_d{dx}{dy}_scale"] = np.float32(f.contents.scale)
            if rows > 2 * ny and cols > 2 * nx:
                for b in range(3):
                    o = np.full((rows, stride), -777.0, dtype=np.float32)
                    rc = L.savgol2d_apply(f, fptr(img), rows, cols, stride, fptr(o), stride, b)
                    assert rc == 0
                    out[f"k{ci}_d{dx}{dy}_b{b}_out"] = o
            L.savgol2d_destroy(f)
    # helper wrappers (gradient / hessian / laplacian), n=3 order 2 and n=7 order 3, CONSTANT + REFLECT
    for hi, (n, order, b, dxs, dys) in enumerate([(3, 2, 1, 1.0, 1.0), (7, 3, 2, 0.5, 0.25), (3, 3, 0, 1.0, 1.0)]):
        gx = np.full((rows, stride), -777.0, np.float32); gy = gx.copy()
        hxx = gx.copy(); hxy = gx.copy(); hyy = gx.copy(); lap = gx.copy()
        assert L.savgol2d_gradient(n, n, order, fptr(img), rows, cols, stride, fptr(gx), fptr(gy), dxs, dys, b) == 0
        assert L.savgol2d_hessian(n, n, order, fptr(img), rows, cols, stride, fptr(hxx), fptr(hxy), fptr(hyy), dxs, dys, b) == 0
        assert L.savgol2d_laplacian(n, n, order, fptr(img), rows, cols, stride, fptr(lap), dxs, dys, b) == 0
        out[f"h{hi}_cfg"] = np.array([n, order, b], dtype=np.int64)
        out[f"h{hi}_delta"] = np.array([dxs, dys], dtype=np.float32)
        for name, arr in (("gx", gx), ("gy", gy), ("hxx", hxx), ("hxy", hxy), ("hyy", hyy), ("lap", lap)):
            out[f"h{hi}_{name}"] = arr


def parse_matlab():
    path = os.path.join(REF, "tool for matlab comparisons", "savgolComparison.m")
    txt = open(path).read()
    arrays = {}
    for name in ("rawData", "yourSavgolData"):
        m = re.search(name + r"\s*=\s*\[(.*?)\]", txt, re.S)
        arrays[name] = np.array([float(v) for v in m.group(1).replace("...", " ").replace(";", " ").split(",") if v.strip()],
                                dtype=np.float64)
    return arrays


def parse_demo_dataset():
    path = os.path.join(REF, "test", "iterative", "test_savgol_main.c")
    txt = open(path).read()
    m = re.search(r"float\s+dataset\[\]\s*=\s*\{(.*?)\};", txt, re.S)
    vals = re.findall(r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?f", m.group(1))
    return np.array([np.float32(v[:-1]) for v in vals], dtype=np.float32)


def gen_embedded(L, out_m, out_d):
    arr = parse_matlab()
    raw = arr["rawData"]; theirs = arr["yourSavgolData"]
    assert raw.size == theirs.size
    x = raw.astype(np.float32)
    cfg = Cfg(6, 3, 0, 1.0, 0)
    f = L.savgol_create(C.byref(cfg))
    y = np.zeros_like(x)
    assert L.savgol_apply(f, fptr(x), fptr(y), x.size) == 0
    out_m["rawData"] = raw; out_m["yourSavgolData"] = theirs; out_m["ref_out_f32"] = y
    L.savgol_destroy(f)

    ds = parse_demo_dataset()
    out_d["dataset"] = ds
    for tag, (n, m, d) in (("smooth_n6_m3", (6, 3, 0)), ("deriv1_n10_m3", (10, 3, 1))):
        cfg = Cfg(n, m, d, 1.0, 0)
        f = L.savgol_create(C.byref(cfg))
        y = np.zeros_like(ds)
        assert L.savgol_apply(f, fptr(ds), fptr(y), ds.size) == 0
        out_d[tag] = y
        L.savgol_destroy(f)


EXPORT_CASES = [(5, 2, 0, None), (10, 3, 1, "deriv"), (32, 4, 0, "sg32"), (3, 2, 2, "Mixed_Case9"), (1, 0, 0, "one")]


def gen_export():
    """Headers written by the reference's own savgol_export tool (oracle/_ref/savgol_export, built unmodified by
    `make -C oracle ref`) -- the reference's on-disk format for the weight tables.  The tool's "Generated by ... on <time>"
    line is the only thing that changes from run to run; the time is replaced by @TIMESTAMP@ in the fixture."""
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "savgol_export")
    for n, m, d, prefix in EXPORT_CASES:
        cmd = [exe, "-n", str(n), "-m", str(m), "-d", str(d)] + (["-p", prefix] if prefix else [])
        text = subprocess.run(cmd, capture_output=True, text=True, check=True).stdout
        text = re.sub(r"(Generated by savgol_export on )\d{4}-\d\d-\d\d \d\d:\d\d:\d\d", r"\1@TIMESTAMP@", text)
        assert "@TIMESTAMP@" in text
        name = f"export_n{n}_m{m}_d{d}_{prefix or 'SAVGOL'}.txt"
        open(os.path.join(HERE, name), "w").write(text)
        print(f"{name}  {len(text)} bytes")


def main():
    if "--export-only" in sys.argv:
        return gen_export()
    if not os.path.exists(LIB):
        sys.exit(f"{LIB} missing: run `make -C oracle ref` first")
    L = load()
    rng = np.random.default_rng(0x5A17601A)
    w, a, s, d2, mt, dm = {}, {}, {}, {}, {}, {}
    gen_weights(L, w); gen_apply(L, a, rng); gen_stream(L, s, rng); gen_2d(L, d2, rng); gen_embedded(L, mt, dm)
    for name, d in (("weights1d", w), ("apply1d", a), ("stream", s), ("filter2d", d2), ("matlab_pair", mt), ("demo360", dm)):
        p = os.path.join(HERE, name + ".npz")
        np.savez_compressed(p, **d)
        print(f"{name:12s} {len(d):4d} arrays  {os.path.getsize(p)/1024:.1f} KiB")
    gen_export()


if __name__ == "__main__":
    main()
