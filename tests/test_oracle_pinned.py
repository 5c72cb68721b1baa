"""Pin the CPU oracle (oracle/sg_oracle.c) before anything else trusts it.

(1) bit-for-bit against the golden fixtures produced by the compiled, unmodified reference
    (tests/golden/make_golden.py);
(2) bit-for-bit against the compiled reference itself when oracle/_ref/libsavgol_ref.so is present
    (built from /root/reference by `make -C oracle ref`; it travels to the GPU box as a binary);
(3) against the one golden vector the reference ships: the 301-point MATLAB comparison pair
    ("tool for matlab comparisons/savgolComparison.m":2,5 -- README.md:253-256 claims 1e-6-order
    agreement).
"""
import numpy as np
import pytest

from tests._util import bits, normwise
from tests.golden.make_golden import APPLY_CASES, CASES_2D, DERIVS, STREAM_CASES, WEIGHT_GRID


def same_bits(a, b):
    return a.shape == b.shape and np.array_equal(bits(a), bits(b))


@pytest.mark.parametrize("n,m,d", WEIGHT_GRID)
def test_weight_tables_bit_exact(sgo, golden, n, m, d):
    g = golden("weights1d")
    cw, ew = sgo.weights(n, m, d)
    key = f"n{n}_m{m}_d{d}"
    assert same_bits(cw, g[key + "_center"])
    assert same_bits(ew, g[key + "_edges"])
    for dt in (1.0, 1e-3, 0.25):
        assert bits(sgo.dt_scale(dt, d)) == bits(g[key + f"_dtscale_{dt:g}"])


def test_weight_validation_matches_reference(sgo):
    # validate_config, savgolFilter.c:639-677 (+ the GenFact bound the oracle adds, see sg_oracle.c)
    assert sgo.weights(0, 0, 0) is None            # half_window == 0
    assert sgo.weights(33, 2, 0) is None           # half_window > 32
    assert sgo.weights(2, 5, 0) is None            # poly_order >= window
    assert sgo.weights(5, 3, 4) is None            # derivative > poly_order
    assert sgo.weights(5, 6, 5) is None            # derivative > 4
    assert sgo.weights(32, 11, 0) is None          # 2n+m+1 >= 76 : reference emits garbage, we refuse
    assert sgo.weights(32, 10, 4) is not None


@pytest.mark.parametrize("ci", range(len(APPLY_CASES)))
def test_apply_bit_exact_all_modes(sgo, golden, ci):
    g = golden("apply1d")
    n, m, d, length = (int(v) for v in g[f"c{ci}_cfg"])
    x = g[f"c{ci}_in"]
    f = sgo.Filter(n, m, d, float(g[f"c{ci}_dt"]))
    for mode in range(4):
        assert same_bits(f.apply(x, mode), g[f"c{ci}_mode{mode}_out"]), f"mode {mode}"
    assert same_bits(f.apply_valid(x), g[f"c{ci}_valid_out"])
    src = g[f"c{ci}_strided_in"].copy()
    dst = src.copy()
    assert f.apply_strided(src, 12, 4, dst, 12, 4, length) == 0
    assert same_bits(dst, g[f"c{ci}_strided_out"])


def test_apply_error_codes_and_sign_quirk(sgo, golden):
    g = golden("apply1d")
    f = sgo.Filter(5, 3, 0)
    with pytest.raises(ValueError):
        f.apply(np.zeros(10, np.float32))
    assert int(g["short_rc"]) == -1 and int(g["short_valid_rc"]) == 0
    assert f.apply_valid(np.zeros(10, np.float32)).size == 0
    # d=1 leading edge has the opposite sign (reference savgolFilter.c:773-777)
    q = sgo.Filter(5, 2, 1)
    y = q.apply(g["quirk_in"])
    assert same_bits(y, g["quirk_out"])
    assert np.allclose(y[:5], -3.0, atol=1e-3) and np.allclose(y[5:], 3.0, atol=1e-3)


@pytest.mark.parametrize("ci", range(len(STREAM_CASES)))
def test_stream_bit_exact(sgo, golden, ci):
    g = golden("stream")
    n, m, d, count = (int(v) for v in g[f"s{ci}_cfg"])
    x = g[f"s{ci}_in"]
    f = sgo.Filter(n, m, d, float(g[f"s{ci}_dt"]))
    s = sgo.Stream(f)
    vals, valid = zip(*[s.push(v) for v in x])
    assert same_bits(np.array(vals, np.float32), g[f"s{ci}_push_val"])
    assert np.array_equal(np.array(valid), g[f"s{ci}_push_valid"])
    assert list(s.counters) == list(g[f"s{ci}_push_counters"])

    s = sgo.Stream(f)
    seq, counts = [], []
    for v in x:
        o = s.push_full(v)
        counts.append(o.size); seq.extend(o.tolist())
    assert np.array_equal(np.array(counts), g[f"s{ci}_full_counts"])
    assert same_bits(np.array(seq, np.float32), g[f"s{ci}_full_seq"])
    c, lead = s.flush_leading()
    assert c == int(g[f"s{ci}_flush_leading_rc"]) and same_bits(lead, g[f"s{ci}_flush_leading"])
    c, tail = s.flush()
    assert c == int(g[f"s{ci}_flush_rc"]) and same_bits(tail, g[f"s{ci}_flush"])
    assert list(s.counters) == list(g[f"s{ci}_full_counters"])

    s = sgo.Stream(f)
    tr = []
    for v in x:
        tr.extend(s.push_full(v, 2).tolist())
    assert same_bits(np.array(tr, np.float32), g[f"s{ci}_full_trunc2"])


def test_stream_equals_batch_like_reference_test(sgo, golden):
    # reference test_savgol_stream.c:140-189: push_full + flush == savgol_apply within 1e-5
    g = golden("stream")
    x = g["s0_in"]
    f = sgo.Filter(5, 3, 0)
    seq = np.concatenate([g["s0_full_seq"], g["s0_flush"]])
    assert seq.size == x.size
    assert np.max(np.abs(seq - f.apply(x))) < 1e-5


@pytest.mark.parametrize("ci", range(len(CASES_2D)))
def test_2d_kernels_and_outputs_bit_exact(sgo, golden, ci):
    g = golden("filter2d")
    nx, ny, order = (int(v) for v in g[f"k{ci}_cfg"])
    ddx, ddy = (float(v) for v in g[f"k{ci}_delta"])
    img = g["img"]
    rows, cols, stride = (int(v) for v in g["img_dims"])
    for dx, dy in DERIVS:
        if dx + dy > order:
            continue
        f = sgo.Filter2D(nx, ny, order, dx, dy, ddx, ddy)
        assert same_bits(f.W, g[f"k{ci}_d{dx}{dy}_W"]), (dx, dy)
        assert bits(f.scale) == bits(g[f"k{ci}_d{dx}{dy}_scale"])
        if rows > 2 * ny and cols > 2 * nx:
            for b in range(3):
                want = g[f"k{ci}_d{dx}{dy}_b{b}_out"]
                got = f.apply(img, cols, b, out=np.full_like(img, -777.0))
                assert same_bits(got, want), (dx, dy, b)
                # the double-accumulation oracle sits within fp32 rounding of it
                hi = f.apply_f64acc(img, cols, b)
                sel = want != -777.0
                assert normwise(hi[sel], want[sel]) < 5e-6


def test_matlab_golden_vector(sgo, golden):
    g = golden("matlab_pair")
    raw, theirs = g["rawData"], g["yourSavgolData"]
    assert raw.size == 301 and theirs.size == 301
    f = sgo.Filter(6, 3, 0)                          # window 13, degree 3 (savgolComparison.m:7-9)
    y = f.apply(raw.astype(np.float32))
    assert same_bits(y, g["ref_out_f32"])
    # `yourSavgolData` is the C filter's own output printed with 6 decimals
    assert np.max(np.abs(y.astype(np.float64) - theirs)) < 2e-5
    assert normwise(y, theirs) < 1e-6
    y64 = f.apply_f64(raw)
    assert normwise(y64, theirs) < 1e-6


def test_demo_dataset(sgo, golden):
    g = golden("demo360")
    ds = g["dataset"]
    assert ds.size == 360
    assert same_bits(sgo.Filter(6, 3, 0).apply(ds), g["smooth_n6_m3"])
    assert same_bits(sgo.Filter(10, 3, 1).apply(ds), g["deriv1_n10_m3"])


def test_fp64_oracle_tracks_fp32_reference(sgo, golden):
    # reference fp32 output sits ~1e-7..3e-7 (normwise) from the fp64-accumulate oracle
    g = golden("apply1d")
    for ci in (3, 4, 5, 6):
        n, m, d, _ = (int(v) for v in g[f"c{ci}_cfg"])
        f = sgo.Filter(n, m, d, float(g[f"c{ci}_dt"]))
        x = g[f"c{ci}_in"]
        for mode in range(4):
            hi = f.apply_f64(x.astype(np.float64), mode)
            assert normwise(g[f"c{ci}_mode{mode}_out"], hi) < 2e-6, (ci, mode)


def test_against_compiled_reference_when_present(sgo):
    """Random configs straight against oracle/_ref/libsavgol_ref.so (skipped if it did not travel)."""
    import ctypes as C
    import os
    from tests.golden import make_golden as mg
    if not os.path.exists(mg.LIB):
        pytest.skip("oracle/_ref/libsavgol_ref.so not built")
    L = mg.load()
    rng = np.random.default_rng(7)
    for _ in range(60):
        n = int(rng.integers(1, 33)); m = int(rng.integers(0, min(2 * n, 10) + 1)); d = int(rng.integers(0, min(m, 4) + 1))
        dt = float(np.float32(rng.choice([1.0, 0.5, 1e-3, 3.0])))
        mode = int(rng.integers(0, 4))
        length = int(rng.integers(2 * n + 1, 2 * n + 400))
        cfg = mg.Cfg(n, m, d, dt, mode)
        rf = L.savgol_create(C.byref(cfg))
        assert rf
        cw, ew = mg.filt_tables(rf)
        f = sgo.Filter(n, m, d, dt, mode)
        assert same_bits(cw, f.center) and same_bits(ew, f.edges)
        x = mg.signal(rng, length)
        y = np.zeros_like(x)
        assert L.savgol_apply(rf, mg.fptr(x), mg.fptr(y), length) == 0
        assert same_bits(f.apply(x), y), (n, m, d, mode, length)
        L.savgol_destroy(rf)
