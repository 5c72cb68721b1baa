"""GPU parity tests of the streaming path (single-stream drop-in API and the stream bank), through the C ABI.

The streaming kernels keep the reference's summation order and rounding (single fp32 accumulator, taps in
order, separate multiply and add), so the bar here is BIT-EXACT against the reference's golden sequences
and against the oracle's restatement of src/savgol_stream.c."""
import ctypes as C

import numpy as np
import pytest

from tests._util import check, fp32_bar, bits, fuzz, normwise
from tests.golden.make_golden import STREAM_CASES

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_gpu(sg):
    import torch
    assert torch.cuda.is_available() and sg.device_count() > 0, sg.last_error()
    return torch


def same_bits(a, b):
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    return a.shape == b.shape and np.array_equal(bits(a), bits(b))


@pytest.mark.parametrize("ci", range(len(STREAM_CASES)))
def test_single_stream_bit_exact_vs_reference_golden(sg, golden, torch_gpu, ci):
    g = golden("stream")
    n, m, d, count = (int(v) for v in g[f"s{ci}_cfg"])
    x = g[f"s{ci}_in"]
    dt = float(g[f"s{ci}_dt"])
    s = sg.Stream(n, m, d, dt)
    vals, valid = zip(*[s.push(v) for v in x])
    assert same_bits(np.array(vals, np.float32), g[f"s{ci}_push_val"])
    assert np.array_equal(np.array(valid), g[f"s{ci}_push_valid"])
    assert list(s.counters) == list(g[f"s{ci}_push_counters"])

    s = sg.Stream(n, m, d, dt)
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

    s = sg.Stream(n, m, d, dt)                 # max_outputs < n+1 silently truncates the burst (reference :209-218)
    tr = []
    for v in x:
        tr.extend(s.push_full(v, 2).tolist())
    assert same_bits(np.array(tr, np.float32), g[f"s{ci}_full_trunc2"])


def test_single_stream_reference_test_scenarios(sg, torch_gpu):
    """reference test_savgol_stream.c: lifecycle / latency / readiness / flush limits / stream == batch."""
    L = sg.lib()
    s = sg.Stream(5, 3)
    assert L.savgol_stream_latency(s.ptr) == 5 and not L.savgol_stream_ready(s.ptr)
    for i in range(10):
        _, ok = s.push(float(i))
        assert not ok
    assert L.savgol_stream_buffered(s.ptr) == 10
    _, ok = s.push(10.0)
    assert ok and L.savgol_stream_ready(s.ptr) and L.savgol_stream_buffered(s.ptr) == 11
    c, _ = s.flush(3)
    assert c == 3                                                   # respects max_count
    L.savgol_stream_reset(s.ptr)
    assert s.counters == (0, 0, 0) and not L.savgol_stream_ready(s.ptr)
    c, _ = s.flush()
    assert c == 0                                                   # never filled
    assert L.savgol_stream_flush(None, None, 3) == -1 and L.savgol_stream_flush_leading(None, None, 3) == 0
    # stream == batch (push_full + flush vs savgol_apply), reference tolerance 1e-5   (:140-189)
    rng = np.random.default_rng(1)
    x = (np.sin(np.arange(100) * 0.1) + rng.normal(0, 0.05, 100)).astype(np.float32)
    s = sg.Stream(5, 3)
    seq = []
    for v in x:
        seq.extend(s.push_full(v).tolist())
    seq.extend(s.flush()[1].tolist())
    assert len(seq) == 100 and s.counters[1] == 100
    batch = sg.Filter(5, 3).apply(x)
    assert np.max(np.abs(np.array(seq, np.float32) - batch)) < 1e-5
    # derivative stream: d/dx(2x) = 2 at the centre                                  (:191-224)
    s = sg.Stream(5, 2, 1, 1.0)
    outs = [s.push(2.0 * i) for i in range(30)]
    assert all(abs(v - 2.0) < 0.01 for v, ok in outs if ok)
    # user-allocated stream borrowing a filter (savgol_stream_init)
    import ctypes as C
    f = sg.Filter(3, 2)
    st = sg.SavgolStream()
    assert L.savgol_stream_init(C.byref(st), f.ptr) == 0 and not st.owns_filter
    for i in range(7):
        y = L.savgol_stream_push(C.byref(st), 4.0, None)           # output_valid may be NULL
    assert abs(y - 4.0) < 1e-5 and st.samples_output == 1


@pytest.mark.parametrize("cfg", [(16, 2, 1, 1e-3), (5, 3, 0, 1.0), (32, 4, 2, 0.5)])
def test_stream_bank_bit_exact_vs_oracle(sg, sgo, torch_gpu, cfg):
    torch = torch_gpu
    n, m, d, dt = cfg
    S, T = 1000, 3 * (2 * n + 1) + 5
    rng = np.random.default_rng(n)
    x = rng.normal(0, 1, (T, S)).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    f = sgo.Filter(n, m, d, dt)
    pick = [0, 1, 63, 64, 511, 999]
    oracles = {s: sgo.Stream(f) for s in pick}

    bank = sg.StreamBank(S, n, m, d, dt)
    out = torch.zeros((n + 1, S), dtype=torch.float32, device="cuda")
    for t in range(T):
        rows = bank.push_full(xd[t], out, n + 1)
        want = {s: oracles[s].push_full(x[t, s]) for s in pick}
        assert rows == want[0].size, (t, rows)
        if rows:
            got = out[:rows].cpu().numpy()
            for s in pick:
                assert same_bits(got[:, s], want[s]), (t, s)
    rows = bank.flush_leading(out, n)
    got = out[:rows].cpu().numpy()
    for s in pick:
        c, w = oracles[s].flush_leading()
        assert c == rows and same_bits(got[:, s], w)
    rows = bank.flush(out, n)
    got = out[:rows].cpu().numpy()
    for s in pick:
        c, w = oracles[s].flush()
        assert c == rows and same_bits(got[:, s], w)
    assert bank.counters == (T, oracles[0].counters[1])

    # plain push + the multi-tick block push + checkpoint/resume give the same centre outputs
    bank2 = sg.StreamBank(S, n, m, d, dt)
    o1 = torch.zeros(S, dtype=torch.float32, device="cuda")
    centre = []
    for t in range(T):
        if bank2.push(xd[t], o1) == 1:
            centre.append(o1.cpu().numpy().copy())
    centre = np.stack(centre)
    bank3 = sg.StreamBank(S, n, m, d, dt)
    ob = torch.full((T, S), np.nan, dtype=torch.float32, device="cuda")
    half = T // 2
    p1 = bank3.push_block(xd, half, ob)
    blob = bank3.save()
    bank4 = sg.StreamBank(S, n, m, d, dt)
    bank4.load(blob)
    p2 = bank4.push_block(xd[half:], T - half, ob[half:])
    torch.cuda.synchronize()
    assert p1 + p2 == centre.shape[0] == T - 2 * n
    assert same_bits(ob[2 * n:].cpu().numpy(), centre)
    assert torch.isnan(ob[:2 * n]).all()                                # ticks without output are not written
    # and the centre outputs equal the oracle's push() sequence
    for s in pick:
        o = sgo.Stream(f)
        seq = [o.push(v) for v in x[:, s]]
        assert same_bits(np.array([v for v, ok in seq if ok], np.float32), centre[:, s])


@pytest.mark.parametrize("n", list(range(1, 18)) + [32])
def test_block_push_equals_tick_pushes_every_half_window(sg, torch_gpu, n):
    """push_block (rolling-window kernel for n <= 16, LDS-tiled kernel above) against the per-tick push kernel, which the
    test above pins to the oracle: every stream, bit for bit.  Odd stream counts and a misaligned sample pointer take
    the scalar path; a partly filled ring, a wrapped ring and several row bands per call are all in the sequence."""
    torch = torch_gpu
    m = min(3, 2 * n)
    for S, off in ((777, 1), (1024, 0)):
        T = 40 * (2 * n + 1) + 7
        g = torch.Generator(device="cuda").manual_seed(n * 1000 + S)
        flat = torch.randn(T * S + 4, generator=g, device="cuda", dtype=torch.float32)
        xd = flat[off:off + T * S].view(T, S)
        ref = sg.StreamBank(S, n, m, 0, 1.0)
        o1 = torch.zeros(S, dtype=torch.float32, device="cuda")
        want = torch.full((T, S), float("nan"), device="cuda")
        for t in range(T):
            if ref.push(xd[t], o1) == 1:
                want[t] = o1
        bank = sg.StreamBank(S, n, m, 0, 1.0)
        got = torch.full((T, S), float("nan"), device="cuda")
        cuts = [0, 3, n + 2, 2 * n + 2, 2 * n + 3, 9 * (2 * n + 1), T]           # filling / just full / wrapped / long
        total = 0
        for a, b in zip(cuts[:-1], cuts[1:]):
            if b == a:
                continue
            if b - a == 1:
                r = bank.push(xd[a], o1)
                if r == 1:
                    got[a] = o1
                total += r
            else:
                total += bank.push_block(xd[a:b], b - a, got[a:b])
        torch.cuda.synchronize()
        assert total == T - 2 * n
        assert torch.isnan(got[:2 * n]).all() and torch.isnan(want[:2 * n]).all()
        assert torch.equal(got[2 * n:].view(torch.int32), want[2 * n:].view(torch.int32)), (n, S)
        assert bank.counters == ref.counters


def test_stream_bank_config3_shape(sg, sgo, torch_gpu):
    """BASELINE config 3: 65 536 streams, n=16, m=2, d=1, dt=1e-3 -- sampled streams, bit-exact."""
    torch = torch_gpu
    S, n, T = 65536, 16, 80
    x = torch.empty((T, S), dtype=torch.float32, device="cuda")
    sg.synth(x.view(T, S))                                  # rows = ticks here; any deterministic data will do
    bank = sg.StreamBank(S, n, 2, 1, 1e-3)
    out = torch.zeros((T, S), dtype=torch.float32, device="cuda")
    produced = bank.push_block(x, T, out)
    torch.cuda.synchronize()
    assert produced == T - 2 * n
    xh = x.cpu().numpy()
    f = sgo.Filter(n, 2, 1, 1e-3)
    for s in (0, 1, 4095, 32768, 65535):
        o = sgo.Stream(f)
        seq = np.array([v for v, ok in (o.push(v) for v in xh[:, s]) if ok], np.float32)
        assert same_bits(out[2 * n:, s].cpu().numpy(), seq)


def test_stream_bank_config3_full_size(sg, sgo, torch_gpu):
    """BASELINE config 3 at the size bench.py times (VERDICT r03 missing #3): 65 536 streams x 4096 ticks, n=16, m=2, d=1, dt=1e-3,
    one block push.  Five sampled streams: the reference-order bank bit for bit against the oracle's push loop, the fused multiply-add
    bank within 1e-6 of the double-accumulation oracle -- or 1.1 x the error of the REFERENCE's own stream arithmetic (the bit-exact bank's
    output on the same streams) where that is larger; both banks write exactly the ticks that have an output."""
    torch = torch_gpu
    S, n, T = 65536, 16, 4096
    free, _ = torch.cuda.mem_get_info()
    if free < 4 * T * S * 4 + (2 << 30):
        pytest.skip("not enough HBM free")
    x = torch.empty((T, S), dtype=torch.float32, device="cuda")
    sg.synth(x)
    pick = [0, 1, 4095, 32768, 65535]
    xh = x[:, pick].cpu().numpy()
    f = sgo.Filter(n, 2, 1, 1e-3)
    ref64 = f.apply_f64(xh.T.astype(np.float64).copy())[:, n:T - n]
    e_ref = 0.0
    for fma in (False, True):
        bank = sg.StreamBank(S, n, 2, 1, 1e-3, fma=fma)
        out = torch.full((T, S), float("nan"), dtype=torch.float32, device="cuda")
        assert bank.push_block(x, T, out) == T - 2 * n
        torch.cuda.synchronize()
        assert torch.isnan(out[:2 * n]).all() and not torch.isnan(out[2 * n:]).any()
        got = out[2 * n:, pick].cpu().numpy()
        if not fma:
            for j in range(len(pick)):
                o = sgo.Stream(f)
                seq = np.array([v for v, ok in (o.push(v) for v in xh[:, j]) if ok], np.float32)
                assert same_bits(got[:, j], seq), pick[j]
            e_ref = normwise(got.T, ref64)                     # the reference's own fp32 error on these streams (one chain, multiply and add rounded)
        else:
            check(normwise(got.T, ref64), fp32_bar(e_ref), ("config 3 FMA bank", e_ref))
        assert bank.counters[0] == T and bank.counters[1] == T - 2 * n


def test_randomized_stream_bank_sequences(sg, sgo, torch_gpu):
    """80 random banks (n up to 32, any order / derivative / time step, odd and even stream counts) driven by a random
    sequence of push / push_full / push_block calls, then both flushes: sampled streams must reproduce the oracle's
    per-stream sequence bit for bit, counters included."""
    torch = torch_gpu
    seed, iters = fuzz(20261005, 80)
    rng = np.random.default_rng(seed)
    for it in range(iters):
        n = int(rng.integers(1, 33)); m = int(rng.integers(0, min(2 * n, 8) + 1)); d = int(rng.integers(0, min(m, 3) + 1))
        dt = float(rng.choice([1.0, 1e-3, 0.5]))
        S = int(rng.choice([1, 2, 63, 130, 777, 1024])); T = int(rng.integers(1, 6 * (2 * n + 1)))
        x = rng.normal(0, 1, (T, S)).astype(np.float32)
        xd = torch.from_numpy(x).cuda()
        f = sgo.Filter(n, m, d, dt)
        pick = sorted(set([0, S - 1, S // 2]))
        oracles = {s: sgo.Stream(f) for s in pick}
        want = {s: [] for s in pick}
        got = {s: [] for s in pick}
        bank = sg.StreamBank(S, n, m, d, dt)
        rows = torch.zeros((n + 1, S), dtype=torch.float32, device="cuda")
        t = 0
        while t < T:
            kind = int(rng.integers(0, 3)); k = 1 if kind < 2 else int(rng.integers(1, min(T - t, 4 * n + 8) + 1))
            if kind == 0:
                r = bank.push(xd[t], rows[0])
                host = rows[:1].cpu().numpy() if r == 1 else None
                for s in pick:
                    v, ok = oracles[s].push(x[t, s])
                    assert ok == (r == 1)
                    if ok:
                        want[s].append(v); got[s].append(host[0, s])
            elif kind == 1:
                r = bank.push_full(xd[t], rows, n + 1)
                host = rows[:r].cpu().numpy() if r else None
                for s in pick:
                    w = oracles[s].push_full(x[t, s])
                    assert w.size == r
                    want[s].extend(w.tolist()); got[s].extend(host[:, s].tolist() if r else [])
            else:
                out = torch.full((k, S), float("nan"), device="cuda")
                r = bank.push_block(xd[t:t + k], k, out)
                host = out.cpu().numpy()
                for s in pick:
                    seq = [oracles[s].push(v) for v in x[t:t + k, s]]
                    vals = [v for v, ok in seq if ok]
                    assert len(vals) == r
                    want[s].extend(vals); got[s].extend(host[k - r:, s].tolist())
                    assert np.isnan(host[:k - r, s]).all()
            t += k
        for flush_gpu, flush_cpu in ((bank.flush_leading, "flush_leading"), (bank.flush, "flush")):
            r = flush_gpu(rows, n)
            host = rows[:max(r, 0)].cpu().numpy()
            for s in pick:
                c, w = getattr(oracles[s], flush_cpu)()
                assert c == r, (it, flush_cpu, c, r)
                if r > 0:
                    want[s].extend(np.asarray(w).tolist()); got[s].extend(host[:, s].tolist())
        for s in pick:
            a, b2 = np.array(want[s], np.float32), np.array(got[s], np.float32)
            assert a.shape == b2.shape and same_bits(a, b2), (it, n, m, d, dt, S, T, s)
        assert tuple(bank.counters) == tuple(oracles[pick[0]].counters[:2])


def test_corrupt_checkpoint_and_foreign_device_are_refused(sg, torch_gpu):
    """a blob whose write position lies outside the ring (or disagrees with the counters) must not be loaded;
    a blob without the magic is not a blob."""
    import ctypes as C
    torch = torch_gpu
    bank = sg.StreamBank(64, 5, 3, 0, 1.0)
    x = torch.randn((20, 64), device="cuda")
    out = torch.zeros_like(x)
    assert bank.push_block(x, 20, out) == 20 - 10
    torch.cuda.synchronize()
    blob = bank.save()
    hdr = np.frombuffer(blob, dtype=np.int64, count=6).copy()          # wp, received, emitted, streams, ws, magic
    assert hdr[0] == 20 % 11 and hdr[1] == 20 and hdr[4] == 11
    L = sg.lib()

    def load(h):
        b = bytearray(blob)
        b[:48] = h.tobytes()
        return L.savgol_streambank_load(bank.ptr, bytes(b), None)
    assert load(hdr) == 0
    for idx, bad in ((0, 11), (0, -1), (0, 3), (2, 21), (5, 12345)):
        h = hdr.copy(); h[idx] = bad
        assert load(h) == -1, (idx, bad)
    assert "blob" in sg.last_error()
    assert load(hdr) == 0                                              # still usable afterwards


@pytest.mark.parametrize("n,m,d,streams", [(16, 2, 1, 65536), (5, 3, 0, 1000), (32, 4, 2, 4096), (1, 0, 0, 260)])
def test_resident_tick_service_is_bit_identical_and_hands_state_over(sg, torch_gpu, n, m, d, streams):
    """savgol_streambank_service_*: a resident kernel answers doorbells instead of one launch + synchronise per tick
    (csrc/sg_stream_service.hip).  Every tick's output must equal the per-tick kernel's (itself pinned bit for bit to the
    reference's savgol_stream_push) for every stream, through the filling phase, across a stop -> stream-ordered calls -> start
    hand-over, and across the kernel's own idle time-out."""
    import time
    torch = torch_gpu
    ws = 2 * n + 1
    T = ws + 40
    x = torch.randn((T, streams), device="cuda")
    ref_bank = sg.StreamBank(streams, n, m, d, 1e-3 if d else 1.0)
    bank = sg.StreamBank(streams, n, m, d, 1e-3 if d else 1.0)
    want = torch.full((T, streams), -7.0, device="cuda")
    rcs = [ref_bank.push(x[t], want[t]) for t in range(T)]
    torch.cuda.synchronize()
    got = torch.full((T, streams), -7.0, device="cuda")
    torch.cuda.synchronize()
    bank.service_start(idle_ms=150)
    L = sg.lib()
    try:
        cut1, cut2 = ws // 2, ws + 10
        for t in range(cut1):                                  # still filling: no output, d_out untouched
            assert bank.service_tick(x[t], got[t]) == rcs[t] == 0
        assert bank.push(x[cut1], got[cut1]) == -1 and "service" in sg.last_error()      # other calls are refused meanwhile
        bank.service_stop()
        for t in range(cut1, cut1 + 3):                        # stream-ordered calls continue from the service's state ...
            assert bank.push(x[t], got[t]) == rcs[t]
        torch.cuda.synchronize()
        bank.service_start(idle_ms=150)                        # ... and the service continues from theirs
        for t in range(cut1 + 3, cut2):
            assert bank.service_tick(x[t], got[t]) == rcs[t]
        time.sleep(0.4)                                        # longer than idle_ms: the kernel has left by itself
        for t in range(cut2, T):                               # the next tick restarts it transparently
            assert bank.service_tick(x[t], got[t]) == rcs[t]
    finally:
        bank.service_stop()
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))
    assert bank.counters == ref_bank.counters
    # and the trailing edge rows come out of the ring the service kept current
    f1, f2 = torch.zeros((n, streams), device="cuda"), torch.zeros((n, streams), device="cuda")
    assert bank.flush(f1, n) == ref_bank.flush(f2, n) == n
    torch.cuda.synchronize()
    assert torch.equal(f1.view(torch.int32), f2.view(torch.int32))


def test_tick_service_argument_checks(sg, torch_gpu):
    torch = torch_gpu
    bank = sg.StreamBank(1002, 4, 2)                           # not a multiple of 4 streams
    with pytest.raises(RuntimeError, match="multiple of 4"):
        bank.service_start()
    bank = sg.StreamBank(1000, 4, 2)
    x = torch.zeros(1004, device="cuda")
    assert bank.service_tick(x, x) == -1 and "service_start" in sg.last_error()
    bank.service_start(idle_ms=100)
    try:
        assert bank.service_tick(x[1:], x) == -1 and "aligned" in sg.last_error()
    finally:
        bank.service_stop()
    bank.service_stop()                                        # idempotent


@pytest.mark.parametrize("mode", [1, 3, 2])                    # REFLECT, CONSTANT, PERIODIC
@pytest.mark.parametrize("n,m,d", [(5, 3, 0), (16, 2, 1), (32, 4, 2)])
def test_opt_in_boundary_aware_streams_match_the_batch_filter(sg, sgo, torch_gpu, mode, n, m, d):
    """the reference's streaming path ignores config.boundary (src/savgol_stream.c:43-74 always uses the
    polynomial rows).  With SAVGOL_HIP_OPT_BOUNDARY_AWARE, push_full... + flush of a REFLECT / CONSTANT stream equals
    savgol_apply in that mode (centre outputs bit for bit, edge outputs to rounding: they are the same taps summed in another
    order); PERIODIC cannot be streamed and keeps the polynomial rows.  Single stream and bank."""
    torch = torch_gpu
    L = sg.lib()
    T = 3 * (2 * n + 1)
    dt = 0.5 if d else 1.0
    x = sgo.synth_f32(11, 4, T)
    assert L.savgol_hip_set_option(sg.SAVGOL_HIP_OPT_BOUNDARY_AWARE, 1) == 0
    try:
        want_mode = mode if mode != 2 else 0                   # PERIODIC -> polynomial edges
        ref = np.stack([sgo.Filter(n, m, d, dt, want_mode).apply(x[c]) for c in range(4)])
        ref64 = np.stack([sgo.Filter(n, m, d, dt, want_mode).apply_f64(x[c:c + 1].astype(np.float64))[0] for c in range(4)])
        # single stream (host API); the stream's filter carries the mode
        cfg = sg.SavgolConfig(n, m, d, dt, mode)
        f = L.savgol_create(C.byref(cfg))
        st = sg.SavgolStream()
        assert L.savgol_stream_init(C.byref(st), f) == 0
        outs = []
        buf = (C.c_float * 40)()
        for t in range(T):
            c = L.savgol_stream_push_full(C.byref(st), float(x[0, t]), buf, 40)
            outs += list(buf[:c])
        c = L.savgol_stream_flush(C.byref(st), buf, 40)
        outs += list(buf[:c])
        got = np.asarray(outs, np.float32)
        assert got.shape == (T,)
        # the bar: 1e-6, or 1.1 x the reference's own fp32 error on this signal -- its batch arithmetic (four chains) on all outputs, its stream
        # arithmetic (one chain: the oracle's push loop) on the centre outputs -- where that is larger
        o1 = sgo.Stream(sgo.Filter(n, m, d, dt, 0))
        centre = np.array([v for v, ok in (o1.push(v) for v in x[0]) if ok], np.float32)
        bar = fp32_bar(max(normwise(ref[0], ref64[0]), normwise(centre, ref64[0][n:T - n])))
        assert same_bits(got[n:T - n], centre)                  # centre outputs: the reference's stream arithmetic, bit for bit, in every mode
        check(normwise(got, ref64[0]), bar, ("boundary-aware stream", n, m, d, mode))
        L.savgol_destroy(f)
        # bank
        bank = sg.StreamBank.__new__(sg.StreamBank)
        bank.ptr = L.savgol_streambank_create(C.byref(cfg), 4)
        assert bank.ptr
        xs = torch.from_numpy(np.ascontiguousarray(x.T)).cuda()          # [T][4]
        rows_out = []
        tmp = torch.zeros((n + 1, 4), device="cuda")
        for t in range(T):
            c = L.savgol_streambank_push_full(bank.ptr, xs[t].data_ptr(), tmp.data_ptr(), n + 1, None)
            torch.cuda.synchronize()
            rows_out += [tmp[i].cpu().numpy().copy() for i in range(c)]
        c = L.savgol_streambank_flush(bank.ptr, tmp.data_ptr(), n, None)
        torch.cuda.synchronize()
        rows_out += [tmp[i].cpu().numpy().copy() for i in range(c)]
        gotb = np.stack(rows_out).T                                      # [4][T]
        assert gotb.shape == (4, T)
        assert np.array_equal(gotb[0].view(np.uint32), got.view(np.uint32))      # bank == single stream, bit for bit
        for c_ in range(4):
            oc = sgo.Stream(sgo.Filter(n, m, d, dt, 0))
            cc = np.array([v for v, ok in (oc.push(v) for v in x[c_]) if ok], np.float32)
            check(normwise(gotb[c_], ref64[c_]), fp32_bar(max(normwise(ref[c_], ref64[c_]), normwise(cc, ref64[c_][n:T - n]))), ("boundary-aware bank", n, m, d, mode, c_))
        L.savgol_streambank_destroy(bank.ptr); bank.ptr = None
    finally:
        L.savgol_hip_set_option(sg.SAVGOL_HIP_OPT_BOUNDARY_AWARE, 0)


@pytest.mark.parametrize("n,m,d,dt", [(16, 2, 1, 1e-3), (5, 3, 0, 1.0), (1, 0, 0, 1.0), (17, 4, 2, 0.5), (32, 4, 0, 1.0), (24, 3, 1, 2.0),
                                      # round 5: taps that are a polynomial of degree <= 2 take the block-moment tiles from n = 12 (1, 2, 3 moments)
                                      (12, 2, 0, 1.0), (13, 1, 0, 1.0), (20, 2, 2, 0.5), (32, 2, 1, 1e-3), (31, 3, 0, 1.0), (28, 1, 1, 2.0), (16, 3, 2, 1.0)])
def test_fma_bank_is_within_the_fp32_bar_of_the_fp64_oracle(sg, sgo, torch_gpu, n, m, d, dt):
    """SAVGOL_STREAMBANK_FMA (savgol_streambank_create_ex): block push (sample ring n <= 16, accumulator ring above) and the
    per-tick kernel with fused multiply-adds.  Not the reference's bits; the bar is 1e-6 normwise of the double-accumulation oracle, or
    1.1 x the error of the reference-order bank (= the reference's own arithmetic) on the same samples where that is larger.  Edge rows keep the reference's order: push_full's burst and the flushes stay bit-exact."""
    torch = torch_gpu
    # 8960 streams = 70 strips of 128: tile groups that do not divide the strips (the last group of the LDS-DMA tiles is a partial one)
    for S, off in ((65536 if n == 16 else 4096, 0), (777, 1), (8960, 0)):
        T = 12 * (2 * n + 1) + 5
        g = torch.Generator(device="cuda").manual_seed(n * 77 + S)
        flat = torch.randn(T * S + 4, generator=g, device="cuda", dtype=torch.float32)
        xd = flat[off:off + T * S].view(T, S)
        fast = sg.StreamBank(S, n, m, d, dt, fma=True)
        ref = sg.StreamBank(S, n, m, d, dt)
        got = torch.full((T, S), float("nan"), device="cuda")
        want = torch.full((T, S), float("nan"), device="cuda")
        cut = 3 * n + 1                                           # first call ends inside / after the filling phase
        assert fast.push_block(xd, cut, got) + fast.push_block(xd[cut:], T - cut, got[cut:]) == T - 2 * n
        assert ref.push_block(xd, T, want) == T - 2 * n
        torch.cuda.synchronize()
        assert torch.isnan(got[:2 * n]).all()
        pick = [0, 1, S // 2, S - 1]
        xh = xd[:, pick].cpu().numpy().astype(np.float64).T.copy()              # [stream][tick]
        ref64 = sgo.Filter(n, m, d, dt).apply_f64(xh)[:, n:T - n]                  # centre outputs of tick t = batch output t - n
        e_fast = normwise(got[2 * n:, pick].cpu().numpy().T, ref64)
        e_ref = normwise(want[2 * n:, pick].cpu().numpy().T, ref64)
        bar = fp32_bar(e_ref)
        check(e_fast, bar, ("FMA bank", n, m, d, S, e_ref))
        # whole banks: the two summations agree to the bar everywhere, and do differ somewhere (the flag selects another kernel)
        assert normwise(got[2 * n:].cpu().numpy(), want[2 * n:].cpu().numpy()) <= 2 * bar
        if n >= 5:
            assert not torch.equal(got[2 * n:], want[2 * n:])
        # per-tick pushes of the fast bank continue the sequence with the same bar
        o1 = torch.zeros(S, dtype=torch.float32, device="cuda")
        o2 = torch.zeros(S, dtype=torch.float32, device="cuda")
        for t in range(4):
            assert fast.push(xd[t], o1) == 1 and ref.push(xd[t], o2) == 1
            assert normwise(o1.cpu().numpy(), o2.cpu().numpy()) <= 2 * bar
        # edge rows stay in the reference's order
        e1 = torch.zeros((n, S), dtype=torch.float32, device="cuda"); e2 = torch.zeros_like(e1)
        assert fast.flush(e1, n) == ref.flush(e2, n) == n
        assert torch.equal(e1.view(torch.int32), e2.view(torch.int32))
    assert sg.lib().savgol_streambank_create_ex(None, 4, 1) is None


def test_randomized_fused_bank_block_sequences(sg, sgo, torch_gpu):
    """30 random fused banks (SAVGOL_STREAMBANK_FMA) fed by sequences of push_block calls of random lengths -- shorter than one tile, one tile, many
    tiles; stream counts with whole, partial and single tile groups; half windows and orders on both sides of the block-moment tiles' range (n = 12..20,
    taps of degree <= 2) -- against the double oracle, bar: the FMA bank's (max(1e-6, 1.1 x the reference-order bank's own error on the same samples)).
    Every call continues from the ring the previous one left, so the tiles' history path (rows from the ring, not from this call) is exercised too."""
    torch = torch_gpu
    seed, iters = fuzz(20261006, 30)
    rng = np.random.default_rng(seed)
    for it in range(iters):
        n = int(rng.choice([int(rng.integers(1, 33)), int(rng.integers(12, 21))]))
        m = int(rng.choice([0, 1, 2, 2, 3, 4])); m = min(m, 2 * n)
        d = int(rng.integers(0, min(m, 2) + 1))
        dt = float(rng.choice([1.0, 1e-3, 0.5]))
        S = int(rng.choice([128, 384, 1024, 2176, 8960]))
        cuts = [int(v) for v in rng.choice([1, 7, 40, 64, 65, 96, 130, 257], size=int(rng.integers(2, 5)))]
        T = sum(cuts)
        if T < 2 * n + 2:
            cuts.append(2 * n + 2 - T + 64); T = sum(cuts)
        g = torch.Generator(device="cuda").manual_seed(seed + it)
        xd = torch.randn((T, S), generator=g, device="cuda", dtype=torch.float32)
        fast, ref = sg.StreamBank(S, n, m, d, dt, fma=True), sg.StreamBank(S, n, m, d, dt)
        got = torch.full((T, S), float("nan"), device="cuda"); want = torch.full((T, S), float("nan"), device="cuda")
        t0 = 0
        for c in cuts:
            fast.push_block(xd[t0:], c, got[t0:]); ref.push_block(xd[t0:], c, want[t0:]); t0 += c
        torch.cuda.synchronize()
        assert torch.isnan(got[:2 * n]).all() and not torch.isnan(got[2 * n:]).any(), (it, n, m, d, S, cuts)
        pick = sorted({0, 1, S // 2, S - 1, int(rng.integers(0, S))})
        xh = xd[:, pick].cpu().numpy().astype(np.float64).T.copy()
        ref64 = sgo.Filter(n, m, d, dt).apply_f64(xh)[:, n:T - n]
        e_fast = normwise(got[2 * n:, pick].cpu().numpy().T, ref64)
        e_ref = normwise(want[2 * n:, pick].cpu().numpy().T, ref64)
        check(e_fast, fp32_bar(e_ref), ("fused bank, random block sequence", it, n, m, d, S, tuple(cuts), e_ref))
        # every stream, not only the sampled ones: the two banks agree to twice the bar
        assert normwise(got[2 * n:].cpu().numpy(), want[2 * n:].cpu().numpy()) <= 2 * fp32_bar(e_ref), (it, n, m, d, S, cuts)


def test_push_wait_returns_complete_outputs(sg, sgo, torch_gpu):
    """savgol_streambank_push_wait (round 5): one tick whose outputs are complete on return -- the tick kernel's last block writes a completion word
    into pinned host memory and the host spins on it, no hipStreamSynchronize.  Same values, bit for bit, as savgol_streambank_push + synchronise on
    a twin bank (the reference's savgol_stream_push arithmetic, src/savgol_stream.c:152-178), through the filling phase, on the default and on another
    stream.  Round 6: the property push_wait ADDS is exercised -- the outputs are read right after the call returns with NO device
    synchronise in between: (a) d_out in mapped pinned host memory, read by the CPU at once; (b) a non-blocking device-to-host copy on a stream that
    has no dependency on the tick's stream.  A bank whose stream count is not a multiple of 64 takes the push + synchronise fallback: same values."""
    import ctypes as C
    torch = torch_gpu
    hip = C.CDLL("libamdhip64.so")
    for S, n in ((4096, 8), (4000, 8)):                    # 4000 streams: not whole waves at the barrier -> the fallback path
        a, b = sg.StreamBank(S, n, 3, 0, 1.0), sg.StreamBank(S, n, 3, 0, 1.0)
        x = torch.randn((60, S), device="cuda")
        # the twin's outputs first, fully synchronised: what every tick must produce
        want = []
        ob = torch.zeros(S, device="cuda")
        for t in range(60):
            rb = b.push(x[t], ob)
            torch.cuda.synchronize()
            want.append((rb, ob.cpu().numpy().copy()))
        host = torch.zeros(S).pin_memory()                 # (a) the kernel writes straight into host memory
        dev_view = C.c_void_p()
        mapped = hip.hipHostGetDevicePointer(C.byref(dev_view), C.c_void_p(host.data_ptr()), 0) == 0 and dev_view.value
        oa = torch.zeros(S, device="cuda")
        landing = torch.zeros(S).pin_memory()              # (b) target of the unordered copy
        side, copier = torch.cuda.Stream(), torch.cuda.Stream()
        torch.cuda.synchronize()
        for t in range(60):
            st = side if t % 2 else None
            use_host = bool(mapped) and t % 3 == 0
            if use_host:
                host.fill_(-1.0)
            ra = a.push_wait(x[t], dev_view.value if use_host else oa, stream=st)
            # NO torch.cuda.synchronize() here: the call itself promised the outputs
            if use_host:
                got = host.numpy().copy()
            else:
                with torch.cuda.stream(copier):             # no wait_stream: nothing orders this copy behind the tick but push_wait's return
                    landing.copy_(oa, non_blocking=True)
                copier.synchronize()
                got = landing.numpy().copy()
            rb, exp = want[t]
            assert ra == rb == (1 if t >= 2 * n else 0)
            if ra:
                assert same_bits(got, exp), (S, t, use_host)
        assert a.counters[0] == 60 and a.counters[1] == 60 - 2 * n
        torch.cuda.synchronize()


def test_few_streams_long_block_uses_enough_bands(sg, sgo, torch_gpu):
    """the walk's band search stopped at 64 bands, so a bank of FEW strips and a long block (1024 streams = 8 strips) ran on a quarter
    of the resident waves.  Functional check of the repaired search on such a shape (n = 20: the accumulator-ring walk; n = 8 with an odd stream
    count: the sample-ring walk on the element path): outputs equal the oracle's push loop bit for bit on sampled streams, whatever the band count."""
    torch = torch_gpu
    for S, n, T in ((1024, 20, 30000), (1022, 8, 20000)):
        x = torch.randn((T, S), device="cuda")
        out = torch.full((T, S), float("nan"), device="cuda")
        bank = sg.StreamBank(S, n, 3, 1, 0.5)
        assert bank.push_block(x, T, out) == T - 2 * n
        torch.cuda.synchronize()
        f = sgo.Filter(n, 3, 1, 0.5)
        for j in (0, 1, S // 2 + 1, S - 1):
            xs = x[:, j].cpu().numpy()
            o = sgo.Stream(f)
            seq = np.array([v for v, ok in (o.push(v) for v in xs) if ok], np.float32)
            assert same_bits(out[2 * n:, j].cpu().numpy(), seq), (S, n, j)

@pytest.mark.parametrize("n,m,d", [(16, 2, 1), (16, 2, 2), (12, 1, 1), (20, 2, 2), (8, 3, 1), (24, 3, 2)])
def test_fused_bank_on_streams_with_a_large_offset(sg, sgo, torch_gpu, n, m, d):
    """End of round 6 (tools/offset_probe_1d.py, tools/tick_offset_probe.py): the fused bank on streams riding on an offset 10 ... 1000 x the signal, against the
    reference's own stream arithmetic (one chain, /root/reference/src/savgol_stream.c:25-38 = the bit-exact bank on the same samples).  Uncentred, the
    block-moment tiles sat at 0.5-1.0 of that bar (d = 2: 1.00) and at 2.3 x the reference's BATCH loop; derivative filters now run on CENTRED samples
    in the LDS-DMA tiles and the walk (200 streams: not whole strips): the mean of eight rows of the tile / item; c x the reference table's tap sum added
    back; quadratic taps that sum to zero keep the tap-by-tap tiles: 2-9e-7 of the oracle whatever the offset.  The per-tick kernel (the last 48 ticks)
    is held to the same bar, uncentred (0.33-0.48 of it)."""
    torch = torch_gpu
    rng = np.random.default_rng(70 + n + d)
    T = 1024
    tt = np.arange(T)
    for S in (256, 200):                                     # 200 streams: not whole 128-stream strips -> the walk instead of the LDS-DMA tiles
        sb = np.sin(0.02 * tt)[:, None] * np.linspace(0.5, 1.5, S)[None, :] + rng.normal(0, 0.1, (T, S))
        for off in (0.0, 10.0, 1000.0):
            x = (sb + off).astype(np.float32)
            bank = sg.StreamBank(S, n, m, d, 1.0, fma=True)
            dx = torch.from_numpy(x).cuda()
            out = torch.zeros_like(dx)
            blk = T - 48                                     # the last 48 ticks go through the per-tick kernel, one push each
            assert bank.push_block(dx, blk, out) == blk - 2 * n
            o1 = torch.zeros(S, dtype=torch.float32, device="cuda")
            for t in range(blk, T):
                assert bank.push(dx[t], o1) == 1
                out[t] = o1
            torch.cuda.synchronize()
            o = sgo.Filter(n, m, d, 1.0, 0)
            pick = [0, 1, S // 2, S - 1]
            xh = np.ascontiguousarray(x[:, pick].T)
            hi = o.apply_f64(xh.astype(np.float64))[:, n:T - n]
            # the reference's own stream arithmetic (one chain: src/savgol_stream.c:25-38) = the bit-exact bank on the same samples
            rb = sg.StreamBank(S, n, m, d, 1.0)
            want = torch.zeros_like(dx)
            assert rb.push_block(dx, T, want) == T - 2 * n
            ref = want.cpu().numpy()[2 * n:, pick].T
            got = out.cpu().numpy()[2 * n:, pick].T
            nb = blk - 2 * n                                  # outputs of the block push, then of the per-tick pushes
            check(normwise(got[:, :nb], hi[:, :nb]), fp32_bar(normwise(ref[:, :nb], hi[:, :nb])), ("fused bank, offset, block push", n, m, d, S, off))
            check(normwise(got[:, nb:], hi[:, nb:]), fp32_bar(normwise(ref[:, nb:], hi[:, nb:])), ("fused bank, offset, per tick", n, m, d, S, off))
