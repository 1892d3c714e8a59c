"""GPU parity of the NTT kernels (through the C ABI) against the CPU oracle."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu

Q = O.PRIME


def _root_for(q, d):
    """a primitive 2d-th root of unity mod q (smallest generator power found)."""
    for g in range(2, 2000):
        r = pow(g, (q - 1) // (2 * d), q)
        if pow(r, d, q) == q - 1:
            return r
    raise AssertionError("no root")


def _edge_rows(d, q):
    h = (q - 1) // 2
    rows = [np.zeros(d, np.int64), np.eye(1, d, 0, dtype=np.int64)[0], np.eye(1, d, 1, dtype=np.int64)[0],
            np.eye(1, d, d - 1, dtype=np.int64)[0], np.full(d, h), np.full(d, -h),
            np.where(np.arange(d) % 2 == 0, h, -h),
            # non-centred int32 inputs: __neg__-style [-(q-1), 0], raw extremes
            -np.arange(d) * ((q - 1) // d), np.full(d, -(q - 1)), np.full(d, 2**31 - 1), np.full(d, -2**31),
            np.where(np.arange(d) % 3 == 0, 2**31 - 1, -2**31),
            # the inverse transforms' fast last stage applies to inputs within 2^30: exactly at the bound (the all-add
            # path then reaches 2^38, the 4-op multiply's limit), alternating signs, and just outside it
            np.full(d, 2**30), np.full(d, -2**30), np.where(np.arange(d) % 2 == 0, 2**30, -2**30),
            np.where(np.arange(d) % 5 == 0, -2**30, 2**30), np.full(d, 2**30 + 1), np.full(d, -2**30 - 1)]
    return np.stack(rows).astype(np.int32)


@pytest.mark.parametrize("d", [2, 4, 8, 16, 32, 64, 128, 256])
def test_ntt_matches_oracle_prime(d, coracle):
    import fusion_hip
    root = {64: 23584283, 256: 3337519}.get(d) or _root_for(Q, d)
    inv = pow(root, Q - 2, Q)
    ctx = fusion_hip.Context(Q, d, root, inv)
    x = np.concatenate([_edge_rows(d, Q), O.splitmix_centered(11 + d, 37 * d).reshape(37, d)])
    assert np.array_equal(ctx.ntt_forward(x), coracle.ntt_forward(x, Q, root))
    assert np.array_equal(ctx.ntt_inverse(x), coracle.ntt_inverse(x, Q, inv))
    f, i = ctx.twiddles()
    assert np.array_equal(f.astype(np.int64), coracle.twiddles(root, Q, d))
    assert np.array_equal(i.astype(np.int64), coracle.twiddles(inv, Q, d))


@pytest.mark.parametrize("q,d", [(17, 8), (97, 16), (257, 64), (12289, 256), (65537, 128), (7681, 32), (5, 2)])
def test_ntt_small_primes(q, d, coracle):
    import fusion_hip
    root = _root_for(q, d)
    inv = pow(root, q - 2, q)
    ctx = fusion_hip.Context(q, d, root, inv)
    rng = np.random.default_rng(q * 1000 + d)
    x = rng.integers(-2**31, 2**31, size=(23, d), dtype=np.int64).astype(np.int32)
    assert np.array_equal(ctx.ntt_forward(x), coracle.ntt_forward(x, q, root))
    assert np.array_equal(ctx.ntt_inverse(x), coracle.ntt_inverse(x, q, inv))


@pytest.mark.parametrize("secpar", [128, 256])
def test_ntt_roundtrip_large_batch(secpar, coracle):
    """size-independent property at BASELINE size: INTT(NTT(x)) == x for centred x, and
    row-sampled equality with the oracle."""
    import fusion_hip
    P = O.PARAMS[secpar]
    d, root, inv = P["d"], P["root"], P["inv_root"]
    ctx = fusion_hip.Context(Q, d, root, inv)
    B = 4096 + 3   # ragged: not a multiple of polynomials-per-wave
    x = O.splitmix_centered(20261003, B * d).reshape(B, d)
    f = ctx.ntt_forward(x)
    assert np.array_equal(ctx.ntt_inverse(f), x)
    sel = [0, 1, 2, 3, 4, 1000, 4095, 4096, 4097, 4098]
    assert np.array_equal(f[sel], coracle.ntt_forward(x[sel], Q, root))
    assert np.array_equal(f, coracle.ntt_forward(x, Q, root))


@pytest.mark.parametrize("kernel", ["4", "16"])
@pytest.mark.parametrize("secpar", [128, 256])
def test_both_schedules_ragged_and_in_place(kernel, secpar, coracle, monkeypatch):
    """Both NTT schedules (radix-4 and 16-per-lane; forced through FZ_NTT_KERNEL, which the context reads
    at creation) on every small batch size 0..9 and a few ragged larger ones, out of place and in place."""
    import fusion_hip
    monkeypatch.setenv("FZ_NTT_KERNEL", kernel)
    P = O.PARAMS[secpar]
    d, root, inv = P["d"], P["root"], P["inv_root"]
    ctx = fusion_hip.Context(Q, d, root, inv)          # not the cached one: picks up the env override
    rng = np.random.default_rng(int(kernel) * 1000 + secpar)
    for B in list(range(0, 10)) + [15, 16, 17, 63, 65, 255, 1025]:
        x = rng.integers(-2**31, 2**31, size=(B, d), dtype=np.int64).astype(np.int32)
        assert np.array_equal(ctx.ntt_forward(x), coracle.ntt_forward(x, Q, root).reshape(B, d)), (kernel, B)
        assert np.array_equal(ctx.ntt_inverse(x), coracle.ntt_inverse(x, Q, inv).reshape(B, d)), (kernel, B)
    # device face, in place (d_in == d_out) and out of place into a poisoned buffer
    B = 1000 + 3
    x = O.splitmix_centered(4, B * d).reshape(B, d)
    buf = fusion_hip.DeviceBuffer.from_numpy(ctx, x)
    out = fusion_hip.DeviceBuffer.from_numpy(ctx, np.full((B + 1, d), 123456789, np.int32))
    ctx.ntt_forward_dev(buf.ptr, out.ptr, B)
    got = out.to_numpy(np.int32, (B + 1, d))
    want = coracle.ntt_forward(x, Q, root)
    assert np.array_equal(got[:B], want) and np.all(got[B] == 123456789)       # nothing written past the batch
    ctx.ntt_forward_dev(buf.ptr, buf.ptr, B)
    assert np.array_equal(buf.to_numpy(np.int32, (B, d)), want)
    ctx.ntt_inverse_dev(buf.ptr, buf.ptr, B)
    assert np.array_equal(buf.to_numpy(np.int32, (B, d)), x)
    ctx.close()


def test_error_codes_on_device():
    import ctypes
    import fusion_hip
    lib = fusion_hip.load_library()
    ctx = fusion_hip.Context(Q, 256, 3337519, pow(3337519, Q - 2, Q))
    buf = fusion_hip.DeviceBuffer(ctx, 4096)
    with pytest.raises(fusion_hip.FusionHipError) as e:
        ctx.ntt_forward_dev(buf.ptr + 4, buf.ptr, 1)              # misaligned
    assert e.value.code == -1
    with pytest.raises(fusion_hip.FusionHipError) as e:
        ctx.ntt_forward_dev(0, buf.ptr, 1)                        # NULL
    assert e.value.code == -1
    ring = fusion_hip.Context(Q, 100, 0, 0)                       # ring-only context: no transforms
    with pytest.raises(fusion_hip.FusionHipError) as e:
        ring.ntt_forward(np.zeros(100, np.int32))
    assert e.value.code == -2
    with pytest.raises(fusion_hip.FusionHipError):
        fusion_hip.Context(Q, 256, 5, 1)                          # not a primitive root
    assert lib.fz_ctx_destroy(None) == 0
    # the entries added after the first pass refuse bad arguments before anything is launched
    big = fusion_hip.DeviceBuffer(ctx, 1 << 16)
    with pytest.raises(fusion_hip.FusionHipError) as e:
        ctx.poly_mul_dev(0, big.ptr, big.ptr, 1)                  # NULL factor
    assert e.value.code == -1
    with pytest.raises(fusion_hip.FusionHipError) as e:
        ring.poly_mul(np.zeros(100, np.int32), np.zeros(100, np.int32))
    assert e.value.code == -2
    ctx.poly_mul_dev(big.ptr, big.ptr, big.ptr, 0)                # empty batch: nothing to do
    with pytest.raises(fusion_hip.FusionHipError, match="aligned"):
        ctx.aggregate_target_partial_batch_dev(big.ptr, big.ptr, big.ptr + 4, big.ptr, big.ptr, big.ptr, 256, big.ptr, 256,
                                               1, 1, 1)
    with pytest.raises(fusion_hip.FusionHipError, match="stride"):
        ctx.verify_partials_batch_async_dev(big.ptr, big.ptr, 8, big.ptr, 8, 2, 1, 1, 1, big.ptr)   # strides < rows
    with pytest.raises(fusion_hip.FusionHipError, match="no capture"):
        ctx.graph_end()
    big.free()


@pytest.mark.gpu
def test_graph_capture_replays_a_recorded_sequence(coracle):
    """fz_graph_*: forward -> pointwise square -> inverse recorded once, replayed on fresh contents; capture
    refuses what cannot be recorded (default stream, synchronisation, host copies)"""
    import fusion_hip
    P = O.PARAMS[256]
    q, d = P["q"], P["d"]
    ctx = fusion_hip.Context(q, d, P["root"], P["inv_root"])
    with pytest.raises(fusion_hip.FusionHipError, match="non-default stream"):
        ctx.graph_begin()
    s = ctx.stream_create()
    ctx.set_stream(s)
    rows = 77
    x = O.splitmix_centered(31, rows * d).reshape(rows, d)
    dx = fusion_hip.DeviceBuffer.from_numpy(ctx, x)
    dy = fusion_hip.DeviceBuffer(ctx, x.nbytes)
    dz = fusion_hip.DeviceBuffer(ctx, x.nbytes)
    ctx.graph_begin()
    ctx.ntt_forward_dev(dx.ptr, dy.ptr, rows)
    ctx.pw_dev(fusion_hip.OP_MUL, dy.ptr, dy.ptr, dy.ptr, rows * d)
    ctx.ntt_inverse_dev(dy.ptr, dz.ptr, rows)
    with pytest.raises(fusion_hip.FusionHipError, match="capture"):
        ctx.synchronize()
    g = ctx.graph_end()
    for seed in (31, 32):                       # same graph, new contents
        x = O.splitmix_centered(seed, rows * d).reshape(rows, d)
        ctx.h2d(dx.ptr, x)
        g.launch()
        got = ctx.d2h(np.empty_like(x), dz.ptr)
        f = coracle.ntt_forward(x, q, P["root"])
        want = coracle.ntt_inverse(coracle.pw_mul(f, f, q), q, P["inv_root"])
        assert np.array_equal(got, want)
    g.destroy()
    with pytest.raises(fusion_hip.FusionHipError, match="still attached"):
        ctx.stream_destroy(s)
    ctx.set_stream(0)
    ctx.stream_destroy(s)
    for b in (dx, dy, dz):
        b.free()
    ctx.close()


@pytest.mark.parametrize("d,form", [(256, 1), (256, 2), (128, 0), (64, 0)])
def test_graph_capture_of_the_fused_product(d, form, coracle, monkeypatch):
    """fz_poly_mul recorded into a graph with NO call before the capture (the kernel's occupancy query then runs inside it) and
    replayed on fresh contents: the radix-4 kernel, the 16-per-lane kernel (forced at degree 256, the only fused form at 128)"""
    import fusion_hip
    root = _root_for(Q, d)
    inv_root = pow(root, Q - 2, Q)
    if form:
        monkeypatch.setenv("FZ_POLYMUL_FORM", str(form))
    ctx = fusion_hip.Context(Q, d, root, inv_root)
    if form:
        monkeypatch.delenv("FZ_POLYMUL_FORM")
    s = ctx.stream_create()
    ctx.set_stream(s)
    rows = 301
    f = O.splitmix_centered(51, rows * d).reshape(rows, d)
    g = O.splitmix_centered(52, rows * d).reshape(rows, d)
    df, dg = fusion_hip.DeviceBuffer.from_numpy(ctx, f), fusion_hip.DeviceBuffer.from_numpy(ctx, g)
    dp = fusion_hip.DeviceBuffer(ctx, f.nbytes)
    ctx.synchronize()
    ctx.graph_begin()
    ctx.poly_mul_dev(df.ptr, dg.ptr, dp.ptr, rows)
    ctx.poly_mul_dev(dp.ptr, dg.ptr, dp.ptr, rows)          # (f * g) * g, in place
    gr = ctx.graph_end()
    for seed in (51, 53):
        f = O.splitmix_centered(seed, rows * d).reshape(rows, d)
        ctx.h2d(df.ptr, f)
        gr.launch()
        got = ctx.d2h(np.empty_like(f), dp.ptr)
        gh = coracle.ntt_forward(g, Q, root)
        want = coracle.ntt_inverse(coracle.pw_mul(coracle.pw_mul(coracle.ntt_forward(f, Q, root), gh, Q), gh, Q), Q, inv_root)
        assert np.array_equal(got, want.reshape(rows, d)), (d, form, seed)
    gr.destroy()
    ctx.set_stream(0)
    ctx.stream_destroy(s)
    for b in (df, dg, dp):
        b.free()
    ctx.close()


def test_events_order_two_contexts_eagerly_and_inside_a_capture(coracle):
    """fz_event_*: context A transforms forward on its stream, context B transforms back on ITS stream after waiting for A's
    event -- launched eagerly, then as ONE captured graph in which B's stream forks off A's capture and joins it again, replayed
    on new contents.  While B's stream is part of A's capture, B obeys the capture rules (no synchronisation, no host copies)."""
    import fusion_hip
    P = O.PARAMS[256]
    q, d = P["q"], P["d"]
    A = fusion_hip.Context(q, d, P["root"], P["inv_root"])
    B = fusion_hip.Context(q, d, P["root"], P["inv_root"])
    sa, sb = A.stream_create("low"), B.stream_create("high")     # (fz_stream_create_priority: priorities change scheduling, never results)
    A.set_stream(sa)
    B.set_stream(sb)
    rows = 4096
    x = O.splitmix_centered(41, rows * d).reshape(rows, d)
    dx = fusion_hip.DeviceBuffer.from_numpy(A, x)
    dy, dz = fusion_hip.DeviceBuffer(A, x.nbytes), fusion_hip.DeviceBuffer(A, x.nbytes)
    dsq = fusion_hip.DeviceBuffer(A, x.nbytes)
    e_fwd, e_back = fusion_hip.Event(A), fusion_hip.Event(A)
    A.synchronize()

    def sequence():
        A.ntt_forward_dev(dx.ptr, dy.ptr, rows)
        e_fwd.record(A)
        e_fwd.wait(B)                                  # B's stream: after A's forward transform
        B.ntt_inverse_dev(dy.ptr, dz.ptr, rows)
        e_back.record(B)
        A.pw_dev(fusion_hip.OP_MUL, dy.ptr, dy.ptr, dsq.ptr, rows * d)      # A goes on beside B
        e_back.wait(A)                                 # ... and joins B again
    for _ in range(20):                                # eagerly: a missing order would show as a torn dz sooner or later
        sequence()
    A.synchronize()
    B.synchronize()
    f = coracle.ntt_forward(x, q, P["root"])
    assert np.array_equal(B.d2h(np.empty_like(x), dz.ptr), x)
    assert np.array_equal(A.d2h(np.empty_like(x), dsq.ptr), coracle.pw_mul(f, f, q))
    A.graph_begin()
    sequence()
    with pytest.raises(fusion_hip.FusionHipError, match="capture"):
        B.synchronize()                                # B's stream has joined A's capture
    with pytest.raises(fusion_hip.FusionHipError, match="capture"):
        B.d2h(np.empty_like(x), dz.ptr)
    g = A.graph_end()
    for seed in (42, 43):
        x = O.splitmix_centered(seed, rows * d).reshape(rows, d)
        A.h2d(dx.ptr, x)
        g.launch()
        A.synchronize()
        assert np.array_equal(A.d2h(np.empty_like(x), dz.ptr), x)
        f = coracle.ntt_forward(x, q, P["root"])
        assert np.array_equal(A.d2h(np.empty_like(x), dsq.ptr), coracle.pw_mul(f, f, q))
    B.synchronize()                                    # allowed again: the capture has ended
    g.destroy()
    for e in (e_fwd, e_back):
        e.destroy()
    for b in (dx, dy, dz, dsq):
        b.free()
    for c, s_ in ((A, sa), (B, sb)):
        c.set_stream(0)
        c.stream_destroy(s_)
        c.close()


@pytest.mark.parametrize("d", [4, 16, 32, 64, 128, 256])
def test_poly_mul_matches_transform_composition(d, coracle, monkeypatch):
    """fz_poly_mul (fused kernel at d = 64 / 256, composed launches otherwise) == INTT(NTT f * NTT g) of the oracle,
    on edge rows and ragged seeded batches, in place, and equal to the unfused path"""
    import fusion_hip
    root = _root_for(Q, d)
    inv_root = pow(root, Q - 2, Q)
    ctx = fusion_hip.Context(Q, d, root, inv_root)
    edge = _edge_rows(d, Q)
    for rows, seed in ((len(edge), None), (1, 3), (5, 4), (1003, 5)):
        f = edge if seed is None else O.splitmix_centered(seed, rows * d).reshape(rows, d)
        g = edge[::-1].copy() if seed is None else O.splitmix_centered(seed + 100, rows * d).reshape(rows, d)
        want = coracle.ntt_inverse(coracle.pw_mul(coracle.ntt_forward(f, Q, root), coracle.ntt_forward(g, Q, root), Q),
                                   Q, inv_root)
        assert np.array_equal(ctx.poly_mul(f, g), want)
        # device pointers, output aliasing the first input
        df, dg = fusion_hip.DeviceBuffer.from_numpy(ctx, f), fusion_hip.DeviceBuffer.from_numpy(ctx, g)
        ctx.poly_mul_dev(df.ptr, dg.ptr, df.ptr, rows)
        assert np.array_equal(ctx.d2h(np.empty_like(f), df.ptr), want)
        df.free(); dg.free()
    if d in (64, 256):
        monkeypatch.setenv("FZ_UNFUSED", "1")
        f = O.splitmix_centered(8, 77 * d).reshape(77, d)
        g = O.splitmix_centered(9, 77 * d).reshape(77, d)
        unfused = ctx.poly_mul(f, g)
        monkeypatch.delenv("FZ_UNFUSED")
        assert np.array_equal(ctx.poly_mul(f, g), unfused)
    # x * 1 = cent(x); x * X = negacyclic shift
    one = np.zeros((1, d), np.int32); one[0, 0] = 1
    xs = O.splitmix_centered(11, d).reshape(1, d)
    assert np.array_equal(ctx.poly_mul(xs, one), xs)
    X = np.zeros((1, d), np.int32); X[0, 1] = 1
    assert np.array_equal(ctx.poly_mul(xs, X)[0], np.concatenate([-xs[0, -1:], xs[0, :-1]]))
    ctx.close()


@pytest.mark.parametrize("q", [Q, 65537, 4294828033])
@pytest.mark.parametrize("d", [32, 64, 128, 256])
def test_poly_mul_sixteen_per_lane_form(d, q, coracle, monkeypatch):
    """FZ_POLYMUL_FORM=2: the product kernel built on the 16-per-lane transforms (what batches of >= 2^14 products take at
    d = 256 and every aligned batch at d = 32 / 128) == the oracle's INTT(NTT f * NTT g) and == the radix-4 form, for the
    scheme's prime (4-op multiply), a prime without the pseudo-Mersenne form and one above 2^31; edge rows (raw int32 extremes),
    ragged last chunks (1, 3, 5, 1003 rows), a batch of several iterations per wave, the product written over either factor"""
    import fusion_hip
    root = _root_for(q, d)
    inv_root = pow(root, q - 2, q)
    monkeypatch.setenv("FZ_POLYMUL_FORM", "2")
    ctx = fusion_hip.Context(q, d, root, inv_root)         # a fresh context: the knob is read here
    monkeypatch.setenv("FZ_POLYMUL_FORM", "1")
    ctx4 = fusion_hip.Context(q, d, root, inv_root)
    monkeypatch.delenv("FZ_POLYMUL_FORM")
    edge = _edge_rows(d, q if q < 2**31 else 2**31 - 1)
    big = 8 * 1024 * (1024 // d) + 3                         # more chunks than a resident grid has waves: the loop runs, and ends ragged
    rng = np.random.default_rng(d + q % 1000)
    for rows, seed in ((len(edge), None), (1, 3), (3, 6), (5, 4), (1003, 5), (big, 7)):
        if seed is None:
            f, g = edge, edge[::-1].copy()
        elif rows == big:
            f = rng.integers(-2**31, 2**31, size=(rows, d), dtype=np.int64).astype(np.int32)       # any int32 is an admissible input
            g = rng.integers(-(q // 2), q // 2 + 1, size=(rows, d), dtype=np.int64).astype(np.int32)
        else:
            f = O.splitmix_centered(seed, rows * d).reshape(rows, d)
            g = O.splitmix_centered(seed + 100, rows * d).reshape(rows, d)
        got = ctx.poly_mul(f, g)
        if rows <= 1003:
            want = coracle.ntt_inverse(coracle.pw_mul(coracle.ntt_forward(f, q, root), coracle.ntt_forward(g, q, root), q), q, inv_root)
            assert np.array_equal(got, want), (d, q, rows)
        if d in (64, 256):
            assert np.array_equal(got, ctx4.poly_mul(f, g)), (d, q, rows)
        else:
            sub = slice(0, rows, max(1, rows // 200))
            want = coracle.ntt_inverse(coracle.pw_mul(coracle.ntt_forward(f[sub], q, root), coracle.ntt_forward(g[sub], q, root), q), q, inv_root)
            assert np.array_equal(got[sub], want), (d, q, rows)
        for alias in (0, 1):
            df, dg = fusion_hip.DeviceBuffer.from_numpy(ctx, f), fusion_hip.DeviceBuffer.from_numpy(ctx, g)
            dst = (df, dg)[alias]
            ctx.poly_mul_dev(df.ptr, dg.ptr, dst.ptr, rows)
            assert np.array_equal(ctx.d2h(np.empty_like(f), dst.ptr), got), (d, q, rows, alias)
            df.free(); dg.free()
    # operands that are not 16-byte aligned: the other form (d = 64 / 256) answers
    if d in (64, 256):
        f = O.splitmix_centered(21, 9 * d).reshape(9, d)
        g = O.splitmix_centered(22, 9 * d).reshape(9, d)
        buf = fusion_hip.DeviceBuffer(ctx, 3 * 9 * d * 4 + 64)
        pf, pg, po = buf.ptr + 4, buf.ptr + 4 + 9 * d * 4, buf.ptr + 4 + 2 * 9 * d * 4
        ctx.h2d(pf, f); ctx.h2d(pg, g)
        ctx.poly_mul_dev(pf, pg, po, 9)
        assert np.array_equal(ctx.d2h(np.empty_like(f), po), ctx4.poly_mul(f, g))
        buf.free()
    ctx.close(); ctx4.close()


def test_c_caller_round_trip(tmp_path):
    """examples/roundtrip.c (C99, gcc, no HIP headers): context, stream, device buffers, graph capture, replay"""
    import subprocess
    from test_cabi_symbols import build_c_example
    r = subprocess.run([build_c_example(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "== x on 1000 rows" in r.stdout


@pytest.mark.parametrize("secpar,rows", [(256, (1 << 15) + 5), (256, (1 << 17) + 3), (128, (1 << 19) + 7)])
def test_large_batches_above_the_schedule_crossover(secpar, rows, coracle):
    """default schedule choice at sizes where the 16-per-lane kernels take over (ragged row counts): the whole batch
    against the C oracle, the round trip, and linearity NTT(x + y) == NTT(x) + NTT(y) mod q"""
    import fusion_hip
    P = O.PARAMS[secpar]
    d, root, inv = P["d"], P["root"], P["inv_root"]
    ctx = fusion_hip.Context(Q, d, root, inv)
    x = O.splitmix_centered(77, rows * d).reshape(rows, d)
    f = ctx.ntt_forward(x)
    assert np.array_equal(f, coracle.ntt_forward(x, Q, root).reshape(rows, d))
    assert np.array_equal(ctx.ntt_inverse(f), x)
    y = O.splitmix_centered(78, rows * d).reshape(rows, d)
    lhs = ctx.ntt_forward(ctx.pw_add(x, y))
    rhs = ctx.pw_add(f, ctx.ntt_forward(y))
    assert np.array_equal(lhs, rhs)
    ctx.close()
