"""The parameter space the reference accepts beyond the scheme's two sets: any odd prime with a 2n-th root and any
power-of-two length (algebra/ntt.py:239-270, algebra/polynomials.py:16-50), and whatever twiddle TABLE the caller hands to
cooley_tukey_ntt / gentleman_sande_intt (ntt.py:274-290, :354-372).  Round 5 widened the HIP path instead of adding a CPU
route: moduli up to 2^32 - 1 (centred residues are int32 for every such q), lengths up to 4096, contexts built from arbitrary
tables.  Everything here is checked against the PURE-PYTHON restatement of the reference's loops (oracle.py py_*: Python
integers, no 64-bit limits).  What the INT32 contexts refuse -- q >= 2^32, lengths above 4096 -- is pinned by exact exception and
message; the drop-in packages serve those parameters through the generic int64 path (tests/test_gpu_wide.py)."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu

Q_LOW = 2147565569       # 2^31 + 81921, = 1 (mod 8192): the first prime above 2^31 with a 8192-th root
Q_TOP = 4294828033       # 2^32 - 139263, = 1 (mod 8192): the last one below 2^32
Q_PM = 4294962689        # 2^32 - 4607, = 1 (mod 512): pseudo-Mersenne shaped, but the 4-op multiply stays below 2^31
I32 = np.iinfo(np.int32)


def root_of(q, n):
    """a primitive 2n-th root of unity mod q"""
    for g in range(2, 5000):
        r = pow(g, (q - 1) // (2 * n), q)
        if pow(r, n, q) == q - 1:
            return r
    raise AssertionError("no root")


def cent_rows(rng, q, shape):
    half = (q - 1) // 2
    return rng.integers(-half, half + 1, size=shape, dtype=np.int64).astype(np.int32)


@pytest.mark.parametrize("kernel", ["auto", "4", "16"])
@pytest.mark.parametrize("q,d", [(Q_LOW, 256), (Q_TOP, 256), (Q_PM, 256), (Q_TOP, 64), (Q_LOW, 64), (Q_TOP, 4), (Q_LOW, 16), (Q_PM, 128),
                                 (Q_TOP, 2), (Q_TOP, 32)])
def test_transforms_over_moduli_between_2_31_and_2_32(q, d, kernel, monkeypatch):
    """forward and inverse transforms, every kernel family (thread-per-row, radix-4 wave-tasks, 16 per lane, the multi-job
    launch), centred and RAW int32 rows, against the reference's loops on Python integers"""
    import fusion_hip
    if kernel != "auto":
        monkeypatch.setenv("FZ_NTT_KERNEL", kernel)
    root = root_of(q, d)
    inv = pow(root, q - 2, q)
    ctx = fusion_hip.Context(q, d, root, inv)
    tw, itw = O.py_twiddles(root, q, d), O.py_twiddles(inv, q, d)
    f_tab, i_tab = ctx.twiddles()
    assert f_tab.tolist() == tw and i_tab.tolist() == itw
    rng = np.random.default_rng(d + q % 1000)
    x = cent_rows(rng, q, (37, d))
    x[0] = 0
    x[1] = (q - 1) // 2
    x[2] = -((q - 1) // 2)
    x[3] = rng.integers(I32.min, I32.max, size=d, dtype=np.int64).astype(np.int32)        # raw int32, not reduced
    x[4, ::2], x[4, 1::2] = I32.max, I32.min
    y = ctx.ntt_forward(x)
    for r in range(8):
        assert y[r].tolist() == O.py_ntt_forward([int(v) for v in x[r]], q, tw), (q, d, r)
    z = ctx.ntt_inverse(y)
    assert np.array_equal(z[:3], x[:3]) and np.array_equal(z[5:], x[5:])
    for r in (3, 4):
        assert z[r].tolist() == [O.py_cent(int(v), q) for v in x[r]]
    w = ctx.ntt_inverse(x)
    for r in range(6):
        assert w[r].tolist() == O.py_ntt_inverse([int(v) for v in x[r]], q, itw), (q, d, r)
    # large batches reach the other schedules; the round trip is the identity on centred rows
    big = cent_rows(rng, q, (3000 if d >= 64 else 300, d))
    assert np.array_equal(ctx.ntt_inverse(ctx.ntt_forward(big)), big)
    if d in (64, 256):
        DB = fusion_hip.DeviceBuffer
        bufs = [DB.from_numpy(ctx, big[k * 500:(k + 1) * 500]) for k in range(6)]
        outs = [DB(ctx, 500 * d * 4) for _ in range(6)]
        ctx.ntt_multi_dev([(b.ptr, o.ptr, 500, k % 2 == 1) for k, (b, o) in enumerate(zip(bufs, outs))])
        ctx.synchronize()
        for k, o in enumerate(outs):
            got = o.to_numpy(np.int32, (500, d))
            want = ctx.ntt_inverse(big[k * 500:(k + 1) * 500]) if k % 2 else ctx.ntt_forward(big[k * 500:(k + 1) * 500])
            assert np.array_equal(got, want), k
            assert got[7].tolist() == (O.py_ntt_inverse if k % 2 else O.py_ntt_forward)([int(v) for v in big[k * 500 + 7]], q, itw if k % 2 else tw)
        for b in bufs + outs:
            b.free()
    ctx.close()


@pytest.mark.parametrize("q,d", [(Q_TOP, 512), (Q_TOP, 1024), (Q_LOW, 2048), (Q_TOP, 4096), (12289, 512), (12289, 2048), (65537, 4096),
                                 (O.PRIME, 256)])
def test_lengths_above_256(q, d):
    """512 .. 4096 coefficients (one workgroup per polynomial through LDS): the reference transforms any power-of-two length
    (ntt.py:239-270).  The scheme's prime has no root beyond order 512: its largest length, 256, runs beside them as the control."""
    import fusion_hip
    root = root_of(q, d)
    inv = pow(root, q - 2, q)
    ctx = fusion_hip.Context(q, d, root, inv)
    tw, itw = O.py_twiddles(root, q, d), O.py_twiddles(inv, q, d)
    rng = np.random.default_rng(d)
    x = cent_rows(rng, q, (300, d))
    x[1] = rng.integers(I32.min, I32.max, size=d, dtype=np.int64).astype(np.int32)
    x[2, ::2], x[2, 1::2] = I32.max, I32.min
    y = ctx.ntt_forward(x)
    z = ctx.ntt_inverse(x)
    for r in range(3):
        assert y[r].tolist() == O.py_ntt_forward([int(v) for v in x[r]], q, tw), (q, d, r)
        assert z[r].tolist() == O.py_ntt_inverse([int(v) for v in x[r]], q, itw), (q, d, r)
    back = ctx.ntt_inverse(y)
    assert np.array_equal(np.delete(back, [1, 2], axis=0), np.delete(x, [1, 2], axis=0))
    # the negacyclic product through the composed launches (fz_poly_mul at a length without a fused kernel)
    f, g = cent_rows(rng, q, (2, d)), cent_rows(rng, q, (2, d))
    f[1, 5:] = 0
    g[1, 3:] = 0                                          # a sparse pair: a cheap schoolbook check
    prod = ctx.poly_mul(f, g)
    want = [0] * d
    for i in range(5):
        for j in range(3):
            want[i + j] += int(f[1, i]) * int(g[1, j])
    assert prod[1].tolist() == [O.py_cent(v, q) for v in want]
    if d <= 1024:
        assert prod[0].tolist() == O.py_schoolbook([int(v) for v in f[0]], [int(v) for v in g[0]], q)
    ctx.close()


@pytest.mark.parametrize("q", [Q_LOW, Q_TOP, Q_PM])
def test_ring_operations_and_scheme_cores_over_32_bit_moduli(q):
    """pointwise + - * mulacc, the (1 x l).(l x 1) product and the fused keygen / sign / aggregate / verify arithmetic
    (fusion.py:363-370, :557, :670-676, :690-727) with a modulus above 2^31, against the reference's formulas on Python integers"""
    import fusion_hip
    d, l, n = 256, 5, 7
    root = root_of(q, d)
    inv = pow(root, q - 2, q)
    ctx = fusion_hip.Context(q, d, root, inv)
    tw, itw = O.py_twiddles(root, q, d), O.py_twiddles(inv, q, d)
    rng = np.random.default_rng(q % 9973)
    a, b = cent_rows(rng, q, (4, d)), cent_rows(rng, q, (4, d))
    a[0] = rng.integers(I32.min, I32.max, size=d, dtype=np.int64).astype(np.int32)
    b[0, ::2], b[0, 1::2] = I32.min, I32.max
    L = lambda m: [[int(v) for v in r] for r in m]                                      # noqa: E731
    assert ctx.pw_mul(a, b).tolist() == [O.py_pw_mul(x, y, q) for x, y in zip(L(a), L(b))]
    assert ctx.pw_add(a, b).tolist() == [O.py_pw_add(x, y, q) for x, y in zip(L(a), L(b))]
    assert ctx.pw_sub(a, b).tolist() == [O.py_pw_sub(x, y, q) for x, y in zip(L(a), L(b))]
    with pytest.raises(fusion_hip.FusionHipError) as e:                                # -(x mod q) is not an int32 for q >= 2^31
        ctx.pw_neg(a)
    assert e.value.code == -2 and "does not fit int32" in str(e.value)
    A = cent_rows(rng, q, (l, d))
    S = cent_rows(rng, q, (3, l, d))
    assert ctx.matvec(A, S).tolist() == [O.py_matvec(L(A), L(s), q) for s in S]
    coef = (rng.integers(1, 53, size=(n, 2, l, d)) * rng.choice(np.array([-1, 1]), size=(n, 2, l, d))).astype(np.int32)
    sk, vk = ctx.keygen_core(A, coef)
    for i in range(n):
        Lh, Rh, vL, vR = O.py_keygen_core(L(A), L(coef[i, 0]), L(coef[i, 1]), q, tw)
        assert sk[i, 0].tolist() == Lh and sk[i, 1].tolist() == Rh and vk[i, 0].tolist() == vL and vk[i, 1].tolist() == vR
    sparse = np.zeros((2, n, d), np.int32)
    for t in range(2):
        for i in range(n):
            sparse[t, i, rng.choice(d, 60, replace=False)] = rng.choice([-1, 1], 60)
    c_hat, al_hat = ctx.ntt_forward(sparse[0]), ctx.ntt_forward(sparse[1])
    sig = ctx.sign_core(sk, c_hat)
    for i in range(n):
        assert sig[i].tolist() == O.py_sign_core(L(sk[i, 0]), L(sk[i, 1]), [int(v) for v in c_hat[i]], q)
    agg = ctx.aggregate_core(sig, al_hat)
    assert agg.tolist() == O.py_aggregate_core([L(s) for s in sig], L(al_hat), q)
    beta = (q - 1) // 2
    args = (L(A), L(agg), L(vk[:, 0]), L(vk[:, 1]), L(c_hat), L(al_hat), q, itw)
    assert ctx.verify_core(A, agg, vk[:, 0], vk[:, 1], c_hat, al_hat, beta, d) == O.py_verify_core(*args, beta, d) == 0
    coefmax = max(abs(v) for r in agg for v in O.py_ntt_inverse([int(x) for x in r], q, itw))
    assert ctx.verify_core(A, agg, vk[:, 0], vk[:, 1], c_hat, al_hat, coefmax - 1, d) == O.py_verify_core(*args, coefmax - 1, d) == 4
    assert ctx.verify_core(A, agg, vk[:, 0], vk[:, 1], c_hat, al_hat, coefmax, d) == 0
    bad = agg.copy()
    bad[2, 9] += 1
    assert ctx.verify_core(A, bad, vk[:, 0], vk[:, 1], c_hat, al_hat, beta, d) == 3
    ctx.close()


@pytest.mark.parametrize("q,d", [(O.PRIME, 256), (O.PRIME, 64), (257, 8), (Q_TOP, 32), (65537, 1024), (17, 2)])
def test_transforms_with_tables_that_are_not_power_tables(q, d):
    """cooley_tukey_ntt / gentleman_sande_intt use WHATEVER table they are handed (ntt.py:277 `s = bit_rev_root_powers[m + i]`,
    :357): random tables, a table of another root's powers in natural order, an all-ones table -- through the drop-in functions
    (contexts built from the lists: fz_ctx_create_tables) against the reference's loops on the same lists"""
    import algebra.ntt as N
    rng = np.random.default_rng(d * 7 + q % 97)
    tables = [[int(v) for v in rng.integers(0, q, size=d)], [1] * d, [pow(3, i, q) for i in range(d)],
              [int(v) for v in rng.integers(0, q, size=d + 5)]]                           # (longer than needed: entries past d are never read)
    for tab in tables:
        for _ in range(2):
            x = [int(v) for v in rng.integers(-(q // 2), q // 2 + 1, size=d)]
            want_f = O.py_ntt_forward(list(x), q, tab)
            want_i = O.py_ntt_inverse(list(x), q, tab)
            got = list(x)
            assert N.cooley_tukey_ntt(got, q, 2 * d, tab) is got and got == want_f
            got = list(x)
            assert N.gentleman_sande_intt(got, q, 2 * d, tab) is got and got == want_i
    with pytest.raises(IndexError):
        N.cooley_tukey_ntt([1] * d, q, 2 * d, [1] * (d - 1))                                  # the reference's own failure for a short table
    # and a real power table still goes through the shared ring context
    root = root_of(q, d)
    tw = O.py_twiddles(root, q, d)
    x = [int(v) for v in rng.integers(-(q // 2), q // 2 + 1, size=d)]
    got = list(x)
    N.cooley_tukey_ntt(got, q, 2 * d, tw)
    assert got == O.py_ntt_forward(list(x), q, tw)
    assert N._root_of_table(tw, q, d) == root and N._root_of_table(tables[0][:d], q, d) is None


def test_polynomial_objects_over_a_modulus_above_2_31():
    """the drop-in classes at q = 4294828033: transform both ways, * + - in both representations, == mod q, norm / weight, and
    __neg__ -- the reference's -(x mod q) in [-(q-1), 0] (polynomials.py:155-163, :325-333), which no longer fits the device's
    int32 and comes back as Python ints"""
    from algebra.polynomials import PolynomialCoefficientRepresentation as PC, PolynomialNTTRepresentation as PN, transform
    q, d = Q_TOP, 64
    root = root_of(q, d)
    inv = pow(root, q - 2, q)
    rng = np.random.default_rng(5)
    f = [int(v) for v in rng.integers(-(q // 2), q // 2 + 1, size=d)]
    g = [int(v) for v in rng.integers(-(q // 2), q // 2 + 1, size=d)]
    pf, pg = PC(q, d, root, inv, 2 * d, list(f)), PC(q, d, root, inv, 2 * d, list(g))
    tw, itw = O.py_twiddles(root, q, d), O.py_twiddles(inv, q, d)
    fh = transform(pf)
    assert isinstance(fh, PN) and fh.values == O.py_ntt_forward(list(f), q, tw)
    assert transform(fh).coefficients == f
    prod = pf * pg
    assert prod.coefficients == O.py_schoolbook(f, g, q)
    assert transform(transform(pf) * transform(pg)) == prod
    assert (pf + pg).coefficients == O.py_pw_add(f, g, q)
    neg = -pf
    assert neg.coefficients == [-(v % q) for v in f] and min(neg.coefficients) < -(2 ** 31)      # really outside int32
    assert (pf - pg).coefficients == O.py_pw_sub(f, g, q)
    assert (-fh).values == [-(v % q) for v in fh.values]
    assert neg == PC(q, d, root, inv, 2 * d, [(-v) % q for v in f])                              # equality is mod q
    assert pf.norm("infty") == max(abs(v) for v in f) and pf.weight() == sum(1 for v in f if v % q)
    assert (neg + pf) == PC(q, d, root, inv, 2 * d, [0] * d)


def test_what_is_still_refused_is_refused_loudly():
    """the two classes the int32 contexts do not take, each with its exact error: a modulus that does not fit their storage
    type (q >= 2^32) and a transform longer than 4096 -- FusionHipError, never a silent CPU route; the drop-in packages hand
    both to the generic int64 kernels"""
    import algebra.ntt as N
    import fusion_hip
    from algebra.polynomials import PolynomialCoefficientRepresentation as PC
    q64 = 4294967311                                   # the first prime above 2^32: served by the generic int64 path since
    assert (PC(q64, 1, 1, 1, 1, [5]) + PC(q64, 1, 1, 1, 1, [7])).coefficients == [12]      # the end of round 5 (tests/test_gpu_wide.py)
    with pytest.raises(fusion_hip.FusionHipError) as e:
        fusion_hip.Context(2 ** 32 + 15, 4, 2, 3)      # the int32 context itself still is what it is
    assert e.value.code == -1 and "outside (0, 2^32)" in str(e.value)
    q, d = 65537, 8192
    root = root_of(q, d)
    with pytest.raises(fusion_hip.FusionHipError) as e:
        fusion_hip.Context(q, d, root, pow(root, q - 2, q))
    assert e.value.code == -2 and "degree 8192 > 4096 not supported" in str(e.value)
    tw = O.py_twiddles(root, q, d)
    assert N.cooley_tukey_ntt(list(range(d)), q, 2 * d, tw) == O.py_ntt_forward(list(range(d)), q, tw)   # (the drop-in: generic path)
    with pytest.raises(fusion_hip.FusionHipError) as e:
        fusion_hip.Context(65536, 4, 2, 3)             # an even modulus
    assert e.value.code == -1 and "odd" in str(e.value)


def test_matrix_negation_between_2_31_and_2_32():
    """-(x mod q) of every entry (matrices.py:125-129 on polynomials.py:325-333) where the value no longer fits the int32 rows"""
    from algebra.matrices import GeneralMatrix
    from algebra.polynomials import PolynomialNTTRepresentation as PN
    q, d = Q_TOP, 8
    root = root_of(q, d)
    inv = pow(root, q - 2, q)
    half = (q - 1) // 2
    rows = [[half, -half, 0, 1, -1, 5, -7, half - 3], [3, -3, half, 0, 0, -half, 2, -2]]
    M = GeneralMatrix(matrix=[[PN(q, d, root, inv, 2 * d, list(r))] for r in rows])
    assert [z[0].values for z in (-M).matrix] == [[-(x % q) for x in r] for r in rows]


# ---- numbers the REFERENCE produced for these parameters (tests/golden/generic.npz, made by tests/golden/gen_golden.py) -----
def _generic():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "generic.npz"))


@pytest.mark.parametrize("kernel", ["auto", "4", "16"])
@pytest.mark.parametrize("tag", ["g4294828033_256", "g4294828033_2048"])
def test_reference_outputs_for_a_prime_between_2_31_and_2_32(tag, kernel, monkeypatch):
    """q = 4294828033 at d = 256 (every kernel family) and d = 2048 (the one-workgroup-per-polynomial kernels): forward and
    inverse transforms of centred AND raw int32 rows, pointwise * + -, the (1 x l)(l x 1) product -- against what the reference's
    own functions and classes returned for the same inputs"""
    import fusion_hip
    g = _generic()
    q, d, root = (int(v) for v in g[f"{tag}_params"])
    if kernel != "auto":
        if d > 256:
            pytest.skip("one schedule above 256 coefficients")
        monkeypatch.setenv("FZ_NTT_KERNEL", kernel)
    ctx = fusion_hip.Context(q, d, root, pow(root, q - 2, q))
    try:
        f_tab, i_tab = ctx.twiddles()
        assert np.array_equal(f_tab.astype(np.int64), g[f"{tag}_tw"]) and np.array_equal(i_tab.astype(np.int64), g[f"{tag}_itw"])
        x = g[f"{tag}_x"].astype(np.int32)
        assert np.array_equal(ctx.ntt_forward(x).astype(np.int64), g[f"{tag}_fwd"])
        assert np.array_equal(ctx.ntt_inverse(x).astype(np.int64), g[f"{tag}_inv"])
        # the same rows in a batch large enough for the other schedules (3000 copies of the block)
        big = np.tile(x, (300, 1))
        assert np.array_equal(ctx.ntt_forward(big).astype(np.int64), np.tile(g[f"{tag}_fwd"], (300, 1)))
        assert np.array_equal(ctx.ntt_inverse(big).astype(np.int64), np.tile(g[f"{tag}_inv"], (300, 1)))
        a, b = g[f"{tag}_pw_a"].astype(np.int32), g[f"{tag}_pw_b"].astype(np.int32)
        assert np.array_equal(ctx.pw_mul(a, b).astype(np.int64), g[f"{tag}_pw_mul"])
        assert np.array_equal(ctx.pw_add(a, b).astype(np.int64), g[f"{tag}_pw_add"])
        assert np.array_equal(ctx.pw_sub(a, b).astype(np.int64), g[f"{tag}_pw_sub"])
        assert np.array_equal(ctx.matvec(g[f"{tag}_mv_A"].astype(np.int32), g[f"{tag}_mv_S"].astype(np.int32)).astype(np.int64), g[f"{tag}_mv_out"])
    finally:
        ctx.close()


@pytest.mark.parametrize("tag", ["t2147465729_64", "t2147465729_256"])
def test_reference_outputs_for_tables_that_are_no_roots_powers(tag):
    """the scheme's prime with random tables (fz_ctx_create_tables): what cooley_tukey_ntt / gentleman_sande_intt returned with
    those tables, for centred and raw int32 rows -- through a context built from the tables and through the drop-in functions"""
    import algebra.ntt as N
    import fusion_hip
    g = _generic()
    q, d, _ = (int(v) for v in g[f"{tag}_params"])
    tw, itw = [int(v) for v in g[f"{tag}_tw"]], [int(v) for v in g[f"{tag}_itw"]]
    ctx = fusion_hip.Context(q, d, 0, 0, tables=(tw, itw))
    try:
        x = g[f"{tag}_x"].astype(np.int32)
        assert np.array_equal(ctx.ntt_forward(x).astype(np.int64), g[f"{tag}_fwd"])
        assert np.array_equal(ctx.ntt_inverse(x).astype(np.int64), g[f"{tag}_inv"])
    finally:
        ctx.close()
    for i in (1, 5, 9):
        row = [int(v) for v in g[f"{tag}_x"][i]]
        got = list(row)
        assert N.cooley_tukey_ntt(got, q, 2 * d, tw) is got and got == [int(v) for v in g[f"{tag}_fwd"][i]]
        got = list(row)
        assert N.gentleman_sande_intt(got, q, 2 * d, itw) is got and got == [int(v) for v in g[f"{tag}_inv"][i]]
