"""The generic-parameter path (csrc/fz_wide.hip, fusion_hip/wide.py): moduli from 2^32 up to 2^63 and transform lengths beyond 4096 --
what the reference accepts (any odd modulus, any power-of-two length, whatever table it is handed: algebra/ntt.py:239-290,
:345-377; algebra/polynomials.py) and the int32 kernels do not take.  Everything is compared with the PURE-PYTHON restatement of
the reference's loops (oracle.py py_*: Python integers, no 64-bit limits) or with Python-integer arithmetic written out here;
the drop-in classes are exercised on the same parameters.  What is still refused (q >= 2^63, even moduli) is pinned."""
import random

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def is_prime(n):
    """deterministic Miller-Rabin for n < 2^64"""
    if n < 2:
        return False
    for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d, s = d // 2, s + 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def prime_near(start, step, down=False):
    """the first prime q = 1 (mod step) at or after `start` (down: at or before)"""
    q = start - (start - 1) % step if down else start + (1 - start) % step
    while not is_prime(q):
        q += -step if down else step
    return q


Q33 = prime_near(2 ** 32, 2 ** 15)               # just above 2^32, has 2^15-th roots of unity
Q62 = prime_near(2 ** 62, 2 ** 15, down=True)    # just below 2^62
Q63 = prime_near(2 ** 63 - 1, 2 ** 15, down=True)   # the largest modulus class the path takes: just below 2^63
Q41 = prime_near(2 ** 41, 2 ** 15)               # (the classes' constructors test primality by trial division, as the reference's do:
#                                                  seconds at 2^41, a minute at 2^62 -- the class tests stay below)


def root_of(q, n):
    for g in range(2, 5000):
        r = pow(g, (q - 1) // (2 * n), q)
        if pow(r, n, q) == q - 1:
            return r
    raise AssertionError("no root")


def cent_rows(rnd, q, rows, d):
    half = (q - 1) // 2
    return [[rnd.randint(-half, half) for _ in range(d)] for _ in range(rows)]


def cent(x, q):
    y = x % q
    return y - q if y > (q - 1) // 2 else y


@pytest.mark.parametrize("q,d", [(Q33, 2), (Q33, 8), (Q33, 256), (Q62, 256), (Q63, 64), (Q63, 1024), (Q62, 8192), (65537, 8192)])
def test_wide_transforms_match_the_reference_loops(q, d):
    """forward and inverse transforms from root-generated tables: every row against cooley_tukey_ntt / gentleman_sande_intt's loops on
    Python integers (ntt.py:274-290, :354-376); extremes +-(q-1)/2 included"""
    from fusion_hip.wide import WideContext, bit_reversed_powers
    root = root_of(q, d)
    inv = pow(root, q - 2, q)
    tw, itw = O.py_twiddles(root, q, d), O.py_twiddles(inv, q, d)
    assert bit_reversed_powers(root, q, d) == tw
    ctx = WideContext(q, d, tw, itw)
    rnd = random.Random(d + q % 1000)
    rows = 3 if d > 1024 else 6
    x = cent_rows(rnd, q, rows, d)
    x[0] = [(q - 1) // 2] * d
    x[1] = [-((q - 1) // 2) if k % 2 else (q - 1) // 2 for k in range(d)]
    y = ctx.ntt_forward(np.array(x, dtype=np.int64))
    assert y.dtype == np.int64
    for r in range(rows):
        assert y[r].tolist() == O.py_ntt_forward(list(x[r]), q, tw), (q, d, r)
    z = ctx.ntt_inverse(y)
    assert z.tolist() == x
    w = ctx.ntt_inverse(np.array(x, dtype=np.int64))                      # the inverse on arbitrary input, not only on forward outputs
    for r in range(min(rows, 3)):
        assert w[r].tolist() == O.py_ntt_inverse(list(x[r]), q, itw), (q, d, r)


@pytest.mark.parametrize("q,d", [(Q62, 64), (2147465729, 16384)])
def test_wide_transforms_use_the_table_they_are_handed(q, d):
    """tables that are not the power table of any root (random residues): the loops use entry m + i / h + i whatever it holds --
    also the scheme's own prime at a length (16 384) for which it has no root of unity and the int32 kernels no schedule"""
    from fusion_hip.wide import WideContext
    rnd = random.Random(5)
    tab = [rnd.randrange(q) for _ in range(d)]
    itab = [rnd.randrange(q) for _ in range(d)]
    ctx = WideContext(q, d, tab, itab)
    x = cent_rows(rnd, q, 4, d)
    y = ctx.ntt_forward(np.array(x, dtype=np.int64))
    z = ctx.ntt_inverse(np.array(x, dtype=np.int64))
    for r in range(4):
        assert y[r].tolist() == O.py_ntt_forward(list(x[r]), q, tab)
        assert z[r].tolist() == O.py_ntt_inverse(list(x[r]), q, itab)


@pytest.mark.parametrize("q", [Q33, Q62, Q63, 2 ** 40 + 15, 3 * (2 ** 50) + 1])
def test_wide_pointwise_matvec_norm_weight(q):
    """+, -, *, negation, the (1 x l)(l x 1) product, norm and weight on Python integers (odd moduli, prime or not: the pointwise
    operations of polynomials.py:140-216, :272-333 and matrices.py:143-181 ask for nothing else)"""
    from fusion_hip.wide import WideContext
    if q % 2 == 0:
        q += 1
    d, l, batch = 48, 7, 5
    ctx = WideContext(q, d)
    rnd = random.Random(q % 9973)
    a, b = cent_rows(rnd, q, batch, d), cent_rows(rnd, q, batch, d)
    a[0][:4] = [(q - 1) // 2, -((q - 1) // 2), 0, 1]
    b[0][:4] = [(q - 1) // 2, (q - 1) // 2, 0, -1]
    A, B = np.array(a, dtype=np.int64), np.array(b, dtype=np.int64)
    assert ctx.pw_add(A, B).tolist() == [[cent(x + y, q) for x, y in zip(r, s)] for r, s in zip(a, b)]
    assert ctx.pw_sub(A, B).tolist() == [[cent(x - y, q) for x, y in zip(r, s)] for r, s in zip(a, b)]
    assert ctx.pw_mul(A, B).tolist() == [[cent(x * y, q) for x, y in zip(r, s)] for r, s in zip(a, b)]
    assert ctx.pw_neg(A).tolist() == [[-(x % q) for x in r] for r in a]
    Am = cent_rows(rnd, q, l, d)
    S = [cent_rows(rnd, q, l, d) for _ in range(batch)]
    got = ctx.matvec(np.array(Am, dtype=np.int64), np.array(S, dtype=np.int64))
    want = [[cent(sum(Am[k][j] * S[bb][k][j] for k in range(l)), q) for j in range(d)] for bb in range(batch)]
    assert got.tolist() == want
    a[1] = [0] * d
    a[2] = [0] * (d - 1) + [-5]
    mx, wt = ctx.norm_weight(np.array(a, dtype=np.int64))
    assert mx.tolist() == [max(abs(x) for x in r) for r in a]
    assert wt.tolist() == [sum(1 for x in r if x != 0) for r in a]


def schoolbook_negacyclic(f, g, q):
    d = len(f)
    out = [0] * d
    for i, x in enumerate(f):
        for j, y in enumerate(g):
            k = i + j
            if k < d:
                out[k] += x * y
            else:
                out[k - d] -= x * y
    return [cent(v, q) for v in out]


@pytest.mark.parametrize("q,d", [(Q33, 16), (Q41, 32), (Q33, 8)])
def test_drop_in_classes_over_wide_moduli(q, d):
    """the reference's classes and functions on a modulus beyond 2^32: constructor checks, + - * (schoolbook-checked), negation,
    norm, weight, transform there and back, ntt_poly_mult, cooley_tukey_ntt / gentleman_sande_intt on lists"""
    import algebra.ntt as N
    from algebra.polynomials import PolynomialCoefficientRepresentation as PC, transform
    root = root_of(q, d)
    inv = pow(root, q - 2, q)
    rnd = random.Random(d)
    half = (q - 1) // 2
    f = [rnd.randint(-half, half) for _ in range(d)]
    g = [rnd.randint(-half, half) for _ in range(d)]
    F, G = PC(q, d, root, inv, 2 * d, list(f)), PC(q, d, root, inv, 2 * d, list(g))
    assert (F + G).coefficients == [cent(x + y, q) for x, y in zip(f, g)]
    assert (F - G).coefficients == [cent(x - y, q) for x, y in zip(f, g)]
    assert (-F).coefficients == [-(x % q) for x in f]
    assert (F * G).coefficients == schoolbook_negacyclic(f, g, q)
    assert F.norm(p="infty") == max(abs(x) for x in f) and F.weight() == sum(1 for x in f if x)
    Fh = transform(F)
    tw, itw = O.py_twiddles(root, q, d), O.py_twiddles(inv, q, d)
    assert Fh.values == O.py_ntt_forward(list(f), q, tw)
    assert transform(Fh).coefficients == f
    assert (Fh * transform(G)).values == [cent(x * y, q) for x, y in zip(Fh.values, transform(G).values)]
    ff, gg = list(f), list(g)
    assert N.ntt_poly_mult(ff, gg, q, root, inv, 2 * d) == schoolbook_negacyclic(f, g, q)
    v = list(f)
    assert N.cooley_tukey_ntt(v, q, 2 * d, tw) == O.py_ntt_forward(list(f), q, tw)
    assert N.gentleman_sande_intt(v, q, 2 * d, itw) == f
    big = PC(q, d, root, inv, 2 * d, [x + 3 * q for x in f])               # values beyond the centred range are reduced on the way in
    assert (big + G).coefficients == (F + G).coefficients


def test_long_transforms_on_a_narrow_modulus_through_the_drop_in():
    """length 8192 over q = 65537 (a modulus of the int32 path, a length beyond its kernels): the generic path serves it"""
    import algebra.ntt as N
    q, d = 65537, 8192
    root = root_of(q, d)
    tw, itw = O.py_twiddles(root, q, d), O.py_twiddles(pow(root, q - 2, q), q, d)
    x = [(7 * i * i + 3) % q - q // 2 for i in range(d)]
    v = list(x)
    assert N.cooley_tukey_ntt(v, q, 2 * d, tw) == O.py_ntt_forward(list(x), q, tw)
    assert N.gentleman_sande_intt(v, q, 2 * d, itw) == [cent(c, q) for c in x]


def test_what_the_wide_path_refuses():
    import fusion_hip
    from fusion_hip.wide import WideContext
    from algebra.polynomials import PolynomialCoefficientRepresentation as PC
    for bad in (prime_near(2 ** 63 + 1, 2), prime_near(2 ** 64 + 1, 2)):
        with pytest.raises(fusion_hip.FusionHipError) as e:
            PC(bad, 1, 1, 1, 1, [5]) + PC(bad, 1, 1, 1, 1, [7])
        assert e.value.code == -2 and "3 <= q < 2^63" in str(e.value) and "no CPU fallback" in str(e.value)
    with pytest.raises(fusion_hip.FusionHipError):
        WideContext(2 ** 40, 8)                                            # even
    lib = fusion_hip.load_library()
    out = np.zeros(4, dtype=np.int64)
    a = np.ones(4, dtype=np.int64)
    from ctypes import POINTER, c_int64
    p = lambda z: z.ctypes.data_as(POINTER(c_int64))
    assert lib.fz_wide_pw_host(0, 2 ** 63 + 1, 1, p(a), p(a), p(out), 4) == -2
    assert b"2^63" in lib.fz_last_error()
    assert lib.fz_wide_ntt_host(0, 97, 12, None, 0, 0, p(a), p(out), 0) == -1     # not a power of two


def test_general_matrix_over_a_wide_modulus():
    """GeneralMatrix + - * (element and (1 x l)(l x 1)), norm, weight on a modulus beyond 2^32 (matrices.py:99-200)"""
    from algebra.matrices import GeneralMatrix
    from algebra.polynomials import PolynomialNTTRepresentation as PN
    q, d, l = Q33, 16, 3
    root = root_of(q, d)
    inv = pow(root, q - 2, q)
    rnd = random.Random(3)
    half = (q - 1) // 2
    mk = lambda: [rnd.randint(1, half) * rnd.choice((-1, 1)) for _ in range(d)]
    a, s = [mk() for _ in range(l)], [mk() for _ in range(l)]
    A = GeneralMatrix(matrix=[[PN(q, d, root, inv, 2 * d, list(v)) for v in a]])
    S = GeneralMatrix(matrix=[[PN(q, d, root, inv, 2 * d, list(v))] for v in s])
    At = GeneralMatrix(matrix=[[PN(q, d, root, inv, 2 * d, list(v))] for v in a])
    prod = A * S
    assert prod.matrix[0][0].values == [cent(sum(a[k][j] * s[k][j] for k in range(l)), q) for j in range(d)]
    assert [z[0].values for z in (At + S).matrix] == [[cent(x + y, q) for x, y in zip(u, v)] for u, v in zip(a, s)]
    assert [z[0].values for z in (-S).matrix] == [[-(x % q) for x in v] for v in s]
    e = PN(q, d, root, inv, 2 * d, mk())
    assert [z[0].values for z in (S * e).matrix] == [[cent(x * y, q) for x, y in zip(v, e.values)] for v in s]


def test_c_caller_of_the_wide_path(tmp_path):
    """examples/wide_roundtrip.c (strict C99, gcc, no HIP headers): a 62-bit prime, the forward transform against a direct evaluation of
    its definition, the inverse, the product, the negation"""
    import subprocess
    from test_cabi_symbols import build_c_example
    r = subprocess.run([build_c_example(tmp_path, "wide_roundtrip")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "wide_roundtrip OK" in r.stdout


@pytest.mark.parametrize("tag", ["g4611686018427322369_64", "g4611686018427322369_1024"])
def test_reference_outputs_for_a_prime_just_below_2_62(tag):
    """tests/golden/generic.npz: what the REFERENCE's functions and classes returned at q = 4611686018427322369 (d = 64 and
    d = 1024) -- transforms both ways, pointwise * + - and negation, the (1 x l)(l x 1) product -- on the generic int64 path"""
    import os
    from fusion_hip.wide import WideContext
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "generic.npz"))
    q, d, root = (int(v) for v in g[f"{tag}_params"])
    assert q == Q62
    ctx = WideContext(q, d, [int(v) for v in g[f"{tag}_tw"]], [int(v) for v in g[f"{tag}_itw"]])
    x = g[f"{tag}_x"]
    assert np.array_equal(ctx.ntt_forward(x), g[f"{tag}_fwd"])
    assert np.array_equal(ctx.ntt_inverse(x), g[f"{tag}_inv"])
    a, b = g[f"{tag}_pw_a"], g[f"{tag}_pw_b"]
    assert np.array_equal(ctx.pw_mul(a, b), g[f"{tag}_pw_mul"])
    assert np.array_equal(ctx.pw_add(a, b), g[f"{tag}_pw_add"])
    assert np.array_equal(ctx.pw_sub(a, b), g[f"{tag}_pw_sub"])
    assert np.array_equal(ctx.pw_neg(a), g[f"{tag}_pw_neg"])
    assert np.array_equal(np.asarray(ctx.matvec(g[f"{tag}_mv_A"], g[f"{tag}_mv_S"])).reshape(g[f"{tag}_mv_out"].shape), g[f"{tag}_mv_out"])


@pytest.mark.parametrize("q,d", [(Q33, 64), (Q62, 64), (Q63, 16)])
def test_unreduced_int64_rows_are_reduced_first(q, d):
    """ANY int64 is a valid input (the int32 contexts accept unreduced rows in the same way): values far outside (-q, q), the
    extremes of the type included, give what the reference's loops give on the same Python integers; the norm of INT64_MIN is 2^63"""
    from fusion_hip.wide import WideContext
    root = root_of(q, d)
    inv = pow(root, q - 2, q)
    tw, itw = O.py_twiddles(root, q, d), O.py_twiddles(inv, q, d)
    ctx = WideContext(q, d, tw, itw)
    rnd = random.Random(q % 977)
    lo, hi = -(2 ** 63), 2 ** 63 - 1
    rows = [[rnd.randint(lo, hi) for _ in range(d)] for _ in range(3)]
    rows.append([lo if k % 2 else hi for k in range(d)])
    rows.append([q * (k % 3) - (k % 2) * (q - 1) for k in range(d)] if q < 2 ** 61 else [q - 1 if k % 2 else -(q - 1) for k in range(d)])
    x = np.array(rows, dtype=np.int64)
    y, z = ctx.ntt_forward(x), ctx.ntt_inverse(x)
    for r in range(len(rows)):
        assert y[r].tolist() == O.py_ntt_forward(list(rows[r]), q, tw), (q, r)
        assert z[r].tolist() == O.py_ntt_inverse(list(rows[r]), q, itw), (q, r)
    a, b = rows[:2], rows[2:4]
    assert ctx.pw_mul(np.array(a), np.array(b)).tolist() == [O.py_pw_mul(u, v, q) for u, v in zip(a, b)]
    assert ctx.pw_add(np.array(a), np.array(b)).tolist() == [O.py_pw_add(u, v, q) for u, v in zip(a, b)]
    assert ctx.pw_sub(np.array(a), np.array(b)).tolist() == [O.py_pw_sub(u, v, q) for u, v in zip(a, b)]
    assert ctx.pw_neg(np.array(a)).tolist() == [O.py_pw_neg(u, q) for u in a]
    assert np.asarray(ctx.matvec(np.array(rows[:3]), np.array([rows[1:4]]))).reshape(-1).tolist() == O.py_matvec(rows[:3], rows[1:4], q)
    mx, wt = ctx.norm_weight(x)
    assert [int(v) for v in mx] == [max(abs(v) for v in r) for r in rows] and int(mx[3]) == 2 ** 63
    assert wt.tolist() == [sum(1 for v in r if v != 0) for r in rows]
