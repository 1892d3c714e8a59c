"""The reference's own unit tests, case for case and under the reference's test names, against the drop-in packages on the MI355X.

Every function below carries the name of ONE test function of the reference (tests/test_ntt.py, test_polynomials.py,
test_matrices.py, test_fusion.py -- 53 functions) and asserts what that function asserts, on the same inputs where the reference
fixes them and on seeded random inputs of the same distribution where it draws them: the (degree, prime) grid is the reference's
TEST_2D_Q_PAIRS (tests/test_ntt.py:18-37: d = 4 .. 64, the doubling walk over the primes q < 2^17 with 2d | q - 1 -- 53 pairs,
rebuilt here by `grid()`), 32 draws per pair as in the reference.  The bodies are written for this repo
(the reference's files do not travel and are not copied); pytest-mock is not installed, so the three mock-based cases use
unittest.mock.  tests/test_gpu_dropin.py / test_host_logic.py hold the consolidated property tests; this file is the
one-to-one map a maintainer of the reference would look for."""
import os
import random
from copy import deepcopy
from math import ceil, log2
from unittest import mock

import pytest

pytestmark = pytest.mark.gpu

SAMPLES = 32                         # the reference's TEST_SAMPLE_SIZE (tests/test_ntt.py:18)
SEED = 8675309                       # the reference's TEST_SEED (tests/test_fusion.py:39)
Q_MAX = 2**17


def grid():
    """the reference's TEST_2D_Q_PAIRS: per degree, walk q = 2d + 1, next prime = 1 (mod 2d), then restart near 2q"""
    from algebra.ntt import is_odd_prime
    out = []
    for d in (4, 8, 16, 32, 64):
        step, q = 2 * d, 2 * d + 1
        while q < Q_MAX:
            while q < Q_MAX and not is_odd_prime(q):
                q += step
            if q < Q_MAX:
                out.append((d, q))
                q = 2 * q - (2 * q - 1) % step
    return out


def first_prime_per_degree():
    seen, out = set(), []
    for d, q in grid():
        if d not in seen:
            seen.add(d)
            out.append((d, q))
    return out


def ring(d, q):
    from algebra.ntt import find_primitive_root, has_primitive_root_of_unity, is_primitive_root
    assert has_primitive_root_of_unity(modulus=q, root_order=2 * d)
    root = find_primitive_root(modulus=q, root_order=2 * d)
    assert is_primitive_root(val=root, modulus=q, root_order=2 * d)
    inv_root = pow(root, q - 2, q)
    assert root * inv_root % q == 1
    return dict(modulus=q, degree=d, root_order=2 * d, root=root, inv_root=inv_root)


def foil(f, g, q):
    d, h = len(f), [0] * (2 * len(f))
    for i, x in enumerate(f):
        for j, y in enumerate(g):
            h[i + j] += x * y
    return [(h[k] - h[k + d]) % q for k in range(d)]


def same_mod(x, y, q):
    return len(x) == len(y) and all((a - b) % q == 0 for a, b in zip(x, y))


def poly(p, coefs):
    from algebra.polynomials import PolynomialCoefficientRepresentation
    return PolynomialCoefficientRepresentation(**p, coefficients=coefs)


def poly_ntt(p, vals):
    from algebra.polynomials import PolynomialNTTRepresentation
    return PolynomialNTTRepresentation(**p, values=vals)


def monomial(rng, d, q):
    c, i = [0] * d, rng.randrange(d)
    c[i] = rng.randrange(1, q)
    return c, i


# ------------------------------------------------------------------------------------------------------------------
# tests/test_ntt.py
# ------------------------------------------------------------------------------------------------------------------
def test_inverse():
    """tests/test_ntt.py:52-57: Fermat inverses over the grid's primes"""
    rng = random.Random(1)
    for _, q in grid():
        x = rng.randrange(1, q)
        assert pow(x, q - 1, q) == 1 and x * pow(x, q - 2, q) % q == 1


def test_ntt_poly_mult_scalars():
    """tests/test_ntt.py:60-118: f * (a constant) for the first prime of every degree"""
    from algebra.ntt import ntt_poly_mult
    rng = random.Random(2)
    for d, q in first_prime_per_degree():
        p = ring(d, q)
        f = [rng.randrange(q) for _ in range(d)]
        g = [rng.randrange(q)] + [0] * (d - 1)
        by_inspection = [x * g[0] % q for x in f]
        assert same_mod(by_inspection, foil(f, g, q), q)
        h = ntt_poly_mult(f=f, g=g, modulus=q, root=p["root"], inv_root=p["inv_root"], root_order=2 * d)
        assert same_mod(h, by_inspection, q)


def test_ntt_poly_mult_monomials():
    """tests/test_ntt.py:121-203: X^i * X^j = +-X^(i + j mod d) for ALL i, j, first prime of every degree"""
    from algebra.ntt import ntt_poly_mult
    for d, q in first_prime_per_degree():
        p = ring(d, q)
        for i in range(d):
            for j in range(d):
                f, g, want = [0] * d, [0] * d, [0] * d
                f[i] = g[j] = 1
                want[(i + j) % d] = -1 if i + j >= d else 1
                h = ntt_poly_mult(f=f, g=g, modulus=q, root=p["root"], inv_root=p["inv_root"], root_order=2 * d)
                assert same_mod(h, want, q), (d, q, i, j)


def test_ntt_poly_mult_scalars_with_monomials():
    """tests/test_ntt.py:206-300: a X^i * b X^j = +-(ab) X^(i + j mod d) for all i, j"""
    from algebra.ntt import ntt_poly_mult
    rng = random.Random(3)
    for d, q in first_prime_per_degree():
        p = ring(d, q)
        for i in range(d):
            for j in range(d):
                f, g, want = [0] * d, [0] * d, [0] * d
                f[i], g[j] = rng.randrange(1, q), rng.randrange(1, q)
                want[(i + j) % d] = f[i] * g[j] * (-1 if i + j >= d else 1)
                h = ntt_poly_mult(f=list(f), g=list(g), modulus=q, root=p["root"], inv_root=p["inv_root"], root_order=2 * d)
                assert same_mod(h, want, q), (d, q, i, j)


def test_poly_mult_simple():
    """tests/test_ntt.py:303-394: q = 17, d = 8, X * (1 + 2X + .. + 8X^7) = -8 + X + .. + 7X^7 by hand, through the in-place
    transforms, and NTT(f) (.) NTT(g) = NTT(f g)"""
    from algebra.ntt import bit_reverse_copy, cent, cooley_tukey_ntt, find_primitive_root, gentleman_sande_intt, ntt_poly_mult
    q, d = 17, 8
    root = find_primitive_root(modulus=q, root_order=2 * d)
    inv_root = pow(root, q - 2, q)
    tw = bit_reverse_copy([root**i for i in range(d)])
    itw = bit_reverse_copy([inv_root**i for i in range(d)])
    f, g = [0, 1, 0, 0, 0, 0, 0, 0], [1, 2, 3, 4, 5, 6, 7, 8]

    def there_and_back(v):
        cooley_tukey_ntt(val=v, modulus=q, root_order=2 * d, bit_rev_root_powers=tw)
        hat = deepcopy(v)
        gentleman_sande_intt(val=v, modulus=q, root_order=2 * d, bit_rev_inv_root_powers=itw)
        return hat
    f_hat, g_hat = there_and_back(f), there_and_back(g)
    assert same_mod(f, [0, 1, 0, 0, 0, 0, 0, 0], q) and same_mod(g, [1, 2, 3, 4, 5, 6, 7, 8], q)
    fg = ntt_poly_mult(f=f, g=g, modulus=q, root=root, inv_root=inv_root, root_order=2 * d)
    by_hand = [-8, 1, 2, 3, 4, 5, 6, 7]
    h = [cent(val=x, modulus=q, halfmod=8, logmod=5) for x in foil(f, g, q)]
    assert same_mod(h, by_hand, q) and same_mod(fg, by_hand, q)
    h_hat = there_and_back(h)
    assert same_mod([cent(val=x * y, modulus=q, halfmod=8, logmod=5) for x, y in zip(f_hat, g_hat)], h_hat, q)


def test_ntt_poly_mult_basic():
    """tests/test_ntt.py:397-429: random f, g against the foiled product, every pair of the grid"""
    from algebra.ntt import ntt_poly_mult
    rng = random.Random(4)
    for d, q in grid():
        p = ring(d, q)
        for _ in range(SAMPLES):
            f, g = [rng.randrange(q) for _ in range(d)], [rng.randrange(q) for _ in range(d)]
            want = foil(f, g, q)
            h = ntt_poly_mult(f=f, g=g, modulus=q, root=p["root"], inv_root=p["inv_root"], root_order=2 * d)
            assert same_mod(h, want, q), (d, q)


def test_ntt_poly_mult_against_one():
    """tests/test_ntt.py:432-465: q = 5, d = 2 -- the smallest ring: 1 * g = g"""
    from algebra.ntt import ntt_poly_mult
    rng = random.Random(5)
    p = ring(2, 5)
    for _ in range(32):
        g = [rng.randrange(5) for _ in range(2)]
        g0 = list(g)
        h = ntt_poly_mult(f=[1, 0], g=g, modulus=5, root=p["root"], inv_root=p["inv_root"], root_order=4)
        assert same_mod(h, g0, 5)


# ------------------------------------------------------------------------------------------------------------------
# tests/test_matrices.py
# ------------------------------------------------------------------------------------------------------------------
def test_is_algebraic_class():
    """tests/test_matrices.py:14-17"""
    from algebra.matrices import is_algebraic_class
    from algebra.polynomials import PolynomialCoefficientRepresentation as Poly, PolynomialNTTRepresentation as PolyNTT
    assert not is_algebraic_class(cls="hello world".__class__())
    assert is_algebraic_class(Poly) and is_algebraic_class(PolyNTT)


def test_general_matrix():
    """tests/test_matrices.py:20-221: 2 x 2 matrices of random monomials, entries and elem_class as given, and
    left * right = the four hand-expanded sums of products"""
    from algebra.matrices import GeneralMatrix
    from algebra.polynomials import PolynomialCoefficientRepresentation as Poly, PolynomialNTTRepresentation as PolyNTT, transform
    rng = random.Random(6)
    for d, q in grid():
        p = ring(d, q)
        for _ in range(2):
            L = [poly(p, monomial(rng, d, q)[0]) for _ in range(4)]
            R = [poly(p, monomial(rng, d, q)[0]) for _ in range(4)]
            for src in (L, R):
                hats = [transform(x=deepcopy(z)) for z in src]
                m = GeneralMatrix(matrix=[[deepcopy(src[0]), deepcopy(src[1])], [deepcopy(src[2]), deepcopy(src[3])]])
                mh = GeneralMatrix(matrix=[[deepcopy(hats[0]), deepcopy(hats[1])], [deepcopy(hats[2]), deepcopy(hats[3])]])
                assert [m.matrix[0][0], m.matrix[0][1], m.matrix[1][0], m.matrix[1][1]] == src and m.elem_class == Poly
                assert [mh.matrix[0][0], mh.matrix[0][1], mh.matrix[1][0], mh.matrix[1][1]] == hats and mh.elem_class == PolyNTT
            a, b, c, e = L
            a2, b2, c2, e2 = R
            want = GeneralMatrix(matrix=[[a * a2 + b * c2, a * b2 + b * e2], [c * a2 + e * c2, c * b2 + e * e2]])
            got = GeneralMatrix(matrix=[[a, b], [c, e]]) * GeneralMatrix(matrix=[[a2, b2], [c2, e2]])
            assert got == want


# ------------------------------------------------------------------------------------------------------------------
# tests/test_polynomials.py
# ------------------------------------------------------------------------------------------------------------------
def test_arithmetic():
    """tests/test_polynomials.py:18-110: a * b (schoolbook face) = foiled product = ntt_poly_mult, and the transform round trips"""
    from algebra.ntt import ntt_poly_mult
    from algebra.polynomials import transform
    rng = random.Random(7)
    for d, q in grid():
        p = ring(d, q)
        for _ in range(SAMPLES):
            fa, fb = [rng.randrange(q) for _ in range(d)], [rng.randrange(q) for _ in range(d)]
            a, b = poly(p, list(fa)), poly(p, list(fb))
            want = poly(p, foil(fa, fb, q))
            c = a * b
            assert same_mod(c.coefficients, want.coefficients, q) and c == want
            other = ntt_poly_mult(f=deepcopy(fa), g=deepcopy(fb), modulus=q, root=p["root"], inv_root=p["inv_root"], root_order=2 * d)
            assert len(other) == d and same_mod(c.coefficients, other, q)
            a_hat, b_hat, c_hat = transform(x=a), transform(x=b), transform(x=c)
            assert transform(x=a_hat) == a and transform(x=b_hat) == b
            assert transform(x=a_hat * b_hat) == c == transform(c_hat)


def test_monomial_products():
    """tests/test_polynomials.py:113-174"""
    from algebra.polynomials import transform
    rng = random.Random(8)
    for d, q in grid():
        p = ring(d, q)
        for _ in range(SAMPLES):
            (fa, i), (fb, j) = monomial(rng, d, q), monomial(rng, d, q)
            want = [0] * d
            want[(i + j) % d] = (fa[i] * fb[j] % q) * (1 - 2 * int(i + j >= d))
            a, b = poly(p, fa), poly(p, fb)
            c = a * b
            assert c == poly(p, want)
            assert transform(x=transform(x=a) * transform(x=b)) == c


def test_poly_init():
    """tests/test_polynomials.py:178-210: the constructor's guards and attributes"""
    from algebra.polynomials import PolynomialCoefficientRepresentation as Poly
    with pytest.raises(ValueError):
        Poly(modulus=1, degree=1, root_order=1, root=1, inv_root=1, coefficients=1)
    with pytest.raises(TypeError):
        Poly(modulus=5, degree=2, root_order=1, root=1, inv_root=1, coefficients=1)
    with pytest.raises(TypeError):
        Poly(modulus=1.0, degree=1, root_order=1, root=1, inv_root=1, coefficients=["hello world"])
    rng = random.Random(9)
    for d, q in grid():
        p = ring(d, q)
        for _ in range(SAMPLES):
            fa = [rng.randrange(q) for _ in range(d)]
            a = poly(p, fa)
            assert (a.modulus, a.degree, a.root_order, a.root, a.inv_root) == (q, d, 2 * d, p["root"], p["inv_root"])
            assert a.coefficients == fa


def _poly_text(p, d, q, fa):
    return (f"PolynomialCoefficientRepresentation(modulus={q}, degree={d}, root={p['root']}, inv_root={p['inv_root']}, "
            f"root_order={2 * d}, coefficients={fa})")


def test_poly_str():
    """tests/test_polynomials.py:213-230"""
    rng = random.Random(10)
    for d, q in grid():
        p = ring(d, q)
        for _ in range(SAMPLES):
            fa = [rng.randrange(q) for _ in range(d)]
            assert str(poly(p, fa)) == _poly_text(p, d, q, fa)


def test_poly_repr():
    """tests/test_polynomials.py:234-251"""
    rng = random.Random(11)
    for d, q in grid():
        p = ring(d, q)
        for _ in range(SAMPLES):
            fa = [rng.randrange(q) for _ in range(d)]
            assert repr(poly(p, fa)) == _poly_text(p, d, q, fa)


def test_poly_eq():
    """tests/test_polynomials.py:255-283: equality is equality of residues -- two objects sharing ONE coefficient list, every
    entry then shifted by random multiples of q (the list is shared: each entry is shifted twice)"""
    rng = random.Random(12)
    for d, q in grid():
        p = ring(d, q)
        for _ in range(SAMPLES):
            a = poly(p, [rng.randrange(q) for _ in range(d)])
            b = poly(p, a.coefficients)
            assert a == a and a == b and len(a.coefficients) == len(b.coefficients)
            for i, x in enumerate(a.coefficients):
                a.coefficients[i] = x + rng.randrange(2**10) * q
                b.coefficients[i] = x + rng.randrange(2**10) * q
            assert a == b


def _attrs_ok(c, p):
    return (c.modulus, c.degree, c.root_order, c.root, c.inv_root) == (p["modulus"], p["degree"], p["root_order"], p["root"], p["inv_root"])


def test_poly_add():
    """tests/test_polynomials.py:287-319"""
    rng = random.Random(13)
    for d, q in grid():
        p = ring(d, q)
        for _ in range(SAMPLES):
            a, b = poly(p, [rng.randrange(1, q) for _ in range(d)]), poly(p, [rng.randrange(1, q) for _ in range(d)])
            c = a + b
            assert _attrs_ok(c, p)
            assert all((c.coefficients[i] - (a.coefficients[i] + b.coefficients[i])) % q == 0 for i in range(d))


def test_poly_sub():
    """tests/test_polynomials.py:323-355"""
    rng = random.Random(14)
    for d, q in grid():
        p = ring(d, q)
        for _ in range(SAMPLES):
            a, b = poly(p, [rng.randrange(1, q) for _ in range(d)]), poly(p, [rng.randrange(1, q) for _ in range(d)])
            c = a - b
            assert _attrs_ok(c, p)
            assert all((c.coefficients[i] - (a.coefficients[i] - b.coefficients[i])) % q == 0 for i in range(d))


def test_poly_mul():
    """tests/test_polynomials.py:359-401: the coefficient-domain product against the foiled one"""
    rng = random.Random(15)
    for d, q in grid():
        p = ring(d, q)
        for _ in range(SAMPLES):
            fa, fb = [rng.randrange(1, q) for _ in range(d)], [rng.randrange(1, q) for _ in range(d)]
            assert poly(p, list(fa)) * poly(p, list(fb)) == poly(p, foil(fa, fb, q))


def test_poly_norm():
    """tests/test_polynomials.py:405-427: the infinity norm is max |stored value|; other norms are not implemented"""
    rng = random.Random(16)
    for d, q in grid():
        p = ring(d, q)
        for _ in range(SAMPLES):
            a = poly(p, [rng.randrange(1, q) for _ in range(d)])
            assert a.norm(p="infty") == max(abs(x) for x in a.coefficients)
            with pytest.raises(NotImplementedError):
                a.norm(p=1)
            with pytest.raises(NotImplementedError):
                a.norm(p=2)


def test_poly_ntt_init():
    """tests/test_polynomials.py:431-495: the NTT-domain constructor's guards; coefficient objects equal up to multiples of q"""
    from algebra.ntt import find_primitive_root
    from algebra.polynomials import PolynomialNTTRepresentation as PolyNTT
    with pytest.raises(ValueError):
        PolyNTT(modulus=2, degree=1, root_order=2, root=1, inv_root=1, values=1)
    with pytest.raises(TypeError):
        PolyNTT(modulus=5, degree=2, root_order=2, root=-1, inv_root=-1, values=1)
    with pytest.raises(ValueError):
        PolyNTT(modulus=5, degree=2, root_order=2, root=1, inv_root=1, values=["hello world"])
    root = find_primitive_root(modulus=5, root_order=2)
    inv_root = pow(root, 5 - 2, 5)
    with pytest.raises(TypeError):
        PolyNTT(modulus=5, degree=2, root_order=2, root=root, inv_root=inv_root, values=["hello world"])
    with pytest.raises(ValueError):
        PolyNTT(modulus=5, degree=2, root_order=2, root=root, inv_root=inv_root, values=[1])
    with pytest.raises(ValueError):
        PolyNTT(modulus=5, degree=2, root_order=2, root=root, inv_root=inv_root, values=[1, 2, 3])
    rng = random.Random(17)
    for d, q in grid():
        p = ring(d, q)
        for _ in range(SAMPLES):
            fa = [rng.randrange(1, q) for _ in range(d)]
            a, b = poly(p, list(fa)), poly(p, list(fa))
            assert a == b and a == a
            for i, x in enumerate(fa):
                a.coefficients[i] = x + rng.randrange(2) * q
                b.coefficients[i] = x + rng.randrange(2) * q
            assert a == b


def test_poly_ntt_str():
    """tests/test_polynomials.py:499-517"""
    rng = random.Random(18)
    for d, q in grid():
        p = ring(d, q)
        for _ in range(SAMPLES):
            va = [rng.randrange(1, q) for _ in range(d)]
            assert str(poly_ntt(p, va)) == (f"PolynomialNTTRepresentation(modulus={q}, degree={d}, root={p['root']}, "
                                            f"inv_root={p['inv_root']}, root_order={2 * d}, values={va})")


def test_poly_ntt_eq():
    """tests/test_polynomials.py:521-548"""
    rng = random.Random(19)
    for d, q in grid():
        p = ring(d, q)
        for _ in range(SAMPLES):
            va = [rng.randrange(1, q) for _ in range(d)]
            a_hat, b_hat = poly_ntt(p, list(va)), poly_ntt(p, list(va))
            assert a_hat == b_hat and a_hat == a_hat
            for i, x in enumerate(va):
                a_hat.values[i] = x + rng.randrange(2) * q
                b_hat.values[i] = x + rng.randrange(2) * q
            assert a_hat == b_hat


def test_poly_ntt_add():
    """tests/test_polynomials.py:552-595: values add pointwise; lists and non-zero integers are refused, 0 is the identity"""
    rng = random.Random(20)
    for d, q in grid():
        p = ring(d, q)
        for _ in range(SAMPLES):
            va, vb = [rng.randrange(1, q) for _ in range(d)], [rng.randrange(1, q) for _ in range(d)]
            a_hat, b_hat = poly_ntt(p, list(va)), poly_ntt(p, list(vb))
            c_hat = a_hat + b_hat
            assert all((z - (x + y)) % q == 0 for x, y, z in zip(va, vb, c_hat.values))
            for bad in (lambda: a_hat + vb, lambda: va + b_hat, lambda: a_hat + vb[0], lambda: va[0] + b_hat, lambda: a_hat + q):
                with pytest.raises(NotImplementedError):
                    bad()
            assert same_mod((a_hat + 0).values, va, q) and same_mod((0 + b_hat).values, vb, q)


def test_poly_ntt_sub():
    """tests/test_polynomials.py:599-641: a - list is a TypeError (the list cannot be negated), list - a, a - int, int - a are
    refused, 0 and 1 behave as identities"""
    rng = random.Random(21)
    for d, q in grid():
        p = ring(d, q)
        for _ in range(SAMPLES):
            va, vb = [rng.randrange(1, q) for _ in range(d)], [rng.randrange(1, q) for _ in range(d)]
            a_hat, b_hat = poly_ntt(p, list(va)), poly_ntt(p, list(vb))
            c_hat = a_hat - b_hat
            assert all((c_hat.values[i] - (va[i] - vb[i])) % q == 0 for i in range(d))
            with pytest.raises(TypeError):
                a_hat - vb
            for bad in (lambda: va - b_hat, lambda: a_hat - vb[0], lambda: va[0] - b_hat, lambda: a_hat - q):
                with pytest.raises(NotImplementedError):
                    bad()
            assert a_hat - 0 == a_hat
            assert b_hat - 0 == b_hat + 0 == b_hat
            assert 0 - b_hat == -(b_hat - 0) == -b_hat
            assert 1 * b_hat == b_hat * 1


def test_poly_ntt_neg():
    """tests/test_polynomials.py:645-661"""
    rng = random.Random(22)
    for d, q in grid():
        p = ring(d, q)
        for _ in range(SAMPLES):
            va = [rng.randrange(1, q) for _ in range(d)]
            b_hat = -poly_ntt(p, list(va))
            assert all((b_hat.values[i] + va[i]) % q == 0 for i in range(d))


def test_poly_ntt_radd():
    """tests/test_polynomials.py:665-704"""
    rng = random.Random(23)
    for d, q in grid():
        p = ring(d, q)
        for _ in range(SAMPLES):
            va, vb = [rng.randrange(1, q) for _ in range(d)], [rng.randrange(1, q) for _ in range(d)]
            a_hat, b_hat = poly_ntt(p, list(va)), poly_ntt(p, list(vb))
            c_hat = b_hat + a_hat
            assert all((c_hat.values[i] - (va[i] + vb[i])) % q == 0 for i in range(d))
            for bad in (lambda: b_hat + va, lambda: vb + a_hat, lambda: b_hat + va[0], lambda: vb[0] + a_hat, lambda: b_hat + q):
                with pytest.raises(NotImplementedError):
                    bad()
            assert b_hat + 0 == b_hat and 0 + a_hat == a_hat


def _ntt_mul_case(rng, d, q, p, swap):
    va, vb = [rng.randrange(1, q) for _ in range(d)], [rng.randrange(1, q) for _ in range(d)]
    a_hat, b_hat = poly_ntt(p, list(va)), poly_ntt(p, list(vb))
    x, y, vx, vy = (b_hat, a_hat, vb, va) if swap else (a_hat, b_hat, va, vb)
    c_hat = x * y
    assert all((c_hat.values[i] - va[i] * vb[i]) % q == 0 for i in range(d))
    with pytest.raises(NotImplementedError):
        x * vy
    with pytest.raises(NotImplementedError):
        vx * y
    assert x * 0 == 0 and 0 * y == 0            # the product with the integer 0 IS the integer 0 (polynomials.py:331-333)
    assert x * 1 == x and 1 * y == y


def test_poly_ntt_mul():
    """tests/test_polynomials.py:708-746"""
    rng = random.Random(24)
    for d, q in grid():
        p = ring(d, q)
        for _ in range(SAMPLES):
            _ntt_mul_case(rng, d, q, p, swap=False)


def test_poly_ntt_rmul():
    """tests/test_polynomials.py:750-788"""
    rng = random.Random(25)
    for d, q in grid():
        p = ring(d, q)
        for _ in range(SAMPLES):
            _ntt_mul_case(rng, d, q, p, swap=True)


def test_transform_2d():
    """tests/test_polynomials.py:792-839: transform refuses what is not a polynomial; INTT(NTT(a) (.) NTT(b)) = the foiled product"""
    from algebra.polynomials import transform
    with pytest.raises(NotImplementedError):
        transform(x="hello, world!")
    rng = random.Random(26)
    for d, q in grid():
        p = ring(d, q)
        for _ in range(SAMPLES):
            fa, fb = [rng.randrange(1, q) for _ in range(d)], [rng.randrange(1, q) for _ in range(d)]
            a_hat, b_hat = transform(x=poly(p, list(fa))), transform(x=poly(p, list(fb)))
            assert transform(x=a_hat * b_hat) == poly(p, foil(fa, fb, q))


def test_comprehensive():
    """tests/test_polynomials.py:842-879: both multiplication routes agree"""
    from algebra.polynomials import transform
    rng = random.Random(27)
    for d, q in grid():
        p = ring(d, q)
        for _ in range(SAMPLES):
            a, b = poly(p, [rng.randrange(1, q) for _ in range(d)]), poly(p, [rng.randrange(1, q) for _ in range(d)])
            assert transform(x=transform(x=a) * transform(x=b)) == a * b


def test_sample_polynomial_coefficient_representation():
    """tests/test_polynomials.py:883-910: q = 65537, d = 1024 (a ring the scheme's prime cannot carry), bounds 1000 / 100"""
    from algebra.ntt import find_primitive_root
    from algebra.polynomials import PolynomialCoefficientRepresentation as Poly, sample_polynomial_coefficient_representation
    q, d = 65537, 1024
    root = find_primitive_root(modulus=q, root_order=2 * d)
    inv_root = pow(root, q - 2, q)
    f = sample_polynomial_coefficient_representation(modulus=q, degree=d, root=root, inv_root=inv_root, root_order=2 * d,
                                                     norm_bound=1000, weight_bound=100, seed=123456789)
    assert isinstance(f, Poly)
    assert (f.modulus, f.degree, f.root, f.inv_root, f.root_order) == (q, d, root, inv_root, 2 * d)
    assert len(f.coefficients) == d and f.norm(p="infty") <= 1000 and f.weight() <= 100


# ------------------------------------------------------------------------------------------------------------------
# tests/test_fusion.py
# ------------------------------------------------------------------------------------------------------------------
def _ring_of(params):
    return dict(modulus=params.modulus, degree=params.degree, root_order=params.root_order, root=params.root, inv_root=params.inv_root)


def test_sample_coefficient_matrix():
    """tests/test_fusion.py:43-83"""
    import fusion.fusion as F
    params = F.fusion_setup(secpar=128, seed=SEED)
    for rows, cols, beta, omega in ((1, 1, 1, 1), (2, 3, 17, 16)):
        x = F.sample_coefficient_matrix(seed=SEED, **_ring_of(params), num_rows=rows, num_cols=cols, norm_bound=beta, weight_bound=omega)
        assert len(x.matrix) == rows and all(len(row) == cols for row in x.matrix)
        assert all(hasattr(z, "norm") and hasattr(z, "weight") for y in x.matrix for z in y)
        assert 0 <= x.norm(p="infty") <= beta and 0 <= x.weight() <= omega


def test_sample_ntt_matrix():
    """tests/test_fusion.py:86-114"""
    import fusion.fusion as F
    params = F.fusion_setup(secpar=128, seed=SEED)
    for rows, cols in ((1, 1), (2, 3)):
        x = F.sample_ntt_matrix(seed=SEED, **_ring_of(params), num_rows=rows, num_cols=cols)
        assert len(x.matrix) == rows and all(len(row) == cols for row in x.matrix)


PARAM_FIELDS = ("capacity", "modulus", "degree", "root_order", "root", "inv_root", "num_rows_pub_challenge", "num_rows_sk",
                "num_rows_vk", "num_cols_pub_challenge", "num_cols_sk", "num_cols_vk", "sign_pre_hash_dst", "sign_hash_dst",
                "agg_xof_dst", "bytes_for_one_coef_bdd_by_beta_ch", "bytes_for_one_coef_bdd_by_beta_ag", "bytes_for_poly_shuffle",
                "beta_sk", "beta_ch", "beta_ag", "beta_vf", "omega_sk", "omega_ch", "omega_ag", "omega_vf")


def test_params_and_fusion_setup():
    """tests/test_fusion.py:117-268: Params(...) and fusion_setup(...) agree with each other and with PREFIX_PARAMETERS on every
    one of the 26 fields the reference compares"""
    import fusion.fusion as F
    for secpar in (128, 256):
        want, got = F.Params(secpar=secpar, seed=SEED), F.fusion_setup(secpar=secpar, seed=SEED)
        assert isinstance(want, F.Params) and isinstance(got, F.Params)
        assert all(isinstance(s, str) for s in (str(want), repr(want), str(got), repr(got)))
        assert want.secpar == secpar == got.secpar
        for field in PARAM_FIELDS:
            assert getattr(want, field) == F.PREFIX_PARAMETERS[secpar][field] == getattr(got, field), field
        assert want == got


def test_key_classes():
    """tests/test_fusion.py:271-305: the key containers hold what they are given and print it"""
    import fusion.fusion as F
    otsk = F.OneTimeSigningKey(seed=None, left_sk_hat="Hello world", right_sk_hat="Goodbye world")
    assert otsk.seed is None and otsk.left_sk_hat == "Hello world" and otsk.right_sk_hat == "Goodbye world"
    assert str(otsk) == "OneTimeSigningKey(seed=None, left_sk_hat=Hello world, right_sk_hat=Goodbye world)" == repr(otsk)
    otvk = F.OneTimeVerificationKey(left_vk_hat="Hello world", right_vk_hat="Goodbye world")
    assert otvk.left_vk_hat == "Hello world" and otvk.right_vk_hat == "Goodbye world"
    assert str(otvk) == "OneTimeVerificationKey(left_vk_hat=Hello world, right_vk_hat=Goodbye world)" == repr(otvk)
    x = (otsk, otvk)
    assert isinstance(x[0], F.OneTimeSigningKey) and isinstance(x[1], F.OneTimeVerificationKey) and x[0] == otsk and x[1] == otvk


def test_keygen():
    """tests/test_fusion.py:308-349: types, the secret rows' norm and weight bounds (through the inverse transform of every row),
    and A * sk = vk for both halves"""
    import fusion.fusion as F
    from algebra.matrices import GeneralMatrix
    from algebra.polynomials import transform
    for secpar in (128, 256):
        params = F.fusion_setup(secpar=secpar, seed=SEED)
        otk = F.keygen(params=params, seed=SEED + 1)
        assert isinstance(otk, tuple) and len(otk) == 2
        otsk, otvk = otk
        assert isinstance(otsk, F.OneTimeSigningKey) and isinstance(otvk, F.OneTimeVerificationKey)
        assert isinstance(otsk.seed, int)
        assert isinstance(otsk.left_sk_hat, GeneralMatrix) and isinstance(otsk.right_sk_hat, GeneralMatrix)
        for half in (otsk.left_sk_hat, otsk.right_sk_hat):
            inv = GeneralMatrix(matrix=[[transform(f) for f in row] for row in half.matrix])
            assert inv.norm(p="infty") <= params.beta_sk and inv.weight() <= params.omega_sk
        assert isinstance(otvk.left_vk_hat, GeneralMatrix) and isinstance(otvk.right_vk_hat, GeneralMatrix)
        assert params.public_challenge * otsk.left_sk_hat == otvk.left_vk_hat
        assert params.public_challenge * otsk.right_sk_hat == otvk.right_vk_hat


def test_signature_challenge_class():
    """tests/test_fusion.py:352-357"""
    import fusion.fusion as F
    x = F.SignatureChallenge(c_hat="Hello world")
    assert isinstance(x, F.SignatureChallenge) and x.c_hat == "Hello world"
    assert str(x) == "SignatureChallenge(c_hat=Hello world)" == repr(x)


def test_signature_class():
    """tests/test_fusion.py:360-365"""
    import fusion.fusion as F
    x = F.Signature(signature_hat="Hello world")
    assert isinstance(x, F.Signature) and x.signature_hat == "Hello world"
    assert str(x) == "Signature(signature_hat=Hello world)" == repr(x)


def test_hash_message_to_int():
    """tests/test_fusion.py:368-394: sha3_256 replaced by a mock -- the function hashes "<dst>,<message>" and reads the digest
    little-endian"""
    import fusion.fusion as F
    for secpar in (128, 256):
        params = F.fusion_setup(secpar=secpar, seed=SEED)
        h = mock.Mock()
        h.digest.return_value = (1234567890).to_bytes(32, byteorder="little")
        with mock.patch("fusion.fusion.sha3_256", return_value=h) as m:
            assert F.hash_message_to_int(params, "my_message") == 1234567890
            m.assert_called_once_with((params.sign_pre_hash_dst.decode("utf-8") + "," + "my_message").encode())


def test_hash_vk_and_int_to_bytes():
    """tests/test_fusion.py:397-435: shake_256 replaced by a mock -- the function hashes "<dst>,<str(vk)>,<int>" """
    import fusion.fusion as F
    for secpar in (128, 256):
        params = F.fusion_setup(secpar=secpar, seed=SEED)
        _, otvk = F.keygen(params, seed=SEED + 1)
        h = mock.Mock()
        h.digest.return_value = b"expected_shake_256_result"
        with mock.patch("fusion.fusion.shake_256", return_value=h) as m:
            assert F.hash_vk_and_int_to_bytes(params=params, key=otvk, i=1234567890, n=1) == b"expected_shake_256_result"
            m.assert_called_once_with((params.sign_hash_dst.decode("utf-8") + "," + str(otvk) + "," + str(1234567890)).encode())


def _decoder_bytes(params, secpar, num_coefs, bound):
    return ceil(params.omega_ch / 8) + ceil((log2(bound) + 1 + secpar) / 8) * num_coefs + params.degree * ceil((log2(params.degree) + secpar) / 8)


def test_decode_bytes_to_polynomial_coefficients():
    """tests/test_fusion.py:438-472: every (weight, norm) bound in 1 .. 11 on random bytes: the decoded polynomial keeps both"""
    import fusion.fusion as F
    for secpar in (128, 256):
        params = F.fusion_setup(secpar=secpar, seed=SEED)
        for omega in range(1, 12):
            for beta in range(1, 12):
                num_coefs, bound = max(0, min(params.degree, omega)), max(0, min(params.modulus // 2, beta))
                n = _decoder_bytes(params, secpar, num_coefs, bound)
                for _ in range(4):
                    y = F.decode_bytes_to_polynomial_coefficients(b=os.urandom(n), log2_bias=secpar, modulus=params.modulus,
                                                                  degree=params.degree, weight_bound=num_coefs, norm_bound=bound)
                    assert len(y) == params.degree
                    assert max(abs(v) for v in y) <= bound and sum(1 for v in y if v % params.modulus) <= num_coefs
                y_as_poly = poly(_ring_of(params), y)            # (the reference measures through the object: once per bound pair)
                assert y_as_poly.norm(p="infty") <= bound and y_as_poly.weight() <= num_coefs


def test_decode_bytes_to_polynomial_coefficient_redux():
    """tests/test_fusion.py:475-558: q = 65537, d = 1024, bounds 1000 / 100, bias 256: the all-zero byte string decodes to
    -X - .. - X^98 - X^1023 and the all-ones pattern to 2 + 2X^2 + .. + 2X^99 + 2X^1023 (worked out by hand in the reference)"""
    import fusion.fusion as F
    q, d, beta, omega, bias = 65537, 1024, 1000, 100, 256
    per_coef, per_step = ceil((log2(beta) + 1 + bias) / 8), ceil((log2(d) + bias) / 8)
    zeros = int("0" * omega, 2).to_bytes(byteorder="big", length=ceil(omega / 8))
    zeros += (0).to_bytes(byteorder="big", length=per_coef) * omega + (0).to_bytes(byteorder="big", length=per_step) * d
    want = [0] * d
    for i in range(1, omega):
        want[i] = -1
    want[-1] = -1
    assert F.decode_bytes_to_polynomial_coefficients(b=zeros, log2_bias=bias, modulus=q, degree=d, norm_bound=beta, weight_bound=omega) == want
    ones = int("1" * omega, 2).to_bytes(byteorder="big", length=ceil(omega / 8))
    ones += (1).to_bytes(byteorder="big", length=per_coef) * omega + (1).to_bytes(byteorder="big", length=per_step) * d
    want = [0] * d
    for i in range(omega):
        want[i] = 2
    want[1], want[-1] = 0, 2
    assert F.decode_bytes_to_polynomial_coefficients(b=ones, log2_bias=bias, modulus=q, degree=d, norm_bound=beta, weight_bound=omega) == want


def _one_hat(params):
    from algebra.polynomials import transform
    one = [1] + [0] * (params.degree - 1)
    one_poly = poly(_ring_of(params), list(one))
    one_hat = transform(one_poly)
    assert transform(one_hat) == one_poly
    return one, one_poly, one_hat


def test_parse_challenge():
    """tests/test_fusion.py:561-604: with the decoder replaced by a mock returning the constant 1, parse_challenge returns NTT(1)"""
    import fusion.fusion as F
    from algebra.polynomials import transform
    for secpar in (128, 256):
        params = F.fusion_setup(secpar=secpar, seed=SEED)
        n = _decoder_bytes(params, secpar, max(0, min(params.degree, params.omega_ch)), max(0, min(params.modulus // 2, params.beta_ch)))
        one, one_poly, one_hat = _one_hat(params)
        with mock.patch("fusion.fusion.decode_bytes_to_polynomial_coefficients", return_value=one):
            got = F.parse_challenge(params=params, b=os.urandom(n))
        assert got == one_hat and transform(got) == one_poly


def test_hash_ch_mocked():
    """tests/test_fusion.py:607-657: hash_ch = SignatureChallenge(parse_challenge(hash_vk_and_int_to_bytes(...))), decoder mocked"""
    import fusion.fusion as F
    for secpar in (128, 256):
        params = F.fusion_setup(secpar=secpar, seed=SEED)
        _, otvk = F.keygen(params, seed=SEED + 1)
        msg = "my_message"
        i = F.hash_message_to_int(params=params, message=msg)
        n = _decoder_bytes(params, params.secpar, max(0, min(params.degree, params.omega_ch)), max(0, min(params.modulus // 2, params.beta_ch)))
        assert len(F.hash_vk_and_int_to_bytes(params=params, key=otvk, i=i, n=n)) >= n
        one, _, one_hat = _one_hat(params)
        with mock.patch("fusion.fusion.decode_bytes_to_polynomial_coefficients", return_value=one):
            assert F.parse_challenge(params=params, b=os.urandom(n)) == one_hat
        with mock.patch("fusion.fusion.decode_bytes_to_polynomial_coefficients", return_value=one):
            assert F.hash_ch(params=params, key=otvk, message=msg) == F.SignatureChallenge(c_hat=one_hat)


def test_hash_ch():
    """tests/test_fusion.py:660-691: the challenge is an NTT-domain polynomial of the scheme's ring whose inverse transform keeps
    beta_ch and omega_ch (asked 32 times, as the reference does)"""
    import fusion.fusion as F
    from algebra.polynomials import PolynomialCoefficientRepresentation as Poly, PolynomialNTTRepresentation as PolyNTT, transform
    for secpar in (128, 256):
        params = F.fusion_setup(secpar=secpar, seed=SEED)
        _, vk = F.keygen(params=params, seed=SEED)
        for _ in range(32):
            ch = F.hash_ch(params=params, key=vk, message="Hello, world!")
            assert isinstance(ch, F.SignatureChallenge) and isinstance(ch.c_hat, PolyNTT)
            assert _attrs_ok(ch.c_hat, _ring_of(params)) and len(ch.c_hat.values) == params.degree
            c = transform(ch.c_hat)
            assert isinstance(c, Poly) and _attrs_ok(c, _ring_of(params)) and len(c.coefficients) == params.degree
            assert c.norm(p="infty") <= params.beta_ch and c.weight() <= params.omega_ch


def test_sign():
    """tests/test_fusion.py:694-731: shape and types of a signature, A * sigma = vk_L * c + vk_R, and the signature's norm / weight
    against the bounds the reference derives"""
    import fusion.fusion as F
    from algebra.matrices import GeneralMatrix
    from algebra.polynomials import PolynomialNTTRepresentation as PolyNTT, transform
    for secpar in (128, 256):
        params = F.fusion_setup(secpar=secpar, seed=SEED)
        omega_v_prime = min(params.degree, params.omega_sk * (1 + params.omega_ch))
        beta_v_prime = params.beta_sk * (1 + min(params.degree, params.omega_sk, params.omega_ch) * params.beta_ch)
        keys = F.keygen(params=params, seed=SEED)
        _, vk = keys
        ch = F.hash_ch(params=params, key=vk, message="Hello, world!")
        sig = F.sign(params=params, key=keys, message="Hello, world!")
        assert isinstance(sig, F.Signature) and isinstance(sig.signature_hat, GeneralMatrix)
        assert len(sig.signature_hat.matrix) == params.num_rows_sk and len(sig.signature_hat.matrix[0]) == params.num_cols_sk
        assert all(isinstance(f, PolyNTT) and len(f.values) == params.degree for row in sig.signature_hat.matrix for f in row)
        assert vk.left_vk_hat * ch.c_hat + vk.right_vk_hat == params.public_challenge * sig.signature_hat
        inv = GeneralMatrix(matrix=[[transform(f) for f in row] for row in sig.signature_hat.matrix])
        assert inv.weight() <= omega_v_prime and inv.norm(p="infty") <= beta_v_prime


def test_aggregation_coefficient_class():
    """tests/test_fusion.py:737-742"""
    import fusion.fusion as F
    x = F.AggregationCoefficient(alpha_hat="Hello world")
    assert isinstance(x, F.AggregationCoefficient) and x.alpha_hat == "Hello world"
    assert str(x) == "AggregationCoefficient(alpha_hat=Hello world)" == repr(x)


def _three_signers(secpar):
    import fusion.fusion as F
    params = F.fusion_setup(secpar=secpar, seed=SEED)
    otks = [F.keygen(params=params, seed=SEED + 1 + k) for k in range(3)]
    msgs = [f"message {k}" for k in range(3)]
    return F, params, otks, [k[1] for k in otks], msgs


def _agg_bytes_per_signer(params):
    """fusion.py:579-586 / :596-603: sign bits + (value bytes + index bytes) per non-zero coefficient"""
    bound = max(0, min(params.modulus // 2, params.beta_ag))
    return ceil(params.omega_ag / 8) + (ceil((log2(bound) + 1 + params.secpar) / 8) + ceil((log2(params.degree) + params.secpar) / 8)) * params.omega_ag


def test_hash_vks_and_ints_and_challs_to_bytes():
    """tests/test_fusion.py:745-746 is an empty body in the reference.  Here: the function is SHAKE-256 of
    "<dst>," + str(list(zip(keys, ints, challenges))), squeezed to the signers' byte budget (fusion.py:573-591)"""
    from hashlib import shake_256
    F, params, _, vks, msgs = _three_signers(128)
    ints = [F.hash_message_to_int(params, m) for m in msgs]
    chs = [F.hash_ch(params, vk, m) for vk, m in zip(vks, msgs)]
    text = params.agg_xof_dst.decode("utf-8") + "," + str(list(zip(vks, ints, chs)))
    got = F.hash_vks_and_ints_and_challs_to_bytes(params=params, keys=vks, prehashed_messages=ints, challenges=chs)
    assert got == shake_256(text.encode()).digest(3 * _agg_bytes_per_signer(params))


def test_decode_bytes_to_agg_coefs():
    """tests/test_fusion.py:749-750 is an empty body in the reference.  Here: one coefficient per whole slice of the bytes
    (fusion.py:594-628), each the transform of the scheme decoder's polynomial for that slice, within beta_ag and omega_ag"""
    from algebra.polynomials import transform
    F, params, _, _, _ = _three_signers(128)
    per = _agg_bytes_per_signer(params)
    blob = os.urandom(3 * per + per // 2)                       # a trailing partial slice is ignored
    got = F.decode_bytes_to_agg_coefs(params=params, b=blob)
    assert len(got) == 3 and all(isinstance(a, F.AggregationCoefficient) for a in got)
    for k, a in enumerate(got):
        c = transform(a.alpha_hat)
        assert c.norm(p="infty") <= params.beta_ag and c.weight() <= params.omega_ag
        assert c == poly(_ring_of(params), F.decode_bytes_to_polynomial_coefficients(
            b=blob[k * per:(k + 1) * per], log2_bias=params.secpar, modulus=params.modulus, degree=params.degree,
            norm_bound=params.beta_ag, weight_bound=params.omega_ag))


def test_hash_ag():
    """tests/test_fusion.py:753-754 is an empty body in the reference.  Here: hash_ag is the composition the reference writes
    (fusion.py:631-652): pre-hashes and challenges of the signers IN THE ORDER GIVEN, one XOF over all of them, one coefficient
    per signer -- so it depends on the order (aggregate and verify sort first)"""
    F, params, _, vks, msgs = _three_signers(128)
    alphas = F.hash_ag(params=params, keys=vks, messages=msgs)
    ints = [F.hash_message_to_int(params, m) for m in msgs]
    chs = [F.hash_ch(params, vk, m) for vk, m in zip(vks, msgs)]
    want = F.decode_bytes_to_agg_coefs(params=params, b=F.hash_vks_and_ints_and_challs_to_bytes(
        params=params, keys=vks, prehashed_messages=ints, challenges=chs))
    assert len(alphas) == 3 and [str(a) for a in alphas] == [str(a) for a in want]
    perm = [2, 0, 1]
    again = F.hash_ag(params=params, keys=[vks[i] for i in perm], messages=[msgs[i] for i in perm])
    assert [str(again[k]) for k in range(3)] != [str(alphas[i]) for i in perm]


def test_aggregate():
    """tests/test_fusion.py:757-758 is an empty body in the reference.  Here: the aggregate is the sum of alpha_i * sigma_i over
    the signers in hash_ag's order (fusion.py:655-677), and it verifies"""
    F, params, otks, vks, msgs = _three_signers(128)
    sigs = [F.sign(params=params, key=k, message=m) for k, m in zip(otks, msgs)]
    agg = F.aggregate(params=params, keys=vks, messages=msgs, signatures=sigs)
    order = sorted(range(3), key=lambda i: str(vks[i]))
    alphas = F.hash_ag(params=params, keys=[vks[i] for i in order], messages=[msgs[i] for i in order])
    want = sigs[order[0]].signature_hat * alphas[0].alpha_hat
    for a, i in zip(alphas[1:], order[1:]):
        want = want + sigs[i].signature_hat * a.alpha_hat
    assert agg.signature_hat == want
    assert F.verify(params=params, keys=vks, messages=msgs, aggregate_signature=agg) == (True, "")


def test_one_sig():
    """tests/test_fusion.py:762-808 (the reference's loop runs secpar 128 twice: `fusion_setup(secpar=128, ...)` inside
    `for next_secpar in [128, 256]`; both levels here)"""
    import fusion.fusion as F
    from algebra.matrices import GeneralMatrix
    for secpar in (128, 256):
        params = F.fusion_setup(secpar=secpar, seed=SEED)
        otk = F.keygen(params=params, seed=SEED + 1)
        sk, vk = otk
        assert params.public_challenge * sk.left_sk_hat == vk.left_vk_hat
        assert params.public_challenge * sk.right_sk_hat == vk.right_vk_hat
        msg = "Hello World"
        ch = F.hash_ch(params=params, key=vk, message=msg)
        sig = F.sign(params=params, key=otk, message=msg)
        assert isinstance(sig, F.Signature) and isinstance(sig.signature_hat, GeneralMatrix)
        assert params.public_challenge * sig.signature_hat == vk.left_vk_hat * ch.c_hat + vk.right_vk_hat
        alpha_hats = F.hash_ag(params=params, keys=[vk], messages=[msg])
        agg_sig = F.aggregate(params=params, keys=[vk], messages=[msg], signatures=[sig])
        assert agg_sig.signature_hat == sig.signature_hat * alpha_hats[0].alpha_hat
        assert params.public_challenge * agg_sig.signature_hat == (vk.left_vk_hat * ch.c_hat + vk.right_vk_hat) * alpha_hats[0].alpha_hat
        ok, why = F.verify(params=params, keys=[vk], messages=[msg], aggregate_signature=agg_sig)
        assert why == "" and ok


def test_many_sigs():
    """tests/test_fusion.py:812-873: 1 .. 4 signers sharing ONE seed (as the reference's loop does: identical keys), aggregate
    verifies, and a change of one value of one entry of the aggregate by a random amount does not"""
    import fusion.fusion as F
    rng = random.Random(28)
    for secpar in (128, 256):
        params = F.fusion_setup(secpar=secpar, seed=SEED)
        for num_keys in range(1, 5):
            otks = [F.keygen(params=params, seed=SEED + 1) for _ in range(num_keys)]
            sks, vks = [k[0] for k in otks], [k[1] for k in otks]
            for sk, vk in zip(sks, vks):
                assert params.public_challenge * sk.left_sk_hat == vk.left_vk_hat
                assert params.public_challenge * sk.right_sk_hat == vk.right_vk_hat
            msgs = ["test_many_sigs_" + str(i) for i in range(num_keys)]
            sigs = [F.sign(params=params, key=k, message=m) for k, m in zip(otks, msgs)]
            agg = F.aggregate(params=params, keys=vks, messages=msgs, signatures=sigs)
            assert F.verify(params=params, keys=vks, messages=msgs, aggregate_signature=agg)[0]
            bad = deepcopy(agg)
            i, j = rng.randrange(len(bad.signature_hat.matrix)), rng.randrange(len(bad.signature_hat.matrix[0]))
            bad.signature_hat.matrix[i][j].values[0] = (bad.signature_hat.matrix[i][j].values[0] + rng.randrange(1, params.modulus)) % params.modulus
            assert not F.verify(params=params, keys=vks, messages=msgs, aggregate_signature=bad)[0]
