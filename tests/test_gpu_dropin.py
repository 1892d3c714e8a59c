"""The properties the reference's own unit tests assert (tests/test_ntt.py, test_polynomials.py,
test_matrices.py, test_fusion.py of the reference), restated against the drop-in package running on
the GPU -- same (degree, prime) grid: d in {4..64}, primes q < 2^17 with 2d | q-1."""
import random

import pytest

pytestmark = pytest.mark.gpu


def pairs():
    from algebra.ntt import is_odd_prime
    out = []
    for d in (4, 8, 16, 32, 64):
        q, found = 2 * d + 1, 0
        while found < 2 and q < 2**17:
            if is_odd_prime(q) and (q - 1) % (2 * d) == 0:
                out.append((d, q))
                found += 1
            q += 2 * d
    return out + [(2, 5)]


def schoolbook(f, g, q):
    d = len(f)
    c = [0] * (2 * d)
    for i, x in enumerate(f):
        for j, y in enumerate(g):
            c[i + j] += x * y
    return [(c[k] - c[k + d]) % q for k in range(d)]


def ring(d, q):
    from algebra.ntt import find_primitive_root
    root = find_primitive_root(modulus=q, root_order=2 * d)
    return dict(modulus=q, degree=d, root=root, inv_root=pow(root, q - 2, q), root_order=2 * d)


def test_ntt_poly_mult_is_negacyclic_product():
    from algebra.ntt import bit_reverse_copy, cooley_tukey_ntt, gentleman_sande_intt, ntt_poly_mult
    rng = random.Random(1)
    for d, q in pairs():
        p = ring(d, q)
        # scalars, monomial pairs (X^i * X^j = +-X^(i+j mod d)), scaled monomials, random polynomials
        cases = [([3] + [0] * (d - 1), [5] + [0] * (d - 1))]
        for i in (0, 1, d // 2, d - 1):
            for j in (0, 1, d - 1):
                f, g = [0] * d, [0] * d
                f[i], g[j] = rng.randrange(1, q), rng.randrange(1, q)
                cases.append((f, g))
        cases += [([rng.randrange(q) for _ in range(d)], [rng.randrange(-q, q) for _ in range(d)]) for _ in range(8)]
        cases.append(([rng.randrange(-2**70, 2**70) for _ in range(d)], [rng.randrange(2**40) for _ in range(d)]))
        for f, g in cases:
            f0, g0 = list(f), list(g)
            fg = ntt_poly_mult(f, g, q, p["root"], p["inv_root"], 2 * d)
            assert [(x - y) % q for x, y in zip(fg, schoolbook(f0, g0, q))] == [0] * d
            assert all(-(q // 2) <= x <= q // 2 for x in fg + f + g)
            assert [(x - y) % q for x, y in zip(f, f0)] == [0] * d          # inputs come back as centred equivalents
            assert [(x - y) % q for x, y in zip(g, g0)] == [0] * d
        tw = bit_reverse_copy([pow(p["root"], i, q) for i in range(d)])
        itw = bit_reverse_copy([pow(p["inv_root"], i, q) for i in range(d)])
        v = [rng.randrange(q) for _ in range(d)]
        v0 = list(v)
        assert cooley_tukey_ntt(v, q, 2 * d, tw) is v                      # in place, returns the same list
        assert gentleman_sande_intt(v, q, 2 * d, itw) is v
        assert [(x - y) % q for x, y in zip(v, v0)] == [0] * d
        # a table that is NOT the power table of one root is used as it stands, like the reference does (ntt.py:277; rounds 1-4
        # raised ValueError here): with all-ones twiddles the network is d/2 ... 1-strided sums and differences
        w = list(v0)
        cooley_tukey_ntt(w, q, 2 * d, [1] * d)
        ref, t, m = list(v0), d, 1
        while m < d:
            t //= 2
            for i in range(m):
                for j in range(2 * i * t, 2 * i * t + t):
                    u, x = ref[j], ref[j + t]
                    ref[j], ref[j + t] = u + x, u - x
            m *= 2
        assert [(x - y) % q for x, y in zip(w, ref)] == [0] * d and all(-(q // 2) <= x <= q // 2 for x in w)


def test_hand_example_q17_d8():
    """(1 + 2X + ... + 8X^7)^2 mod (X^8 + 1, 17) computed by hand-expansion."""
    from algebra.ntt import ntt_poly_mult
    f = list(range(1, 9))
    want = schoolbook(f, f, 17)
    got = ntt_poly_mult(list(f), list(f), 17, 3, 6, 16)
    assert [(x - y) % 17 for x, y in zip(got, want)] == [0] * 8
    assert got == [x - 17 if x > 8 else x for x in want]


def test_polynomial_classes_arithmetic():
    from algebra.polynomials import (PolynomialCoefficientRepresentation as PC, PolynomialNTTRepresentation as PN,
                                     transform)
    rng = random.Random(2)
    for d, q in pairs():
        p = ring(d, q)
        for _ in range(3):
            fa, fb = [rng.randrange(q) for _ in range(d)], [rng.randrange(1, q) for _ in range(d)]
            a, b = PC(**p, coefficients=list(fa)), PC(**p, coefficients=list(fb))
            assert (a + b).coefficients == [((x + y + q // 2) % q) - q // 2 for x, y in zip(fa, fb)]
            assert (-a).coefficients == [-(x % q) for x in fa]
            assert all((z - (x - y)) % q == 0 for x, y, z in zip(fa, fb, (a - b).coefficients))
            assert (a - 0) == a and (0 - a) == -a
            prod = a * b
            assert isinstance(prod, PC) and [x % q for x in prod.coefficients] == schoolbook(fa, fb, q)
            a_hat, b_hat = transform(a), transform(b)
            assert isinstance(a_hat, PN) and transform(a_hat) == a and transform(a_hat).coefficients == \
                [((x + q // 2) % q) - q // 2 for x in fa]
            assert transform(a_hat * b_hat) == prod
            assert all((z - x * y) % q == 0 for x, y, z in zip(a_hat.values, b_hat.values, (a_hat * b_hat).values))
            assert all((z - (x + y)) % q == 0 for x, y, z in zip(a_hat.values, b_hat.values, (a_hat + b_hat).values))
            assert (-a_hat).values == [-(x % q) for x in a_hat.values]
            assert (a_hat - b_hat) == (a_hat + (-b_hat)) and 1 * b_hat == b_hat * 1
            assert a.norm("infty") == max(abs(x) for x in fa) and a.weight() == sum(1 for x in fa if x % q)
            assert a.coefficients == fa and b.coefficients == fb            # operands untouched
            big = PC(**p, coefficients=[x + q * rng.randrange(2**40) for x in fa])
            assert big + b == a + b and big * b == prod and big.weight() == a.weight()
            # stored values outside int32: the reference's max(abs(x)) over the STORED list (polynomials.py:221-224), also
            # through GeneralMatrix.norm (matrices.py:144-148)
            assert big.norm("infty") == max(abs(x) for x in big.coefficients) > 2**31
            from algebra.matrices import GeneralMatrix
            assert GeneralMatrix(matrix=[[a, big], [b, a]]).norm("infty") == big.norm("infty")


def test_sampler_bounds_at_q65537_d1024_like_the_reference():
    """the reference's tests/test_polynomials.py:883-910, verbatim in its assertions: a degree-1024 polynomial over q = 65537
    (a ring the scheme's prime cannot carry: norm / weight run on a ring-only context), norm and weight through the object API"""
    from algebra.polynomials import PolynomialCoefficientRepresentation as Poly, sample_polynomial_coefficient_representation
    p = ring(1024, 65537)
    f = sample_polynomial_coefficient_representation(**p, norm_bound=1000, weight_bound=100, seed=123456789)
    assert isinstance(f, Poly)
    assert (f.modulus, f.degree, f.root, f.inv_root, f.root_order) == (65537, 1024, p["root"], p["inv_root"], 2048)
    assert len(f.coefficients) == 1024
    assert f.norm(p="infty") <= 1000 and f.norm(p="infty") == max(abs(x) for x in f.coefficients)
    assert f.weight() <= 100 and f.weight() == sum(1 for x in f.coefficients if x % 65537)


def test_matrix_products():
    """2x2 matrix of monomials times 2x2 matrix of monomials equals the hand-expanded sums, in the
    coefficient domain (generic element path) and in the NTT domain (batched path)."""
    from algebra.matrices import GeneralMatrix
    from algebra.polynomials import PolynomialCoefficientRepresentation as PC, transform
    rng = random.Random(3)
    for d, q in pairs()[:6]:
        p = ring(d, q)

        def mono():
            c = [0] * d
            c[rng.randrange(d)] = rng.randrange(1, q)
            return c
        A = [[mono(), mono()], [mono(), mono()]]
        B = [[mono(), mono()], [mono(), mono()]]
        want = [[[(x + y) % q for x, y in zip(schoolbook(A[i][0], B[0][j], q), schoolbook(A[i][1], B[1][j], q))]
                 for j in range(2)] for i in range(2)]
        Ac = GeneralMatrix(matrix=[[PC(**p, coefficients=list(c)) for c in row] for row in A])
        Bc = GeneralMatrix(matrix=[[PC(**p, coefficients=list(c)) for c in row] for row in B])
        C = Ac * Bc
        assert [[[x % q for x in z.coefficients] for z in row] for row in C.matrix] == want
        Ah = GeneralMatrix(matrix=[[transform(z) for z in row] for row in Ac.matrix])
        Bh = GeneralMatrix(matrix=[[transform(z) for z in row] for row in Bc.matrix])
        Ch = Ah * Bh
        assert [[[x % q for x in transform(z).coefficients] for z in row] for row in Ch.matrix] == want
        assert all(-(q // 2) <= x <= q // 2 for row in Ch.matrix for z in row for x in z.values)
        s = Ah.matrix[0][1]
        assert (Bh * s).matrix[1][0] == Bh.matrix[1][0] * s and (Ah + Bh).matrix[0][0] == Ah.matrix[0][0] + Bh.matrix[0][0]
        assert (Ah - Bh) == Ah + (-Bh) and Ac.norm("infty") == max(max(c) for row in A for c in row) and Ac.weight() == 1


@pytest.mark.parametrize("secpar", [128, 256])
def test_scheme_properties(secpar):
    """keygen: INTT(sk) within (beta_sk, omega_sk) and A.sk == vk; sign: A.sigma == vk_L*c + vk_R with
    bounded INTT(sigma); challenge weights (reference tests/test_fusion.py:308-349, :660-731)."""
    import fusion.fusion as F
    from algebra.polynomials import transform
    params = F.fusion_setup(secpar, 8675309)
    sk, vk = F.keygen(params, 1234)
    for m in (sk.left_sk_hat, sk.right_sk_hat):
        coef = [transform(z) for row in m.matrix for z in row]
        assert len(coef) == params.num_rows_sk
        assert all(c.norm("infty") <= params.beta_sk and c.weight() <= params.omega_sk for c in coef)
    A = params.public_challenge
    assert A * sk.left_sk_hat == vk.left_vk_hat and A * sk.right_sk_hat == vk.right_vk_hat
    msg = "the quick brown fox"
    c = F.hash_ch(params, vk, msg)
    cc = transform(c.c_hat)
    assert cc.norm("infty") <= params.beta_ch and cc.weight() == params.omega_ch
    sig = F.sign(params, (sk, vk), msg)
    assert A * sig.signature_hat == vk.left_vk_hat * c.c_hat + vk.right_vk_hat
    bound = params.beta_sk * (1 + min(params.degree, params.omega_ch) * params.beta_ch)
    assert all(transform(z).norm("infty") <= bound for row in sig.signature_hat.matrix for z in row)
    alphas = F.hash_ag(params, [vk], [msg])
    assert len(alphas) == 1 and transform(alphas[0].alpha_hat).weight() == params.omega_ag
    agg = F.aggregate(params, [vk], [msg], [sig])
    assert F.verify(params, [vk], [msg], agg) == (True, "")
    assert F.verify(params, [vk], [msg + "!"], agg) == (False, "Target doesn't match image of aggregate signature.")


@pytest.mark.parametrize("secpar", [128, 256])
def test_lists_are_built_on_demand_and_stay_authoritative(secpar):
    """What the library produces is array-backed until somebody reads `.values` / `.coefficients` (algebra/polynomials.py,
    storage note); what a caller can observe is the reference's: lists of Python ints that may be mutated in place
    (reference tests/test_polynomials.py:280-283, tests/test_fusion.py:861-867) and are what every later call uses."""
    import copy
    import fusion.fusion as F
    from algebra.polynomials import transform
    params = F.fusion_setup(secpar, 77)
    sk, vk = F.keygen(params, 4321)
    z = sk.left_sk_hat.matrix[3][0]
    assert z._arr is not None and z._list is None                  # nobody asked for the list yet
    msg = "a message"
    sig = F.sign(params, (sk, vk), msg)
    assert z._arr is not None                                      # sign() read the rows, not lists
    assert sig.signature_hat.matrix[0][0]._arr is not None
    agg = F.aggregate(params, [vk], [msg], [sig])
    assert F.verify(params, [vk], [msg], agg) == (True, "")
    assert all(p._arr is not None for row in sig.signature_hat.matrix for p in row)      # still no lists of the signature
    # reading builds the list once; the same object comes back every time; its elements are Python ints
    vals = z.values
    assert vals is z.values and type(vals) is list and all(type(v) is int for v in vals) and z._arr is None
    assert len(vals) == params.degree and "values=[" + str(vals[0]) + "," in str(z)
    # a copy made BEFORE a mutation is independent of it (array-backed siblings share rows, lists are never shared)
    twin = copy.deepcopy(sk)
    before = F.sign(params, (sk, vk), msg)
    vals[0] = (vals[0] + 1) % 7
    after = F.sign(params, (sk, vk), msg)
    assert after.signature_hat.matrix[3][0] != before.signature_hat.matrix[3][0]      # the mutated list is what sign() used
    assert after.signature_hat.matrix[2][0] == before.signature_hat.matrix[2][0]
    assert F.sign(params, (twin, vk), msg).signature_hat == before.signature_hat
    # assignment replaces the data; in-place element mutation of a signature changes the verdict (test_fusion.py:861-867)
    z.values = list(twin.left_sk_hat.matrix[3][0].values)
    assert F.sign(params, (sk, vk), msg).signature_hat == before.signature_hat
    agg.signature_hat.matrix[0][0].values[0] += 1
    assert F.verify(params, [vk], [msg], agg) == (False, "Target doesn't match image of aggregate signature.")
    # the entries of a seeded sample are independent objects although they start from one row
    A = params.public_challenge
    a0, a1 = A.matrix[0][0], A.matrix[0][1]
    assert a0 == a1 and a0 is not a1
    a1.values[5] += 1
    assert a0 != a1 and A.matrix[0][2] == a0
    a1.values[5] -= 1
    # transform results are array-backed too and equal their list-backed selves
    c = transform(a0)
    assert c._arr is not None and transform(c) == a0 and c.coefficients == list(c.coefficients)


def test_keygen_leaves_the_global_generator_where_the_reference_does():
    """keygen(params, seed) samples ONE polynomial per half and lets the kernel read it for all l rows; the reference samples
    every entry with the same seed (fusion.py:156-173, :338-362), which leaves `random` in the state one seeded call leaves"""
    import fusion.fusion as F
    params = F.fusion_setup(128, 5)
    F.keygen(params, 99)
    state = random.getstate()
    F.sample_coefficient_matrix(seed=100, modulus=params.modulus, degree=params.degree, root_order=params.root_order,
                                root=params.root, inv_root=params.inv_root, num_rows=params.num_rows_sk,
                                num_cols=params.num_cols_sk, norm_bound=params.beta_sk, weight_bound=params.omega_sk)
    assert random.getstate() == state
    with pytest.raises(TypeError):
        F.keygen(params, None)
    # seeds the C clone does not take go through the Python sampler: same contract (random.seed takes abs() of a negative int)
    sk_n, vk_n = F.keygen(params, -7)
    state = random.getstate()
    F.sample_coefficient_matrix(seed=-6, modulus=params.modulus, degree=params.degree, root_order=params.root_order,
                                root=params.root, inv_root=params.inv_root, num_rows=params.num_rows_sk,
                                num_cols=params.num_cols_sk, norm_bound=params.beta_sk, weight_bound=params.omega_sk)
    assert random.getstate() == state
    sk_p, vk_p = F.keygen(params, 7)                       # abs(-7) = 7 seeds the left half ...
    assert sk_n.left_sk_hat == sk_p.left_sk_hat and sk_n.right_sk_hat != sk_p.right_sk_hat      # ... -6 -> 6, not 8, the right one
    assert params.public_challenge * sk_n.right_sk_hat == vk_n.right_vk_hat
    # a replaced generator function is honoured (the fast path steps aside): every draw 0 -> magnitude 1, sign +1
    from unittest import mock
    with mock.patch("algebra.polynomials.randrange", return_value=0):
        sk_m, _ = F.keygen(params, 5)
    from algebra.polynomials import transform
    assert transform(sk_m.left_sk_hat.matrix[0][0]).coefficients == [1] * params.degree


@pytest.mark.parametrize("secpar", [128, 256])
def test_replaced_module_functions_are_honoured(secpar):
    """The reference's tests replace fusion.fusion.sha3_256, shake_256 and decode_bytes_to_polynomial_coefficients
    (tests/test_fusion.py:368-435, :559-657, with pytest-mock) and expect hash_message_to_int, hash_vk_and_int_to_bytes,
    parse_challenge and hash_ch to go through the replacements.  The drop-in's shortcuts (C decoder, batched challenge
    transforms) must step aside for them -- restated with unittest.mock."""
    from unittest import mock
    import fusion.fusion as F
    from algebra.polynomials import PolynomialCoefficientRepresentation, transform
    params = F.fusion_setup(secpar, 20240101)
    sk, vk = F.keygen(params, 20240102)
    # sha3_256
    digest = mock.Mock()
    digest.digest.return_value = (1234567890).to_bytes(32, byteorder="little")
    with mock.patch("fusion.fusion.sha3_256", return_value=digest) as m:
        assert F.hash_message_to_int(params, "my_message") == 1234567890
        m.assert_called_once_with((params.sign_pre_hash_dst.decode("utf-8") + ",my_message").encode())
    # shake_256
    xof = mock.Mock()
    xof.digest.return_value = b"expected_shake_256_result"
    with mock.patch("fusion.fusion.shake_256", return_value=xof) as m:
        assert F.hash_vk_and_int_to_bytes(params=params, key=vk, i=1234567890, n=1) == b"expected_shake_256_result"
        m.assert_called_once_with((params.sign_hash_dst.decode("utf-8") + "," + str(vk) + ",1234567890").encode())
    # the decoder: parse_challenge and hash_ch must return the transform of what the REPLACEMENT decodes
    one = [1] + [0] * (params.degree - 1)
    one_hat = transform(PolynomialCoefficientRepresentation(
        modulus=params.modulus, degree=params.degree, root=params.root, inv_root=params.inv_root,
        root_order=params.root_order, coefficients=list(one)))
    n = F._challenge_bytes_needed(params)
    real = F.hash_ch(params, vk, "my_message")
    assert real.c_hat != one_hat
    with mock.patch("fusion.fusion.decode_bytes_to_polynomial_coefficients", return_value=list(one)) as m:
        assert F.parse_challenge(params=params, b=bytes(n)) == one_hat and m.call_count == 1
        assert F.hash_ch(params=params, key=vk, message="my_message") == F.SignatureChallenge(c_hat=one_hat)
        # aggregate() and verify() reach the decoder through hash_ch / hash_ag: they see the replacement too
        alphas = F.hash_ag(params, [vk], ["my_message"])
        assert alphas[0].alpha_hat == one_hat
    # ... and the shortcuts are back once the replacement is gone
    assert F.hash_ch(params, vk, "my_message") == real
    # a replaced hash_ch is what aggregate() / verify() use for their challenges
    sig = F.sign(params, (sk, vk), "my_message")
    agg = F.aggregate(params, [vk], ["my_message"], [sig])
    assert F.verify(params, [vk], ["my_message"], agg) == (True, "")
    with mock.patch("fusion.fusion.hash_ch", return_value=F.SignatureChallenge(c_hat=one_hat)) as m:
        assert F.verify(params, [vk], ["my_message"], agg)[0] is False and m.call_count >= 1
