"""Host-side logic of the drop-in package that needs no GPU: parameter predicates, scalar cent,
container validation and string formats, error classes, seeded samplers, SHA3/SHAKE pipeline and the
byte decoder -- pinned by the reference's reproducible KAT rows (tests/golden/kat.json) and by the
properties the reference's own tests assert (tests/test_ntt.py, test_polynomials.py,
test_matrices.py, test_fusion.py of the reference)."""
import hashlib
import json
import os
import random
from math import ceil, log2

import pytest

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def kat():
    with open(os.path.join(G, "kat.json")) as fh:
        return json.load(fh)


# ---- algebra.ntt predicates and scalar helpers -------------------------------------------------
def test_predicates():
    from algebra import ntt
    primes = [3, 5, 7, 17, 257, 65537, 12289, 2147465729]
    assert all(ntt.is_odd_prime(p) for p in primes)
    assert not any(ntt.is_odd_prime(v) for v in (0, 1, 2, 9, 15, 21, 25, 2147465729 * 3, -7, 11.5, "7"))
    # the reference never divides by 2: even numbers without a small odd factor pass (ntt.py:30)
    assert ntt.is_odd_prime(4) and ntt.is_odd_prime(8) and ntt.is_odd_prime(6) and not ntt.is_odd_prime(18)
    assert ntt.has_primitive_root_of_unity(17, 16) and not ntt.has_primitive_root_of_unity(17, 32)
    assert not ntt.has_primitive_root_of_unity(2, 1) and not ntt.has_primitive_root_of_unity(17, 1)
    assert [ntt.is_pow_two_geq_two(v) for v in (1, 2, 3, 4, 256, 0, -2, 6.5)] == [False, True, False, True, True, False, False, False]
    assert ntt.is_root_of_unity(16, 17, 2) and not ntt.is_root_of_unity(3, 17, 8)
    assert ntt.is_primitive_root(3, 17, 16) and not ntt.is_primitive_root(16, 17, 16) and not ntt.is_primitive_root(9, 17, 16)
    assert ntt.find_primitive_root(17, 16) == 3 and ntt.find_primitive_root(5, 4) == 2
    assert ntt.find_primitive_root(17, 32) is None
    q = 2147465729
    assert pow(3337519, 256, q) == q - 1 and pow(23584283, 64, q) == q - 1
    assert ntt.is_primitive_root(3337519, q, 512) and ntt.is_primitive_root(23584283, q, 128)


def test_bit_reverse_copy_and_cent():
    from algebra import ntt
    assert ntt.bit_reverse_copy(list(range(8))) == [0, 4, 2, 6, 1, 5, 3, 7]
    assert ntt.bit_reverse_copy(["a", "b"]) == ["a", "b"]
    nested = [[1], [2], [3], [4]]
    out = ntt.bit_reverse_copy(nested)
    out[1].append(9)
    assert nested == [[1], [2], [3], [4]]            # elements are copied
    with pytest.raises(ValueError):
        ntt.bit_reverse_copy((1, 2))
    q = 2147465729
    h = q // 2
    for lm in (31, 32, 30 + 1):                      # the three logmod conventions of the reference agree
        for v, want in ((0, 0), (h, h), (h + 1, h + 1 - q), (q - 1, -1), (q, 0), (-1, -1), (-h, -h), (-h - 1, h),
                        (7 * q + 5, 5), (-(2**62), (-(2**62)) % q - q if (-(2**62)) % q > h else (-(2**62)) % q)):
            assert ntt.cent(v, q, h, lm) == want
    for q in (5, 17, 257):
        for v in range(-3 * q, 3 * q):
            z = ntt.cent(v, q, q // 2, q.bit_length() - 1)
            assert (z - v) % q == 0 and -(q // 2) <= z <= q // 2
    with pytest.raises(TypeError):
        ntt.cent(1.0, 5, 2, 2)
    for bad in ((1, 1, 1, 1), (1, 5, 0, 2), (1, 5, 2, 0)):
        with pytest.raises(ValueError):
            ntt.cent(*bad)


def test_transform_argument_errors_need_no_gpu():
    """Validation order and classes of cooley_tukey_ntt / gentleman_sande_intt (ntt.py:239-270)."""
    from algebra import ntt
    tw = ntt.bit_reverse_copy([pow(3, i, 17) for i in range(8)])
    for fn in (ntt.cooley_tukey_ntt, ntt.gentleman_sande_intt):
        with pytest.raises(TypeError):
            fn((1, 2), 17, 16, tw)
        with pytest.raises(TypeError):
            fn([1] * 8, 17.0, 16, tw)
        with pytest.raises(TypeError):
            fn([1] * 8, 17, 16, tuple(tw))
        with pytest.raises(TypeError):
            fn([1] * 8, 17, 16, [1.5] * 8)
        with pytest.raises(TypeError):
            fn([1] * 8, 17, "16", tw)
        with pytest.raises(TypeError):
            fn([1.0] * 8, 17, 16, tw)
        with pytest.raises(ValueError):
            fn([1] * 8, 15, 16, tw)               # not an odd prime
        with pytest.raises(ValueError):
            fn([1] * 8, 17, 32, tw)               # no root of that order
        with pytest.raises(ValueError):
            fn([1] * 6, 17, 16, tw)               # length not a power of two
        with pytest.raises(ValueError):
            fn([1] * 8, 17, 4, tw)                # root order neither n nor 2n
        with pytest.raises(NotImplementedError):
            fn([1] * 8, 17, 8, tw)                # root order == degree
    with pytest.raises(ValueError):
        ntt.ntt_poly_mult([1] * 8, [1] * 8, 17, 2, 9, 16)          # 2 is not primitive of order 16
    with pytest.raises(ValueError):
        ntt.ntt_poly_mult([1] * 8, [1] * 4, 17, 3, 6, 16)
    with pytest.raises(ValueError):
        ntt.ntt_poly_mult([1] * 8, [1] * 8, 17, 3, 5, 16)          # 3*5 != 1
    with pytest.raises(ValueError):
        ntt.ntt_poly_mult_half([1] * 16, [1] * 16, 17, 3, 6, 16)   # dead code of the reference: always ValueError


# ---- algebra.polynomials / algebra.matrices containers ------------------------------------------
def _params(q=17, d=8):
    from algebra.ntt import find_primitive_root
    root = find_primitive_root(q, 2 * d)
    return dict(modulus=q, degree=d, root=root, inv_root=pow(root, q - 2, q), root_order=2 * d)


def test_polynomial_constructors_and_strings():
    from algebra.polynomials import (PolynomialCoefficientRepresentation as PC, PolynomialNTTRepresentation as PN,
                                     PolynomialRepresentation as PR, transform)
    with pytest.raises(ValueError):
        PC(modulus=1, degree=1, root_order=1, root=1, inv_root=1, coefficients=1)
    with pytest.raises(TypeError):
        PC(modulus=5, degree=2, root_order=1, root=1, inv_root=1, coefficients=1)
    with pytest.raises(TypeError):
        PC(modulus=1.0, degree=1, root_order=1, root=1, inv_root=1, coefficients=["hello world"])
    p = _params()
    with pytest.raises(ValueError):
        PR(**{**p, "root_order": 5})                           # does not divide q - 1
    with pytest.raises(ValueError):
        PR(**{**p, "root": 2})                                 # 2^16 = 1 but 2 is not primitive of order 16 ... or not a root
    with pytest.raises(ValueError):
        PR(**{**p, "root": 16})                                # order 2, not primitive
    with pytest.raises(ValueError):
        PR(**{**p, "inv_root": p["inv_root"] + 1})
    for cls, field in ((PC, "coefficients"), (PN, "values")):
        with pytest.raises(TypeError):
            cls(**p, **{field: tuple(range(8))})
        with pytest.raises(TypeError):
            cls(**p, **{field: [0.5] * 8})
        with pytest.raises(ValueError):
            cls(**p, **{field: [1] * 7})
    vals = [3, -4, 5, 0, 16, 17, -20, 1]
    a = PC(**p, coefficients=list(vals))
    assert str(a) == repr(a) == (f"PolynomialCoefficientRepresentation(modulus=17, degree=8, root={p['root']}, "
                                 f"inv_root={p['inv_root']}, root_order=16, coefficients={vals})")
    b = PN(**p, values=list(vals))
    assert str(b) == repr(b) == (f"PolynomialNTTRepresentation(modulus=17, degree=8, root={p['root']}, "
                                 f"inv_root={p['inv_root']}, root_order=16, values={vals})")
    assert (a.halfmod, a.logmod) == (8, 4)
    # equality is equality of residues, also for non-reduced storage (test_polynomials.py:255-283 of the reference)
    a2 = PC(**p, coefficients=[v + 17 * k for k, v in enumerate(vals)])
    b2 = PN(**p, values=[v - 34 * k for k, v in enumerate(vals)])
    assert a == a2 and b == b2 and not (a == b) and not (a == 3)
    assert PN(**p, values=[0, 17, -34, 0, 0, 0, 0, 51]) == 0 and not (b == 0)
    assert not (PN(**{**p, "degree": 8}, values=list(vals)) == PN(**_params(97, 8), values=list(vals)))
    # identities that short-circuit before any arithmetic (polynomials.py:115-116, :172-175, :283-284, :342-345)
    assert a + 0 is a and 0 + a is a and a * 1 is a and 1 * a is a and a * 0 == 0 and 0 * a == 0
    assert b + 0 is b and 0 + b is b and b * 1 is b and 1 * b is b and b * 0 == 0 and 0 * b == 0
    zero_hat = PN(**p, values=[0] * 8)
    assert b + zero_hat is b and b * zero_hat == 0 and isinstance(b * zero_hat, int)
    for bad in ([1] * 8, 3, "x"):
        with pytest.raises(NotImplementedError):
            a + bad
        with pytest.raises(NotImplementedError):
            b * bad
    with pytest.raises(TypeError):
        b - [1] * 8                                            # unary minus on a list
    with pytest.raises(NotImplementedError):
        b + PN(**_params(97, 8), values=list(vals))            # different modulus
    with pytest.raises(NotImplementedError):
        a * PC(**_params(17, 4), coefficients=[1, 2, 3, 4])    # different degree
    with pytest.raises(NotImplementedError):
        a.norm(p=2)
    with pytest.raises(NotImplementedError):
        transform([1, 2, 3])


def test_general_matrix_container():
    from algebra.matrices import GeneralMatrix, is_algebraic_class
    from algebra.polynomials import PolynomialNTTRepresentation as PN
    assert is_algebraic_class(int) and is_algebraic_class(PN) and not is_algebraic_class(str) and not is_algebraic_class(dict)
    for bad in (3, [], [1, 2], [[]], [[1, 2], [3]], [["a"]], [[1, 2.0]]):
        with pytest.raises(ValueError):
            GeneralMatrix(matrix=bad)
    rows = [[1, 2], [3, 4]]
    m = GeneralMatrix(matrix=rows)
    assert m.matrix is rows and len(m) == 2 and m[1] == [3, 4] and list(iter(m)) == rows
    assert str(m) == repr(m) == "GeneralMatrix(elem_class=<class 'int'>, matrix=[[1, 2], [3, 4]])"
    # generic element path: plain integers
    n = GeneralMatrix(matrix=[[5, 6], [7, 8]])
    assert (m + n).matrix == [[6, 8], [10, 12]] and (m - n).matrix == [[-4, -4], [-4, -4]]
    assert (m * n).matrix == [[19, 22], [43, 50]] and (m * 3).matrix == [[3, 6], [9, 12]]
    assert (-m).matrix == [[-1, -2], [-3, -4]] and (m % 3).matrix == [[1, 2], [0, 1]]
    assert m + 0 is m and 0 + m is m and m == GeneralMatrix(matrix=[[1, 2], [3, 4]]) and not (m == n)
    assert GeneralMatrix(matrix=[[0, 0]]) == 0 and not (m == 0)
    with pytest.raises(ValueError):
        m + GeneralMatrix(matrix=[[1, 2]])
    with pytest.raises(ValueError):
        m * GeneralMatrix(matrix=[[1, 2]])
    with pytest.raises(NotImplementedError):
        m + GeneralMatrix(matrix=[[1.0, 2.0], [3.0, 4.0]])
    with pytest.raises(TypeError):
        m * GeneralMatrix(matrix=[[1.0, 2.0], [3.0, 4.0]])
    with pytest.raises(TypeError):
        m % 2.5
    with pytest.raises(ValueError):
        m % 1
    with pytest.raises(NotImplementedError):
        m.norm("infty")
    m[0] = [9, 9]
    assert rows[0] == [9, 9]
    del m[1]
    assert m.matrix[1] == 0
    p = _params()
    pm = GeneralMatrix(matrix=[[PN(**p, values=[1] * 8)]])
    assert str(pm).startswith("GeneralMatrix(elem_class=<class 'algebra.polynomials.PolynomialNTTRepresentation'>, "
                              "matrix=[[PolynomialNTTRepresentation(modulus=17, degree=8, ")


# ---- fusion.fusion: tables, samplers, hashing, decoder ----------------------------------------------
def test_parameter_tables():
    import fusion.fusion as F
    p128, p256 = F.PREFIX_PARAMETERS[128], F.PREFIX_PARAMETERS[256]
    assert (p128["degree"], p128["num_rows_sk"], p128["capacity"], p128["omega_ch"], p128["omega_ag"]) == (64, 195, 1796, 27, 35)
    assert (p256["degree"], p256["num_rows_sk"], p256["capacity"], p256["omega_ch"], p256["omega_ag"]) == (256, 83, 2818, 60, 60)
    assert p128["beta_vf"] == 536070080 and p256["beta_vf"] == 536321760
    assert p128["inv_root"] == 540632852 and p256["inv_root"] == 1978410468
    assert (p128["bytes_for_one_coef_bdd_by_beta_ch"], p256["bytes_for_one_coef_bdd_by_beta_ch"]) == (17, 33)
    assert (p128["bytes_for_poly_shuffle"], p256["bytes_for_poly_shuffle"]) == (1088, 8448)
    assert (p128["sign_pre_hash_dst"], p128["sign_hash_dst"], p128["agg_xof_dst"]) == (b"\x01\x00", b"\x01\x01", b"\x01\x02")
    assert (p256["sign_pre_hash_dst"], p256["sign_hash_dst"], p256["agg_xof_dst"]) == (b"\x03\x00", b"\x03\x01", b"\x03\x02")
    assert F._challenge_bytes_needed(F.fusion_setup(128, 1)) == 1551
    assert F._challenge_bytes_needed(F.fusion_setup(256, 1)) == 10436
    assert F._agg_coef_bytes(F.fusion_setup(128, 1)) == 1195 and F._agg_coef_bytes(F.fusion_setup(256, 1)) == 3968
    empty = F.Params(secpar=512, seed=1)                      # unknown secpar: attribute-less object
    assert not hasattr(empty, "modulus")
    assert F.fusion_setup(128, 7) == F.fusion_setup(128, 7) and not (F.fusion_setup(128, 7) == F.fusion_setup(128, 8))


def test_setup_kat_and_samplers(kat):
    """fusion_setup_KAT_{128,256}.csv: pins the NTT-domain sampler and every str() format."""
    import fusion.fusion as F
    for row in kat["setup"]:
        params = F.fusion_setup(row["secpar"], row["seed"])
        assert hashlib.sha256(str(params).encode()).hexdigest() == row["sha256_str_params"]
        polys = [z for y in params.public_challenge.matrix for z in y]
        assert len(polys) == row["n_polys"] and polys[0].values == row["first_poly"]
        assert all(z.values == row["first_poly"] for z in polys)          # same seed for every entry
    from algebra.polynomials import sample_polynomial_coefficient_representation as samp
    p = _params(65537, 1024)
    f = samp(**p, norm_bound=1000, weight_bound=100, seed=123456789)
    assert len(f.coefficients) == 1024 and max(abs(v) for v in f.coefficients) <= 1000
    assert sum(1 for v in f.coefficients if v) == 100
    g = samp(**p, norm_bound=1000, weight_bound=100, seed=123456789)
    assert f.coefficients == g.coefficients
    random.seed(5)
    state = random.getstate()
    samp(**p, norm_bound=3, weight_bound=2000, seed=None)          # seed=None continues the global stream
    assert random.getstate() != state


def test_hash_kats(kat):
    import fusion.fusion as F
    from algebra.matrices import GeneralMatrix
    from algebra.polynomials import PolynomialNTTRepresentation as PN
    params = F.fusion_setup(128, 1)
    for row in kat["hash_message_to_int"]:
        assert str(F.hash_message_to_int(params, row["message"])) == row["expected"]

    def vk_of(row):
        def poly(v):
            return PN(modulus=params.modulus, degree=params.degree, root=params.root, inv_root=params.inv_root,
                      root_order=params.root_order, values=list(v))
        return F.OneTimeVerificationKey(left_vk_hat=GeneralMatrix(matrix=[[poly(row["vk_left"])]]),
                                        right_vk_hat=GeneralMatrix(matrix=[[poly(row["vk_right"])]]))
    for row in kat["hash_vk_and_int_to_bytes"]:
        b = F.hash_vk_and_int_to_bytes(params, vk_of(row), int(row["i"]), row["n"])
        assert len(b) == row["n"] and hashlib.sha256(b).hexdigest() == row["sha256_expected_bytes"]
    k = vk_of(kat["hash_ch"][0])
    assert str(k).startswith("OneTimeVerificationKey(left_vk_hat=GeneralMatrix(elem_class=<class "
                             "'algebra.polynomials.PolynomialNTTRepresentation'>, matrix=[[PolynomialNTTRepresentation(")
    assert str(F.Signature(signature_hat=k.left_vk_hat)).startswith("Signature(signature_hat=GeneralMatrix(")
    assert str(F.OneTimeSigningKey(seed=3, left_sk_hat=k.left_vk_hat, right_sk_hat=k.right_vk_hat)).startswith(
        "OneTimeSigningKey(seed=3, left_sk_hat=GeneralMatrix(")
    c = F.SignatureChallenge(c_hat=k.left_vk_hat.matrix[0][0])
    assert str(c) == f"SignatureChallenge(c_hat={k.left_vk_hat.matrix[0][0]})" and c == c
    assert str(F.AggregationCoefficient(alpha_hat=c.c_hat)) == f"AggregationCoefficient(alpha_hat={c.c_hat})"


def test_decoder_vectors_and_bounds():
    """The two hand-computable vectors of the reference's decoder test (all-zero and all-one bytes,
    q=65537, d=1024, weight 100, bias 256) plus the norm/weight bounds on random bytes."""
    import fusion.fusion as F
    q, d, beta, omega, bias = 65537, 1024, 1000, 100, 256
    cb, ib, sb = ceil((log2(beta) + 1 + bias) / 8), ceil((log2(d) + bias) / 8), ceil(omega / 8)
    zeros = bytes(sb + cb * omega + ib * d)
    # all signs negative, all magnitudes 1; every shuffle draw is j = 0: only position 0 and d-1 move
    want = [0] + [-1] * (omega - 1) + [0] * (d - omega - 1) + [-1]
    assert F.decode_bytes_to_polynomial_coefficients(zeros, bias, q, d, beta, omega) == want
    ones = int("1" * omega, 2).to_bytes(sb, "big") + (1).to_bytes(cb, "big") * omega + (1).to_bytes(ib, "big") * d
    want = [2, 0] + [2] * (omega - 2) + [0] * (d - omega - 1) + [2]
    assert F.decode_bytes_to_polynomial_coefficients(ones, bias, q, d, beta, omega) == want
    with pytest.raises(ValueError):
        F.decode_bytes_to_polynomial_coefficients(zeros[:-1 - ib * (d - omega)], bias, q, d, beta, omega)
    rng = random.Random(1)
    for secpar in (128, 256):
        params = F.fusion_setup(secpar, 2)
        for om, be in ((1, 1), (5, 3), (11, 11), (params.omega_ch, 1)):
            cb = ceil((log2(be) + 1 + secpar) / 8)
            ib = ceil((log2(params.degree) + secpar) / 8)
            n = ceil(params.omega_ch / 8) + cb * om + params.degree * ib
            y = F.decode_bytes_to_polynomial_coefficients(rng.randbytes(n), secpar, params.modulus, params.degree, be, om)
            assert len(y) == params.degree and max(abs(v) for v in y) <= be and sum(1 for v in y if v) == om
    with pytest.raises(ValueError):
        F.parse_challenge(F.fusion_setup(128, 2), b"\x00" * 100)


def test_seeded_matrices_are_replicated_without_changing_observable_state():
    """sample_coefficient_matrix / sample_ntt_matrix with a seed: every entry equal (the reference re-seeds per entry,
    fusion.py:156-199), entries independent objects, and the process-global `random` left exactly where l separate
    calls would leave it; without a seed every entry is a fresh draw"""
    import fusion.fusion as F
    from algebra.polynomials import sample_polynomial_coefficient_representation as samp
    p = _params(65537, 64)
    m = F.sample_coefficient_matrix(seed=77, **p, num_rows=5, num_cols=1, norm_bound=9, weight_bound=20)
    state_after_matrix = random.getstate()
    one = samp(**p, norm_bound=9, weight_bound=20, seed=77)
    for _ in range(4):
        samp(**p, norm_bound=9, weight_bound=20, seed=77)               # what the reference does for the other rows
    assert random.getstate() == state_after_matrix
    rows = [r[0] for r in m.matrix]
    assert all(r.coefficients == one.coefficients for r in rows)
    assert len({id(r) for r in rows}) == 5 and len({id(r.coefficients) for r in rows}) == 5
    rows[1].coefficients[0] += 1                                        # mutation stays local
    assert rows[0].coefficients == one.coefficients and rows[2].coefficients == one.coefficients
    n = F.sample_ntt_matrix(seed=78, **p, num_rows=3, num_cols=2)
    flat = [z for r in n.matrix for z in r]
    assert all(z.values == flat[0].values for z in flat) and len({id(z.values) for z in flat}) == 6
    random.seed(1)
    u = F.sample_coefficient_matrix(seed=None, **p, num_rows=3, num_cols=1, norm_bound=9, weight_bound=20)
    assert len({tuple(r[0].coefficients) for r in u.matrix}) == 3      # unseeded: independent draws


def test_bench_starts_its_own_ranks_and_fails_loudly_without_gpus():
    """`bench.py --gpus 2` without a launcher spawns its ranks itself (before touching a GPU) and exits non-zero -- without
    hanging and without a result line -- when a rank cannot run (here: no GPU in the CPU container)."""
    import subprocess
    import sys as _sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([_sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5"], capture_output=True,
                       text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert '{"metric"' not in r.stdout
    assert "rank" in r.stderr and "no result" in r.stderr


def test_numa_helper_is_harmless_without_a_gpu(monkeypatch):
    """fusion_hip.numa reads sysfs only: no KFD topology here -> no GPU nodes, nothing pinned, affinity untouched"""
    import os
    from fusion_hip import numa
    assert numa._cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert numa._cpulist("") == set()
    before = os.sched_getaffinity(0)
    if not os.path.isdir("/sys/class/kfd/kfd/topology/nodes"):
        assert numa.gpu_numa_nodes() == []
        assert numa.pin_to_gpu_node(0) is None
    monkeypatch.setenv("FZ_NO_PIN", "1")
    assert numa.pin_to_gpu_node(0) is None
    assert os.sched_getaffinity(0) == before


def test_tools_and_examples_compile():
    """the measurement tools and examples are not imported by any CPU test (they need a GPU to run): at least their syntax is
    checked here, and the shell scripts name files that exist"""
    import glob
    import py_compile
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = glob.glob(os.path.join(root, "tools", "*.py")) + glob.glob(os.path.join(root, "tools", "probes", "*.py")) + \
        glob.glob(os.path.join(root, "examples", "*.py")) + [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")]
    assert len(files) > 20
    for f in files:
        py_compile.compile(f, doraise=True)
    for sh in glob.glob(os.path.join(root, "tools", "*.sh")) + glob.glob(os.path.join(root, "tools", "probes", "*.sh")):
        text = open(sh).read()
        for m in re.finditer(r"(?:python3?|bash)\s+(?:\$R/)?(tools/[\w/]+\.(?:py|sh))", text):
            assert os.path.exists(os.path.join(root, m.group(1))), (sh, m.group(1))
        for m in re.finditer(r"tools/microbench/build/(\w+)", text):
            name = re.sub(r"_(clang|gcc)$", "", m.group(1))                    # host microbenchmarks: <name>_<compiler> built from <name>.cpp
            assert any(os.path.exists(os.path.join(root, "tools", "microbench", name + ext)) for ext in (".hip", ".cpp")), (sh, m.group(1))


def test_seed_array_takes_what_the_c_samplers_take():
    """BatchScheme.keygen_batch validates its seeds in one numpy conversion: non-negative ints below 2^64 - 1 become a uint64
    array (lists, arrays, bools, integral floats as int() reads them); anything negative or wider returns None -- the caller
    then uses the Python sampler, which follows random.seed()'s abs() and wide-key rules (fusion.py:339-362)"""
    import numpy as np
    from fusion_hip.scheme import _seed_array
    assert _seed_array([1, 2, 3]).tolist() == [1, 2, 3] and _seed_array([1, 2, 3]).dtype == np.uint64
    assert _seed_array([2**64 - 2, 5]).tolist() == [2**64 - 2, 5]
    assert _seed_array(np.array([3, 4], dtype=np.uint64)).tolist() == [3, 4]
    assert _seed_array(np.array([3, 4], dtype=np.int32)).tolist() == [3, 4]
    assert _seed_array([]).size == 0 and _seed_array([True, 7]).tolist() == [1, 7] and _seed_array([2.0]).tolist() == [2]
    assert _seed_array(range(4)).tolist() == [0, 1, 2, 3]
    for bad in ([2**64 - 1, 5], [-1, 5], [-1, 2**63], [2**64, 1], np.array([-3, 4]), [2**70]):
        assert _seed_array(bad) is None


def test_table_context_caches_are_bounded(monkeypatch):
    """contexts built from caller-supplied tables are kept in an LRU of 8 and CLOSED on eviction (a caller iterating over tables
    used to leak a context -- streams, events, device tables, a pool -- per table); the generic path's cache is bounded the same way"""
    import fusion_hip.context as C
    import fusion_hip.wide as W
    closed = []

    class FakeCtx:
        def __init__(self, *a, **k):
            self.key = (a, tuple(sorted(k.items())))

        def close(self):
            closed.append(self)
    monkeypatch.setattr(C, "Context", FakeCtx)
    monkeypatch.setattr(C, "_TABLE_CTX_CACHE", type(C._TABLE_CTX_CACHE)())
    first = C.get_table_context(17, 4, (1, 2, 3, 4), (4, 3, 2, 1))
    assert C.get_table_context(17, 4, (1, 2, 3, 4), (4, 3, 2, 1)) is first
    for k in range(C._TABLE_CTX_MAX + 3):
        C.get_table_context(17, 4, (1, 2, 3, 5 + k), (4, 3, 2, 1))
        C.get_table_context(17, 4, (1, 2, 3, 4), (4, 3, 2, 1))                # kept alive by use
    assert len(C._TABLE_CTX_CACHE) == C._TABLE_CTX_MAX and len(closed) == 4 and first not in closed
    assert C.get_table_context(17, 4, (1, 2, 3, 4), (4, 3, 2, 1)) is first

    class FakeWide:
        def __init__(self, *a):
            pass
    monkeypatch.setattr(W, "WideContext", FakeWide)
    monkeypatch.setattr(W, "_WIDE_CACHE", type(W._WIDE_CACHE)())
    for k in range(W._WIDE_MAX + 5):
        W.get_wide_context(2 ** 40 + 15, 4, (1, 2, 3, k), None)
    assert len(W._WIDE_CACHE) == W._WIDE_MAX
