"""The C host pipeline (csrc/fz_host.cpp: serialiser, SHA3/SHAKE, decoder -- SURVEY 8f row N1) against
CPython's hashlib, the drop-in Python host functions (themselves pinned by the reference's KAT rows in
tests/test_host_logic.py) and the KAT rows directly.  No GPU needed."""
import hashlib
import json
import os
import random

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def hp():
    import __graft_entry__ as g
    g.build()
    from fusion_hip import hostpipe
    return hostpipe


@pytest.fixture(scope="module")
def kat():
    with open(os.path.join(G, "kat.json")) as fh:
        return json.load(fh)


def test_keccak_against_hashlib(hp):
    rng = random.Random(7)
    for n in [0, 1, 3, 135, 136, 137, 271, 272, 273, 1000, 6344, 100000]:
        data = rng.randbytes(n)
        assert hp.sha3_256(data) == hashlib.sha3_256(data).digest()
        for out in (0, 1, 32, 135, 136, 137, 1551, 10436, 3968 * 3 + 1):
            assert hp.shake256(data, out) == hashlib.shake_256(data).digest(out)


@pytest.mark.parametrize("variant", ["scalar", "bmi2", "x64", "x64v"])
def test_every_keccak_variant_against_hashlib(variant):
    """the four Keccak-f[1600] implementations (FZ_KECCAK is read when the library is loaded: one process per variant)"""
    import subprocess
    import sys
    code = (
        "import sys, hashlib, random\n"
        f"sys.path.insert(0, {os.path.join(ROOT, 'fusion-cryptography_amd')!r})\n"
        "from fusion_hip import hostpipe\n"
        "rng = random.Random(11)\n"
        "for n in [0, 1, 7, 8, 135, 136, 137, 271, 272, 273, 1000, 6344, 13600, 100001]:\n"
        "    data = rng.randbytes(n)\n"
        "    assert hostpipe.sha3_256(data) == hashlib.sha3_256(data).digest(), n\n"
        "    for out in (0, 1, 32, 135, 136, 137, 1551, 10436):\n"
        "        assert hostpipe.shake256(data, out) == hashlib.shake_256(data).digest(out), (n, out)\n"
        "print(hostpipe.keccak_variant())\n")
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, FZ_KECCAK=variant), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = r.stdout.split()[-1]
    flags = open("/proc/cpuinfo").read()
    bmi = " bmi2" in flags and " bmi1" in flags
    supported = {"scalar": True, "bmi2": bmi, "x64": bmi, "x64v": bmi and " avx512f" in flags and " avx512vl" in flags}[variant]
    assert got == variant if supported else got in ("scalar", "bmi2", "x64", "x64v")


def test_generated_keccak_assembly_is_up_to_date():
    """csrc/fz_keccak_x64.inc is what tools/gen_keccak_x64.py writes (the generator is the source, the .inc is committed so that
    the build needs no Python)"""
    import subprocess
    import sys
    assert subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_keccak_x64.py"), "--check"]).returncode == 0


@pytest.mark.parametrize("secpar", [128, 256])
def test_formats_and_decoder_match_dropin(secpar, hp):
    import fusion.fusion as F
    from algebra.matrices import GeneralMatrix
    from algebra.polynomials import PolynomialNTTRepresentation as PN
    params = F.fusion_setup(secpar, 3)
    P = hp.scheme_params(params)
    rng = np.random.default_rng(secpar)
    d, q = params.degree, params.modulus
    N = 5
    vkL = rng.integers(-(q // 2), q // 2 + 1, size=(N, d)).astype(np.int32)
    vkR = rng.integers(-(q // 2), q // 2 + 1, size=(N, d)).astype(np.int32)
    vkL[0, :4] = [0, -1, q // 2, -(q // 2)]
    msgs = ["", "a", "message number 0003 !!", "ünï¢ode ✓", "x" * 1000]

    def poly(v):
        return PN(modulus=q, degree=d, root=params.root, inv_root=params.inv_root, root_order=params.root_order,
                  values=[int(t) for t in v])
    keys = [F.OneTimeVerificationKey(left_vk_hat=GeneralMatrix(matrix=[[poly(vkL[i])]]),
                                     right_vk_hat=GeneralMatrix(matrix=[[poly(vkR[i])]])) for i in range(N)]
    for i in range(N):
        assert hp.format_vk(P, vkL[i], vkR[i]) == str(keys[i])
    pre = hp.hash_messages(P, msgs)
    for i in range(N):
        assert int.from_bytes(bytes(pre[i]), "little") == F.hash_message_to_int(params, msgs[i])
    coefs, pre2 = hp.challenge_coefficients(P, vkL, vkR, msgs, threads=3)
    assert np.array_equal(pre, pre2)
    n = F._challenge_bytes_needed(params)
    for i in range(N):
        xof = F.hash_vk_and_int_to_bytes(params, keys[i], F.hash_message_to_int(params, msgs[i]), n)
        want = F.decode_bytes_to_polynomial_coefficients(xof, secpar, q, d, params.beta_ch, params.omega_ch)
        assert coefs[i].tolist() == want
        assert hp.decode_coefficients(xof, secpar, q, d, params.beta_ch, params.omega_ch).tolist() == want
    # sorted(key=str(vk))
    order = hp.sort_by_vk_string(P, vkL, vkR)
    assert order.tolist() == sorted(range(N), key=lambda i: str(keys[i]))
    # aggregation coefficients: one XOF over str(list(zip(keys, ints, challs)))
    c_hat = rng.integers(-(q // 2), q // 2 + 1, size=(N, d)).astype(np.int32)
    challs = [F.SignatureChallenge(c_hat=poly(c_hat[i])) for i in range(N)]
    ints = [F.hash_message_to_int(params, m) for m in msgs]
    b = F.hash_vks_and_ints_and_challs_to_bytes(params, keys, ints, challs)
    per = F._agg_coef_bytes(params)
    want = [F.decode_bytes_to_polynomial_coefficients(b[i * per:(i + 1) * per], secpar, q, d, params.beta_ag,
                                                      params.omega_ag) for i in range(N)]
    got = hp.aggregation_coefficients(P, vkL, vkR, pre, c_hat, threads=2)
    assert got.tolist() == want


def test_decoder_general_bounds(hp):
    import fusion.fusion as F
    rng = random.Random(11)
    for (q, d, beta, omega, bias) in [(65537, 1024, 1000, 100, 256), (2147465729, 64, 3, 27, 128),
                                      (2147465729, 256, 52, 256, 256), (17, 8, 5, 3, 16), (257, 16, 1, 16, 8)]:
        from math import ceil, log2
        bound = max(1, min(q // 2, beta))
        cb, ib, sb = ceil((log2(bound) + 1 + bias) / 8), ceil((log2(d) + bias) / 8), ceil(omega / 8)
        for extra in (0, ib * d):
            b = rng.randbytes(sb + (cb + ib) * omega + extra)
            want = F.decode_bytes_to_polynomial_coefficients(b, bias, q, d, beta, omega)
            assert hp.decode_coefficients(b, bias, q, d, beta, omega).tolist() == want[:d]
        with pytest.raises(ValueError):
            hp.decode_coefficients(b"\x00" * (sb + (cb + ib) * omega - 1), bias, q, d, beta, omega)


def test_reference_kats_through_c_pipeline(hp, kat, coracle):
    """intermediate_hash_message_to_int / hash_vk_and_int_to_bytes / hash_ch KAT rows of the reference,
    replayed through the C pipeline (forward NTT from the oracle: this test runs without a GPU)."""
    import fusion.fusion as F
    from oracle import oracle as O
    params = F.fusion_setup(128, 1)
    P = hp.scheme_params(params)
    rows = kat["hash_message_to_int"]
    pre = hp.hash_messages(P, [r["message"] for r in rows])
    assert [str(int.from_bytes(bytes(p), "little")) for p in pre] == [r["expected"] for r in rows]
    rows = kat["hash_vk_and_int_to_bytes"]
    for r in rows:
        x = (params.sign_hash_dst + b"," + hp.format_vk(P, r["vk_left"], r["vk_right"]).encode() + b"," + r["i"].encode())
        assert hashlib.sha256(hp.shake256(x, r["n"])).hexdigest() == r["sha256_expected_bytes"]
    rows = kat["hash_ch"]
    coefs, _ = hp.challenge_coefficients(P, [r["vk_left"] for r in rows], [r["vk_right"] for r in rows],
                                         [r["message"] for r in rows])
    c_hat = coracle.ntt_forward(coefs, O.PRIME, O.PARAMS[128]["root"])
    assert c_hat.tolist() == [r["c_hat"] for r in rows]


def test_sampler_clone_matches_cpython_random(hp, kat):
    """C clone of CPython's MT19937 seeding + randrange against the drop-in samplers
    (which call CPython's own `random`) and the reference's fusion_setup KAT rows."""
    from algebra.polynomials import (sample_polynomial_coefficient_representation as samp_c,
                                     sample_polynomial_ntt_representation as samp_n)
    q = 2147465729
    ring = dict(modulus=q, degree=256, root=3337519, inv_root=pow(3337519, q - 2, q), root_order=512)
    for seed in (0, 1, 42, 2**31, 2**32 - 1, 2**32, 2**32 + 5, 2**40 + 12345, 2173728648):
        assert hp.sample_ntt_values(seed, q, 256).tolist() == samp_n(**ring, seed=seed).values
        assert hp.sample_coefficients(seed, q, 256, 52, 256).tolist() == \
            samp_c(**ring, norm_bound=52, weight_bound=256, seed=seed).coefficients
        # sparse case: exercises the Fisher-Yates shuffle
        assert hp.sample_coefficients(seed, q, 256, 3, 60).tolist() == \
            samp_c(**ring, norm_bound=3, weight_bound=60, seed=seed).coefficients
    r64 = dict(modulus=q, degree=64, root=23584283, inv_root=pow(23584283, q - 2, q), root_order=128)
    assert hp.sample_coefficients(7, q, 64, 52, 64).tolist() == samp_c(**r64, norm_bound=52, weight_bound=64, seed=7).coefficients
    small = dict(modulus=65537, degree=1024, root=None, inv_root=None, root_order=2048)
    from algebra.ntt import find_primitive_root
    small["root"] = find_primitive_root(65537, 2048)
    small["inv_root"] = pow(small["root"], 65535, 65537)
    assert hp.sample_coefficients(123456789, 65537, 1024, 1000, 100).tolist() == \
        samp_c(**small, norm_bound=1000, weight_bound=100, seed=123456789).coefficients
    for row in kat["setup"]:                      # fusion_setup_KAT_{128,256}.csv
        d = 64 if row["secpar"] == 128 else 256
        assert hp.sample_ntt_values(row["seed"], q, d).tolist() == row["first_poly"]
    polys = hp.sample_secret_polys([5, 2**32 - 1], q, 256, 52, 256, threads=2)
    assert polys[1, 1].tolist() == samp_c(**ring, norm_bound=52, weight_bound=256, seed=2**32).coefficients
    assert polys[0, 0].tolist() == samp_c(**ring, norm_bound=52, weight_bound=256, seed=5).coefficients


@pytest.mark.parametrize("secpar", [128, 256])
def test_sort_by_vk_string_with_ties_and_shared_prefixes(secpar, hp):
    """sorted(keys, key=str) (fusion.py:661-663, :693) through the short-prefix comparison: keys that are identical (the
    reference's demo generates every key from one seed), keys that differ only deep inside the left polynomial, only in the right
    one, only in the sign or the number of digits of the first value -- the order, ties included (stable), is Python's"""
    import fusion.fusion as F
    params = F.fusion_setup(secpar, 7)
    P = hp.scheme_params(params)
    q, d = params.modulus, params.degree
    rng = np.random.default_rng(secpar)
    base = rng.integers(-(q // 2), q // 2 + 1, size=(2, d)).astype(np.int32)
    rows = []
    for _ in range(3):
        rows.append(base.copy())                                   # identical keys
    for pos in (0, 1, 7, 8, 9, d - 1):                             # one value of the left polynomial differs, early and late
        k = base.copy()
        k[0, pos] += 1
        rows.append(k)
    for pos in (0, d - 1):                                         # only the right polynomial differs
        k = base.copy()
        k[1, pos] -= 1
        rows.append(k)
    for v0 in (5, 57, -5, -57, 0, 570, 9, 10, -1):                 # "5" < "57", "-" < digits, "10" < "9" as text
        k = base.copy()
        k[0, 0] = v0
        rows.append(k)
        rows.append(k.copy())
    vk = np.stack(rows)
    perm = rng.permutation(len(rows))
    vk = vk[perm]
    vkL, vkR = np.ascontiguousarray(vk[:, 0]), np.ascontiguousarray(vk[:, 1])

    def text(i):
        return hp.format_vk(P, vkL[i], vkR[i])
    for threads in (1, 3):
        order = hp.sort_by_vk_string(P, vkL, vkR, threads)
        assert order.tolist() == sorted(range(len(rows)), key=text)


def test_sampler_state_export_equals_cpython(hp):
    """fz_sample_coefficients_state: the polynomial AND the generator's state afterwards are what CPython's `random` holds after
    sample_polynomial_coefficient_representation(seed=...) (polynomials.py:436-467) -- the drop-in keygen relies on it to
    leave the process-global generator where the reference leaves it (random.setstate) without 512 randrange() calls"""
    import fusion.fusion as F
    from algebra.polynomials import sample_polynomial_coefficient_representation as sample
    for secpar in (128, 256):
        T = F.PREFIX_PARAMETERS[secpar]
        for seed in (0, 1, 42, 2**32 - 1, 2**32, 2**63 + 5, 2**64 - 2):
            ref = sample(modulus=T["modulus"], degree=T["degree"], root=T["root"], inv_root=T["inv_root"], root_order=T["root_order"],
                         norm_bound=T["beta_sk"], weight_bound=T["omega_sk"], seed=seed)
            want = random.getstate()
            row, state = hp.sample_coefficients_with_state(seed, T["modulus"], T["degree"], T["beta_sk"], T["omega_sk"])
            assert row.tolist() == ref.coefficients and want == (3, state, None), (secpar, seed)
    # a weight bound below the degree: the shuffle's draws are part of the state too
    ref = sample(modulus=65537, degree=64, root=1, inv_root=1, root_order=1, norm_bound=9, weight_bound=10, seed=7)
    want = random.getstate()
    row, state = hp.sample_coefficients_with_state(7, 65537, 64, 9, 10)
    assert row.tolist() == ref.coefficients and want == (3, state, None)
