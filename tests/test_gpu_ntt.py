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
            np.where(np.arange(d) % 3 == 0, 2**31 - 1, -2**31)]
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
