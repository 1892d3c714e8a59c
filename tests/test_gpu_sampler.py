"""GPU parity of the device sampler (csrc/fz_sample.hip: CPython's MT19937 per lane) against
 * CPython's own `random` driven exactly as the reference's sampler drives it (algebra/polynomials.py:436-467),
 * the C clone on the host (fz_sample_secret_polys, itself pinned by the fusion_setup KAT rows),
 * the reference's golden keys (tests/golden/scheme_*.npz) through keygen_batch.
Reference: fusion/fusion.py:339-362 (seed for the left matrix, seed + 1 for the right one)."""
import os
import random

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _reference_poly(seed, degree, bound):
    """sample_polynomial_coefficient_representation with weight bound = degree, on CPython's generator"""
    rng = random.Random()
    rng.seed(seed)
    return [(1 + rng.randrange(bound)) * (1 - 2 * rng.randrange(2)) for _ in range(degree)]


@pytest.mark.parametrize("degree,bound", [(256, 52), (64, 52), (256, 1), (16, 2**31 - 1), (100, 7)])
def test_device_sampler_is_cpythons_mt19937(degree, bound):
    import fusion_hip
    from fusion_hip import hostpipe
    P = O.PARAMS[256]
    ctx = fusion_hip.get_context(P["q"], P["d"], P["root"], P["inv_root"])
    q = 2**32 - 5 if bound > 2**30 else P["q"]          # the sampler's modulus only caps the bound at q // 2
    seeds = [0, 1, 2, 42, 2**32 - 2, 2**32 - 1, 2**32, 2**32 + 1, 2**63 + 12345, 2**64 - 2, 987654321987654321]
    seeds += [10_000 + 2 * i for i in range(150)]       # more than two waves; ragged last wave
    n = len(seeds)
    dout = fusion_hip.DeviceBuffer(ctx, n * 2 * degree * 4)
    try:
        ctx.sample_secret_polys_dev(seeds, q, degree, bound, degree, dout.ptr)
        got = dout.to_numpy(np.int32, (n, 2, degree))
        want = hostpipe.sample_secret_polys(seeds, q, degree, bound, degree)
        assert np.array_equal(got, want)
        for i in (0, 1, 4, 5, 6, 8, 9, 10, 11, n - 1):
            for half in (0, 1):
                assert got[i, half].tolist() == _reference_poly(seeds[i] + half, degree, min(bound, q // 2)), (seeds[i], half)
    finally:
        dout.free()


def test_unsupported_shapes_fall_back(coracle):
    import fusion_hip
    P = O.PARAMS[256]
    ctx = fusion_hip.get_context(P["q"], P["d"], P["root"], P["inv_root"])
    dout = fusion_hip.DeviceBuffer(ctx, 4 * 2 * 256 * 4)
    try:
        with pytest.raises(fusion_hip.FusionHipError) as e:
            ctx.sample_secret_polys_dev([1, 2], P["q"], 256, 52, 255, dout.ptr)      # weight < degree: a shuffle follows
        from fusion_hip._lib import FZ_E_UNSUPPORTED
        assert e.value.code == FZ_E_UNSUPPORTED
    finally:
        dout.free()


@pytest.mark.parametrize("secpar", [128, 256])
def test_keygen_batch_with_the_device_sampler_equals_the_host_one(secpar):
    import fusion.fusion as F
    from fusion_hip.scheme import BatchScheme
    params = F.fusion_setup(secpar, 42)
    bs = BatchScheme(params)
    seeds = [7, 8, 2**40 + 3] + list(range(100, 140))
    assert bs.device_sampler
    sk_d, vk_d = bs.keygen_batch(seeds)
    assert bs.device_sampler                      # no fallback happened
    bs.device_sampler = False
    sk_h, vk_h = bs.keygen_batch(seeds)
    assert np.array_equal(sk_d, sk_h) and np.array_equal(vk_d, vk_h)


def test_sampler_argument_checks():
    import fusion_hip
    from fusion_hip._lib import FZ_E_BADARG
    P = O.PARAMS[256]
    ctx = fusion_hip.get_context(P["q"], P["d"], P["root"], P["inv_root"])
    dout = fusion_hip.DeviceBuffer(ctx, 2 * 2 * 64 * 4)
    try:
        with pytest.raises(fusion_hip.FusionHipError) as e:
            ctx.sample_secret_polys_dev([1, 2], P["q"], 64, 0, 64, dout.ptr)         # randrange(0): empty range
        assert e.value.code == FZ_E_BADARG
        ctx.sample_secret_polys_dev([], P["q"], 64, 52, 64, dout.ptr)                 # empty batch: nothing to do
    finally:
        dout.free()


def test_large_batch_more_waves_than_cus():
    """20 000 keys = 625 one-wave workgroups (one per CU at a time: the 156 KiB of generator states fill a CU's LDS), i.e. more
    than one round over the chip; every polynomial against the C clone"""
    import fusion_hip
    from fusion_hip import hostpipe
    P = O.PARAMS[256]
    ctx = fusion_hip.get_context(P["q"], P["d"], P["root"], P["inv_root"])
    n = 20000
    rng = np.random.default_rng(5)
    seeds = [int(v) for v in rng.integers(0, 2**40, size=n, dtype=np.uint64)]
    dout = fusion_hip.DeviceBuffer(ctx, n * 2 * 256 * 4)
    try:
        ctx.sample_secret_polys_dev(seeds, P["q"], 256, 52, 256, dout.ptr)
        assert np.array_equal(dout.to_numpy(np.int32, (n, 2, 256)), hostpipe.sample_secret_polys(seeds, P["q"], 256, 52, 256))
    finally:
        dout.free()


def test_keygen_batch_takes_every_seed_the_reference_takes():
    """negative seeds (random.seed uses abs(); the right half is seeded with seed + 1 = abs(seed) - 1) and seeds of 2^64 - 1 and
    beyond (three-word MT19937 keys) go through CPython's `random` itself; results equal the object API's keygen"""
    import fusion.fusion as F
    from fusion_hip.scheme import BatchScheme, signature_from_object
    params = F.fusion_setup(128, 3)
    bs = BatchScheme(params, threads=2)
    seeds = [-1, -12345, 2 ** 64 - 1, 2 ** 64, 2 ** 70 + 5, 7]
    sk, vk = bs.keygen_batch(seeds)
    for i, s in enumerate(seeds):
        sk_o, vk_o = F.keygen(params, s)
        assert np.array_equal(vk[i, 0], np.array(vk_o.left_vk_hat.matrix[0][0].values)), s
        assert np.array_equal(vk[i, 1], np.array(vk_o.right_vk_hat.matrix[0][0].values)), s
        assert np.array_equal(sk[i, 0], signature_from_object(params, F.Signature(signature_hat=sk_o.left_sk_hat))), s
    from fusion_hip import hostpipe
    from fusion_hip._lib import FusionHipError
    with pytest.raises(FusionHipError):
        hostpipe.sample_secret_polys(np.array([2 ** 64 - 1], dtype=np.uint64), params.modulus, params.degree, 52, params.degree)
