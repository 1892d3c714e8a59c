"""GPU parity (through the C ABI) against golden vectors produced by the reference and against
the CPU oracle: pointwise ops, matrix-vector product, norm/weight, bulk digests."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def alg():
    return np.load(os.path.join(G, "algebra.npz"))


def tags(alg):
    return ["p128", "p256"] + [str(t) for t in alg["small_tags"]]


def sha_i32(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<i4").tobytes()).hexdigest()


def test_golden_transforms_and_pointwise(alg):
    import fusion_hip
    for t in tags(alg):
        q, d, root, inv = (int(v) for v in alg[f"{t}_params"])
        ctx = fusion_hip.Context(q, d, root, inv)
        f, i = ctx.twiddles()
        assert np.array_equal(f, alg[f"{t}_tw"]) and np.array_equal(i, alg[f"{t}_itw"]), t
        x = alg[f"{t}_x"]
        assert np.array_equal(ctx.ntt_forward(x), alg[f"{t}_fwd"]), t
        assert np.array_equal(ctx.ntt_inverse(x), alg[f"{t}_inv"]), t
        a, b = x, np.concatenate([x[1:], x[1:2]])
        ok = ~alg[f"{t}_pw_b_is_zero"]
        assert np.array_equal(ctx.pw_mul(a, b), alg[f"{t}_pw_mul"]), t
        assert np.array_equal(ctx.pw_add(a, b)[ok], alg[f"{t}_pw_add"][ok]), t
        assert np.array_equal(ctx.pw_sub(a, b)[ok], alg[f"{t}_pw_sub"][ok]), t
        assert np.array_equal(ctx.pw_neg(a).astype(np.int64), alg[f"{t}_pw_neg"]), t
        # schoolbook product of the reference == NTT -> pointwise -> INTT on the device
        hats_f, hats_g = ctx.ntt_forward(alg[f"{t}_sb_f"]), ctx.ntt_forward(alg[f"{t}_sb_g"])
        assert np.array_equal(ctx.ntt_inverse(ctx.pw_mul(hats_f, hats_g)), alg[f"{t}_sb_fg"]), t
        assert np.array_equal(ctx.matvec(alg[f"{t}_mv_A"], alg[f"{t}_mv_S"]), alg[f"{t}_mv_out"]), t


def test_bulk_digests_config2():
    """BASELINE config 2: B = 4096 degree-256 forward + inverse, checked by SHA-256 against the
    digests of the reference's outputs (and the secpar-128 twin)."""
    import fusion_hip
    with open(os.path.join(G, "bulk.json")) as fh:
        bulk = json.load(fh)
    for secpar, case in bulk["cases"].items():
        q, d, root, inv = case["q"], case["d"], case["root"], case["inv_root"]
        ctx = fusion_hip.Context(q, d, root, inv)
        x = O.splitmix_centered(20261003, bulk["B"] * d, q).reshape(bulk["B"], d)
        fwd = ctx.ntt_forward(x)
        assert sha_i32(fwd) == case["sha256_fwd"]
        assert sha_i32(ctx.ntt_inverse(x)) == case["sha256_inv"]
        assert sha_i32(ctx.ntt_inverse(ctx.pw_mul(fwd, fwd))) == case["sha256_fwd_square_inv"]
        assert np.array_equal(ctx.ntt_inverse(fwd), x)


@pytest.mark.parametrize("count", [0, 1, 2, 3, 5, 255, 256, 1021, 4096 * 3 + 1])
def test_pointwise_ragged_and_unaligned(count, coracle):
    import fusion_hip
    q = O.PRIME
    ctx = fusion_hip.Context(q, 1, 0, 0)        # ring-only context
    rng = np.random.default_rng(count)
    a = rng.integers(-2**31, 2**31, size=count + 3, dtype=np.int64).astype(np.int32)
    b = rng.integers(-2**31, 2**31, size=count + 3, dtype=np.int64).astype(np.int32)
    # host face
    assert np.array_equal(ctx.pw_mul(a[:count], b[:count]), coracle.pw_mul(a[:count], b[:count], q))
    if count == 0:
        return
    # device face with deliberately unaligned pointers (offset 4 bytes)
    da, db = fusion_hip.DeviceBuffer.from_numpy(ctx, a), fusion_hip.DeviceBuffer.from_numpy(ctx, b)
    dout = fusion_hip.DeviceBuffer(ctx, a.nbytes)
    for off in (0, 4):
        sl = slice(off // 4, off // 4 + count)
        for op, ref in ((fusion_hip.OP_MUL, coracle.pw_mul), (fusion_hip.OP_ADD, coracle.pw_add),
                        (fusion_hip.OP_SUB, coracle.pw_sub)):
            ctx.pw_dev(op, da.ptr + off, db.ptr + off, dout.ptr + off, count)
            got = dout.to_numpy(np.int32, (count + 3,))[sl]
            assert np.array_equal(got, ref(a[sl], b[sl], q)), (op, off)
        ctx.pw_neg_dev(da.ptr + off, dout.ptr + off, count)
        assert np.array_equal(dout.to_numpy(np.int32, (count + 3,))[sl], coracle.pw_neg(a[sl], q))
        acc = rng.integers(-2**31, 2**31, size=count + 3, dtype=np.int64).astype(np.int32)
        ctx.h2d(dout.ptr, acc)
        ctx.pw_mulacc_dev(dout.ptr + off, da.ptr + off, db.ptr + off, count)
        assert np.array_equal(dout.to_numpy(np.int32, (count + 3,))[sl], coracle.pw_mulacc(acc[sl], a[sl], b[sl], q))


@pytest.mark.parametrize("secpar", [128, 256])
def test_norm_weight_and_matvec_random(secpar, coracle):
    import fusion_hip
    P = O.PARAMS[secpar]
    q, d, l = P["q"], P["d"], P["rank"]
    ctx = fusion_hip.Context(q, d, P["root"], P["inv_root"])
    rng = np.random.default_rng(secpar)
    x = rng.integers(-2**31, 2**31, size=(l + 5, d), dtype=np.int64).astype(np.int32)
    x[0] = 0
    x[1, ::2] = q
    x[2, ::3] = -q
    x[3] = -2**31
    mx, wt = ctx.norm_weight(x)
    rmx, rwt = coracle.norm_weight(x, q)
    assert np.array_equal(mx, rmx) and np.array_equal(wt, rwt)
    A = O.splitmix_centered(1, l * d).reshape(l, d)
    S = rng.integers(-2**31, 2**31, size=(7, l, d), dtype=np.int64).astype(np.int32)
    assert np.array_equal(ctx.matvec(A, S), coracle.matvec(A, S, q))


def test_linearity_at_full_size():
    """size-independent property on the BASELINE batch: NTT(a + b) == NTT(a) + NTT(b) (mod q, centred)."""
    import fusion_hip
    P = O.PARAMS[256]
    ctx = fusion_hip.Context(P["q"], P["d"], P["root"], P["inv_root"])
    B = 4096
    a = O.splitmix_centered(1, B * 256).reshape(B, 256)
    b = O.splitmix_centered(2, B * 256).reshape(B, 256)
    lhs = ctx.ntt_forward(ctx.pw_add(a, b))
    rhs = ctx.pw_add(ctx.ntt_forward(a), ctx.ntt_forward(b))
    assert np.array_equal(lhs, rhs)
    # convolution theorem against X: multiplying by X rotates with a sign flip (negacyclic)
    xpoly = np.zeros(256, np.int32)
    xpoly[1] = 1
    prod = ctx.ntt_inverse(ctx.pw_mul(ctx.ntt_forward(a), np.broadcast_to(ctx.ntt_forward(xpoly), a.shape)))
    expect = np.concatenate([-a[:, -1:], a[:, :-1]], axis=1)
    assert np.array_equal(prod, expect)


def test_device_synthetic_generator_matches_host_generator():
    """fz_fill_synthetic == oracle.splitmix_centered (the generator tests and bench use), any offset, ragged counts"""
    import fusion_hip
    from oracle import oracle as O
    P = O.PARAMS[256]
    ctx = fusion_hip.Context(P["q"], P["d"], P["root"], P["inv_root"])
    small = fusion_hip.Context(257, 4, 0, 0)                         # another modulus (ring-only context)
    for c, q in ((ctx, P["q"]), (small, 257)):
        for seed, count in ((20261003, 4096 * 256 + 7), (5, 1), (2**63 + 11, 1000)):
            buf = fusion_hip.DeviceBuffer(c, count * 4)
            c.fill_synthetic_dev(buf.ptr, count, seed)
            want = O.splitmix_centered(seed, count, q)
            assert np.array_equal(buf.to_numpy(np.int32, (count,)), want)
            buf.free()
    ctx.fill_synthetic_dev(0, 0, 1)                                   # empty: nothing to do
