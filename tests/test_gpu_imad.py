"""The A (.) y accumulation of the fused keygen and verification kernels -- 64-bit integer multiply-adds on A split into 16-bit
halves (default) and the general fp64 multiply (FZ_NO_IMAD=1, read at context creation) -- against the oracle: centred rows, RAW
int32 extremes in A, in the secrets and in the signature rows (the kernels accept any int32), ranks far beyond the scheme's.
Reference arithmetic: fusion/fusion.py:369-370 (A * sk_hat), :715-717 (A * aggregate), :690-727 (verdict order)."""
import os
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
Q = O.PRIME
I32 = np.iinfo(np.int32)


def _cent(x, q=Q):
    y = np.asarray(x, dtype=np.int64) % q
    return np.where(y > q // 2, y - q, y)


def _contexts(P):
    """(name, Context) for both forms of the accumulation"""
    import fusion_hip
    out = [("imad", fusion_hip.Context(P["q"], P["d"], P["root"], P["inv_root"]))]
    os.environ["FZ_NO_IMAD"] = "1"
    try:
        out.append(("fp64", fusion_hip.Context(P["q"], P["d"], P["root"], P["inv_root"])))
    finally:
        del os.environ["FZ_NO_IMAD"]
    return out


def _A(rng, l, d, extremes):
    A = O.splitmix_centered(int(rng.integers(1 << 30)), l * d).reshape(l, d).copy()
    if extremes:                                   # any int32 is a legal input: the reference would reduce it on use
        A[0, :] = I32.max
        A[1, :] = I32.min
        A[2, ::2] = -65536
        A[2, 1::2] = 65535
        A[3, :] = rng.integers(I32.min, I32.max, size=d, dtype=np.int64).astype(np.int32)
        A[4, :] = -1
    return A


@pytest.mark.parametrize("secpar,l,extremes", [(256, 83, False), (256, 83, True), (128, 195, True), (256, 300, True), (128, 700, True)])
def test_keygen_accumulation_both_forms(secpar, l, extremes, coracle):
    import fusion_hip
    P = O.PARAMS[secpar]
    q, d = P["q"], P["d"]
    rng = np.random.default_rng(secpar + l)
    n = 3
    A = _A(rng, l, d, extremes)
    coef = (rng.integers(1, 53, size=(n, 2, l, d)) * rng.choice(np.array([-1, 1]), size=(n, 2, l, d))).astype(np.int32)
    if extremes:
        coef[0, 0, 0] = I32.max                    # raw int32 secrets too (the transform takes any int32)
        coef[0, 1, 1] = I32.min
    rsk, rvk = coracle.keygen_core(A, coef, q, P["root"])
    for name, ctx in _contexts(P):
        dA, dC = fusion_hip.DeviceArray.from_numpy(ctx, A), fusion_hip.DeviceArray.from_numpy(ctx, coef)
        dS, dV = fusion_hip.DeviceArray(ctx, coef.shape), fusion_hip.DeviceArray(ctx, (n, 2, d))
        try:
            ctx.keygen_core_dev(dA.ptr, dC.ptr, dS.ptr, dV.ptr, n, l)
            assert np.array_equal(dS.numpy(), rsk), name
            assert np.array_equal(dV.numpy(), rvk), name
        finally:
            for b in (dA, dC, dS, dV):
                b.free()
            ctx.close()


@pytest.mark.parametrize("secpar,l,groups", [(256, 83, 1), (256, 83, 700), (128, 195, 5), (256, 300, 3), (128, 700, 2)])
def test_verification_accumulation_both_forms(secpar, l, groups, coracle):
    """observed = A * sigma must equal the oracle's matvec for centred AND raw int32 rows: with target = the oracle's product
    the verdict can be OK / NORM / WEIGHT but never TARGET_MISMATCH; with one coefficient of the target off by one it must be
    TARGET_MISMATCH; both forms of the accumulation agree on every aggregate, for int32 rows and for int64 partial sums."""
    import fusion_hip
    P = O.PARAMS[secpar]
    q, d = P["q"], P["d"]
    rng = np.random.default_rng(secpar * 3 + l + groups)
    A = _A(rng, l, d, True)
    ctx0 = fusion_hip.Context(q, d, P["root"], P["inv_root"])
    small = ctx0.ntt_forward((rng.integers(-50, 51, size=(groups, l, d))).astype(np.int32))      # passes the norm bound
    ctx0.close()
    sig = small.copy()
    raw = min(groups - 1, 2)
    if groups > 1:
        sig[raw] = rng.integers(I32.min, I32.max, size=(l, d), dtype=np.int64).astype(np.int32)   # raw int32 rows: norm fails
        sig[raw, 0, :] = I32.min
        sig[raw, 1, :] = I32.max
    target = coracle.matvec(A, sig, q)                                  # [groups][d] centred
    want = []
    for g in range(groups):
        coef = coracle.ntt_inverse(sig[g], q, P["inv_root"])
        mx, wt = coracle.norm_weight(coef, q)
        want.append(4 if mx.max() > P["beta_vf"] else (5 if wt.max() > d else 0))
    assert want[0] == 0 and (groups == 1 or want[raw] == 4)
    bad = target.astype(np.int32).copy()
    bad[:, 7] += 1
    # int64 rows (the sums an all-reduce leaves): the same aggregates shifted by multiples of q (centred on load; the norm
    # of an inverse transform depends on the residues only, so the verdicts are the same)
    sig64 = sig.astype(np.int64) + q * rng.integers(-1000, 1000, size=sig.shape)
    tgt64 = target.astype(np.int64) - 3 * q
    for name, ctx in _contexts(P):
        dA, dS, dT, dB = (fusion_hip.DeviceArray.from_numpy(ctx, x) for x in (A, sig, target.astype(np.int32), bad))
        d64, dT64 = fusion_hip.DeviceArray.from_numpy(ctx, sig64), fusion_hip.DeviceArray.from_numpy(ctx, tgt64)
        dV = fusion_hip.DeviceArray(ctx, (groups,))
        try:
            assert ctx.verify_with_target_batch_dev(dA.ptr, dS.ptr, dT.ptr, groups, l, P["beta_vf"], d) == want, name
            assert ctx.verify_with_target_batch_dev(dA.ptr, dS.ptr, dB.ptr, groups, l, P["beta_vf"], d) == [3] * groups, name
            ctx.verify_partials_batch_async_dev(dA.ptr, d64.ptr, l * d, dT64.ptr, d, groups, l, P["beta_vf"], d, dV.ptr)
            assert dV.numpy().tolist() == want, name
        finally:
            for b in (dA, dS, dT, dB, d64, dT64, dV):
                b.free()
            ctx.close()


def test_runtime_info_reports_what_the_library_is_bound_to():
    import fusion_hip
    P = O.PARAMS[128]
    ctx = fusion_hip.Context(P["q"], P["d"], P["root"], P["inv_root"])
    info = ctx.runtime_info()
    assert info["arch"].startswith("gfx950") and info["build_hip_version"] // 10_000_000 == info["runtime_hip_version"] // 10_000_000
