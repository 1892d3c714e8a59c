"""Every launch shape and arithmetic form the library instantiates, forced through its knob (read once, at context creation)
and checked against the oracle at both parameter sets: the radix-4 transform kernels with 1 / 2 / 4 row groups per wave and
1 / 4 / 8 waves per workgroup, the 16-per-lane persistent kernels, the fused kernels with integer or fp64 accumulation, the
centring or lazy norm test, relaxed or acquire / release arrival, the multi-launch paths behind them, both aggregation kernels.  The defaults pick among these by batch size; a size-dependent choice that is never hit by the other tests'
sizes would otherwise go unchecked.  Reference arithmetic: algebra/ntt.py:216-377, fusion/fusion.py:338-373, :680-728."""
import os

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
I32 = np.iinfo(np.int32)

TRANSFORM_KNOBS = [
    {"FZ_NTT_KERNEL": "4", "FZ_NTT_ROWS": "1"},        # workgroups of 1 / 4 / 8 waves by the row counts below
    {"FZ_NTT_KERNEL": "4", "FZ_NTT_ROWS": "2"},
    {"FZ_NTT_KERNEL": "4", "FZ_NTT_ROWS": "4"},
    {"FZ_NTT_KERNEL": "16"},
    {},                                                # the default choice
]

FUSED_KNOBS = [
    {},
    {"FZ_NO_IMAD": "1"},
    {"FZ_VERIFY_CENT": "1"},
    {"FZ_VERIFY_CENT": "1", "FZ_VERIFY_ORDERED": "1", "FZ_NO_IMAD": "1"},
    {"FZ_UNFUSED": "1"},                               # the multi-launch paths other degrees take
]


def _ctx(P, env):
    import fusion_hip
    for k, v in env.items():
        os.environ[k] = v
    try:
        return fusion_hip.Context(P["q"], P["d"], P["root"], P["inv_root"])
    finally:
        for k in env:
            os.environ.pop(k, None)


def _ident(env):
    return ",".join(f"{k[3:]}={v}" for k, v in env.items()) or "defaults"


@pytest.mark.parametrize("env", TRANSFORM_KNOBS, ids=_ident)
@pytest.mark.parametrize("secpar", [128, 256])
def test_transform_launch_shapes(secpar, env, coracle):
    P = O.PARAMS[secpar]
    q, d = P["q"], P["d"]
    ctx = _ctx(P, env)
    rng = np.random.default_rng(secpar)
    try:
        # one-row-group launches use workgroups of 1 / 4 / 8 waves from 4 x / 8 x 256 wave-tasks on (a task = 1 row at degree
        # 256, 4 rows at degree 64): 1500 and 4099 rows at degree 256, 4099 and 8300 at degree 64 reach the larger two
        for rows in (1, 3, 64, 65, 257, 1500, 4099, 8300):
            x = O.splitmix_centered(rows + secpar, rows * d).reshape(rows, d).copy()
            x[0, :] = I32.min                      # any int32 is a legal input
            if rows > 1:
                x[1] = rng.integers(I32.min, I32.max, size=d, dtype=np.int64).astype(np.int32)
            f = coracle.ntt_forward(x, q, P["root"]).reshape(rows, d)
            assert np.array_equal(ctx.ntt_forward(x), f), ("fwd", rows)
            assert np.array_equal(ctx.ntt_inverse(x), coracle.ntt_inverse(x, q, P["inv_root"]).reshape(rows, d)), ("inv", rows)
            g = O.splitmix_centered(rows + 7, rows * d).reshape(rows, d)
            want = coracle.ntt_inverse(coracle.pw_mul(f, coracle.ntt_forward(g, q, P["root"]).reshape(rows, d), q), q, P["inv_root"])
            assert np.array_equal(ctx.poly_mul(x, g), want.reshape(rows, d)), ("polymul", rows)
    finally:
        ctx.close()


@pytest.mark.parametrize("env", FUSED_KNOBS, ids=_ident)
@pytest.mark.parametrize("secpar", [128, 256])
def test_fused_kernel_forms(secpar, env, coracle):
    """keygen_core, the coefficient-domain product and verification (int32 rows and int64 partial sums; passing, norm-failing
    and target-mismatching aggregates; few and many aggregates per launch: the rows-per-wave gates sit between them)"""
    import fusion_hip
    P = O.PARAMS[secpar]
    q, d, l = P["q"], P["d"], P["rank"]
    ctx = _ctx(P, env)
    rng = np.random.default_rng(secpar + 11)
    A = O.splitmix_centered(5, l * d).reshape(l, d).copy()
    A[0, :] = I32.max
    A[1, :] = I32.min
    try:
        for n in (1, 5, 67):
            coef = (rng.integers(1, 53, size=(n, 2, l, d)) * rng.choice(np.array([-1, 1]), size=(n, 2, l, d))).astype(np.int32)
            coef[0, 0, 0] = I32.max
            rsk, rvk = coracle.keygen_core(A, coef, q, P["root"])
            sk, vk = ctx.keygen_core(A, coef)
            assert np.array_equal(sk, rsk) and np.array_equal(vk, rvk), ("keygen", n)
        rows = 517
        f = rng.integers(I32.min, I32.max, size=(rows, d), dtype=np.int64).astype(np.int32)
        g = O.splitmix_centered(3, rows * d).reshape(rows, d)
        want = coracle.ntt_inverse(coracle.pw_mul(coracle.ntt_forward(f, q, P["root"]).reshape(rows, d),
                                                  coracle.ntt_forward(g, q, P["root"]).reshape(rows, d), q), q, P["inv_root"])
        assert np.array_equal(ctx.poly_mul(f, g), want.reshape(rows, d))
        for groups in (1, 6, 900 if secpar == 256 else 400):
            sig = coracle.ntt_forward(rng.integers(-40, 41, size=(groups * l, d)).astype(np.int32), q, P["root"]).reshape(groups, l, d)
            if groups > 2:
                sig[2] = rng.integers(I32.min, I32.max, size=(l, d), dtype=np.int64).astype(np.int32)
            target = coracle.matvec(A, sig, q).astype(np.int32)
            want = []
            for gi in range(groups):
                mx, wt = coracle.norm_weight(coracle.ntt_inverse(sig[gi], q, P["inv_root"]), q)
                want.append(4 if mx.max() > P["beta_vf"] else (5 if wt.max() > d else 0))
            assert want[0] == 0
            bad = target.copy()
            bad[:, d - 1] -= 1
            expect_bad = [3] * groups                  # the target is compared before the norm (fusion.py:718-727)
            sig64 = sig.astype(np.int64) + q * rng.integers(-500, 500, size=sig.shape)
            bufs = [fusion_hip.DeviceArray.from_numpy(ctx, a) for a in (A, sig, target, bad, sig64, target.astype(np.int64) + 5 * q)]
            dA, dS, dT, dB, d64, dT64 = bufs
            dV = fusion_hip.DeviceArray(ctx, (groups,))
            try:
                assert ctx.verify_with_target_batch_dev(dA.ptr, dS.ptr, dT.ptr, groups, l, P["beta_vf"], d) == want, groups
                assert ctx.verify_with_target_batch_dev(dA.ptr, dS.ptr, dB.ptr, groups, l, P["beta_vf"], d) == expect_bad, groups
                ctx.verify_partials_batch_async_dev(dA.ptr, d64.ptr, l * d, dT64.ptr, d, groups, l, P["beta_vf"], d, dV.ptr)
                assert dV.numpy().tolist() == want, groups
            finally:
                for b in bufs + [dV]:
                    b.free()
    finally:
        ctx.close()


@pytest.mark.parametrize("slices", ["-1", "0", "1", "2", "4", "8", "16"])
@pytest.mark.parametrize("secpar", [128, 256])
def test_matvec_forms(secpar, slices, coracle):
    """the 1 x l by l x 1 product (matrices.py:115-131) for batches that take the many-products path: integer accumulation
    with 1 / 2 / 4 slices of the k range per column, the fp64 kernel (-1) and the default choice; any int32 in A and S,
    scheme ranks and ranks that leave ragged slices, a last workgroup that is not full"""
    import fusion_hip
    P = O.PARAMS[secpar]
    q, d = P["q"], P["d"]
    ctx = _ctx(P, {"FZ_MATVEC_SLICES": slices})
    rng = np.random.default_rng(secpar + 5)
    try:
        batch = 65536 * 4 // d + 3                 # past the few-products kernel's range (fz_launch_matvec), not a multiple of 64 columns
        for l in (P["rank"], 7, 33, 100):
            A = rng.integers(I32.min, I32.max, size=(l, d), dtype=np.int64).astype(np.int32)
            S = rng.integers(I32.min, I32.max, size=(batch, l, d), dtype=np.int64).astype(np.int32)
            A[0, :] = I32.min
            S[0, 0, :] = I32.min
            A[l - 1, ::2] = I32.max
            S[1, l - 1, :] = I32.max
            assert np.array_equal(ctx.matvec(A, S), coracle.matvec(A, S, q)), (l, slices)
    finally:
        ctx.close()


@pytest.mark.parametrize("secpar", [128, 256])
def test_many_aggregates_verification_forms(secpar, coracle):
    """a workgroup per aggregate (512 aggregates or more per launch) gives the oracle's verdicts -- int32 rows and int64 partial
    sums, ranks that fill the last wave-task and ranks that do not, passing / norm-failing / target-mismatching aggregates, any
    int32 in A (fusion.py:690-727)"""
    waves = "0"
    import fusion_hip
    P = O.PARAMS[secpar]
    q, d = P["q"], P["d"]
    ctx = _ctx(P, {})
    rng = np.random.default_rng(secpar + int(waves) + 40)
    groups = 530
    try:
        for l in (P["rank"], 1, 1024 // d, 1024 // d + 1, 37):
            A = rng.integers(I32.min, I32.max, size=(l, d), dtype=np.int64).astype(np.int32)
            A[0, :] = I32.min
            sig = coracle.ntt_forward(rng.integers(-40, 41, size=(groups * l, d)).astype(np.int32), q, P["root"]).reshape(groups, l, d)
            sig[2] = rng.integers(I32.min, I32.max, size=(l, d), dtype=np.int64).astype(np.int32)      # raw rows: the norm fails
            sig[3, l - 1, :] = I32.min
            sig[groups - 1, 0, :] = I32.max
            target = coracle.matvec(A, sig, q).astype(np.int32)
            want, want_w = [], []
            for gi in range(groups):
                mx, wt = coracle.norm_weight(coracle.ntt_inverse(sig[gi], q, P["inv_root"]), q)
                want.append(4 if mx.max() > P["beta_vf"] else (5 if wt.max() > d else 0))
                want_w.append(4 if mx.max() > P["beta_vf"] else (5 if wt.max() > d - 3 else 0))     # a weight bound that can fail
            assert want[0] == 0 and want[2] == 4 and 5 in want_w
            bad = target.copy()
            bad[::3, (7 * l) % d] += 1
            expect_bad = [3 if gi % 3 == 0 else want[gi] for gi in range(groups)]
            sig64 = sig.astype(np.int64) + q * rng.integers(-500, 500, size=sig.shape)
            bufs = [fusion_hip.DeviceArray.from_numpy(ctx, a) for a in (A, sig, target, bad, sig64, target.astype(np.int64) - 7 * q)]
            dA, dS, dT, dB, d64, dT64 = bufs
            dV = fusion_hip.DeviceArray(ctx, (groups,))
            try:
                assert ctx.verify_with_target_batch_dev(dA.ptr, dS.ptr, dT.ptr, groups, l, P["beta_vf"], d) == want, l
                assert ctx.verify_with_target_batch_dev(dA.ptr, dS.ptr, dB.ptr, groups, l, P["beta_vf"], d) == expect_bad, l
                ctx.verify_partials_batch_async_dev(dA.ptr, d64.ptr, l * d, dT64.ptr, d, groups, l, P["beta_vf"], d, dV.ptr)
                assert dV.numpy().tolist() == want, l
                assert ctx.verify_with_target_batch_dev(dA.ptr, dS.ptr, dT.ptr, groups, l, P["beta_vf"], d - 3) == want_w, l
            finally:
                for b in bufs + [dV]:
                    b.free()
    finally:
        ctx.close()


@pytest.mark.parametrize("unfused", ["0", "1"])
@pytest.mark.parametrize("secpar", [128, 256])
def test_broadcast_keygen_forms(secpar, unfused, coracle):
    """fz_keygen_core_bcast -- one secret polynomial per key half, what the reference's seeded keygen produces (fusion.py:156-173,
    :338-362) -- through the one-transform kernel (keygen_bcast_fused, the default) and through the three launches every other
    degree takes (FZ_UNFUSED=1: rows expanded, transformed, multiplied by A): both equal the oracle's keygen on the replicated
    rows, for any int32 in A and in the secret, scheme ranks and ranks that leave row slots empty"""
    P = O.PARAMS[secpar]
    q, d = P["q"], P["d"]
    ctx = _ctx(P, {"FZ_UNFUSED": unfused})
    rng = np.random.default_rng(secpar + 77)
    try:
        for l, n in ((P["rank"], 9), (1, 3), (5, 2), (4 * (256 // d) * 4 + 1, 2), (300, 2)):
            A = rng.integers(I32.min, I32.max, size=(l, d), dtype=np.int64).astype(np.int32)
            A[0, :] = I32.min
            A[l - 1, ::2] = I32.max
            polys = (rng.integers(1, 53, size=(n, 2, d)) * rng.choice(np.array([-1, 1]), size=(n, 2, d))).astype(np.int32)
            polys[0, 0] = rng.integers(I32.min, I32.max, size=d, dtype=np.int64).astype(np.int32)
            polys[0, 1, :] = I32.min
            rsk, rvk = coracle.keygen_core(A, np.repeat(polys[:, :, None, :], l, axis=2), q, P["root"])
            sk, vk = ctx.keygen_core_bcast(A, polys)
            assert np.array_equal(sk, rsk) and np.array_equal(vk, rvk), (l, n)
    finally:
        ctx.close()


def test_device_sampler_forms():
    """the reference's seeded secret-key sampler on the device (polynomials.py:436-467 driven by fusion.py:339-362) in BOTH its
    forms, which the launcher picks by size: seed + draw kernels up to 4096 keys, the one lane-per-polynomial kernel beyond
    (the 4100-key case) -- each equals the C clone on the host (itself pinned to CPython's random): one- and two-word seeds,
    bounds with few and many rejections, degrees that end inside a generation and that need several"""
    import fusion_hip
    from fusion_hip import hostpipe
    P = O.PARAMS[256]
    q = P["q"]
    ctx = _ctx(P, {})
    rng = np.random.default_rng(2024)
    try:
        for nn, deg, bound in ((1, 256, 52), (5, 64, 52), (70, 256, 1), (33, 16, 2**31 - 1), (3, 100, 7), (200, 256, 33),
                               (9, 256, 2**20 + 7), (4100, 64, 52), (2, 4, 1000)):
            seeds = [int(v) for v in rng.integers(0, 2**63, size=nn, dtype=np.uint64)]
            seeds[0] = 0
            if nn > 2:
                seeds[1] = 2**32 - 1                 # seed + 1 crosses into a two-word key
                seeds[2] = 2**64 - 2
            for k in range(3, nn, 2):
                seeds[k] %= 2**32                    # one-word keys
            do = fusion_hip.DeviceBuffer(ctx, nn * 2 * deg * 4)
            try:
                ctx.sample_secret_polys_dev(seeds, q, deg, bound, deg, do.ptr)
                got = do.to_numpy(np.int32, (nn, 2, deg))
            finally:
                do.free()
            assert np.array_equal(got, hostpipe.sample_secret_polys(seeds, q, deg, bound, deg)), (nn, deg, bound)
    finally:
        ctx.close()


@pytest.mark.parametrize("direct", ["-1", "0", "2", "4"])
@pytest.mark.parametrize("secpar", [128, 256])
def test_aggregation_forms(secpar, direct, coracle):
    """sum_i sigma_i (.) alpha_i (fusion.py:670-676) and the verification target (:706-714) through both aggregation kernels --
    the sliced one with shared accumulator words (FZ_AGG_DIRECT=-1) and the slice-free one with 2 / 4 rows per tile -- and the
    default choice: centred int32 and int64 partial sums, with and without the target in the same launch, uniform and ragged
    groups (an empty one included), signer counts on both sides of every fold / depth boundary, any int32 operands"""
    import fusion_hip
    DA = fusion_hip.DeviceArray
    P = O.PARAMS[secpar]
    q, d = P["q"], P["d"]
    ctx = _ctx(P, {"FZ_AGG_DIRECT": direct})
    rng = np.random.default_rng(secpar + 77)

    def raw(*shape):
        return rng.integers(I32.min, I32.max, size=shape, dtype=np.int64).astype(np.int32)

    def cent(v):
        return np.asarray((v + q // 2) % q - q // 2).astype(np.int64)
    try:
        for l in (P["rank"], 5):
            for N, groups in ((1, 1), (3, 2), (33, 1), (130, 3), (300, 1)):
                sig, al = raw(groups, N, l, d), raw(groups, N, d)
                sig[0, 0, 0, :] = I32.min
                al[0, 0, :] = I32.min
                want = np.stack([coracle.aggregate_core(sig[g], al[g], q) for g in range(groups)]).reshape(groups, l, d)
                vkL, vkR, ch = raw(groups, N, d), raw(groups, N, d), raw(groups, N, d)
                Lo, Ro, Co, Ao = (a_.astype(object) for a_ in (vkL, vkR, ch, al))          # exact Python integers
                tgt = cent(((Lo * Co + Ro) * Ao).sum(axis=1))
                bufs = [DA.from_numpy(ctx, a) for a in (sig, al, vkL, vkR, ch)]
                dS, dAl, dL, dR, dC = bufs
                dO, dP, dT = DA(ctx, (groups, l, d)), DA(ctx, (groups, l * d), np.int64), DA(ctx, (groups, d), np.int64)
                try:
                    if groups == 1:
                        ctx.aggregate_core_dev(dS.ptr, dAl.ptr, dO.ptr, N, l)
                        assert np.array_equal(dO.numpy()[0], want[0]), (l, N, "core")
                    ctx.aggregate_partial_batch_dev(dS.ptr, dAl.ptr, dP.ptr, l * d, groups, N, l)
                    assert np.array_equal(cent(dP.numpy()).reshape(groups, l, d), want), (l, N, groups, "partial")
                    ctx.aggregate_target_partial_batch_dev(dS.ptr, dAl.ptr, dL.ptr, dR.ptr, dC.ptr, dP.ptr, l * d, dT.ptr, d, groups, N, l)
                    assert np.array_equal(cent(dP.numpy()).reshape(groups, l, d), want), (l, N, groups, "partial+target")
                    assert np.array_equal(cent(dT.numpy()), tgt), (l, N, groups, "target")
                    # ragged: the same rows cut into aggregates of different sizes, one of them empty
                    rows = groups * N
                    cuts = sorted({0, rows, min(rows, 1), rows // 2, rows // 2, (2 * rows) // 3})
                    offsets = [0] + [c_ for c_ in cuts if 0 < c_ < rows] + [rows // 2 if rows > 1 else rows, rows]
                    offsets = sorted(offsets)                                   # a repeated offset = an empty aggregate
                    G = len(offsets) - 1
                    fs, fa = sig.reshape(rows, l, d), al.reshape(rows, d)
                    dRg = DA(ctx, (G, l, d))
                    try:
                        ctx.aggregate_core_ragged_dev(dS.ptr, dAl.ptr, offsets, l, dRg.ptr)
                        got = dRg.numpy()
                        for g in range(G):
                            lo_, hi_ = offsets[g], offsets[g + 1]
                            exp = coracle.aggregate_core(fs[lo_:hi_], fa[lo_:hi_], q).reshape(l, d) if hi_ > lo_ else np.zeros((l, d), np.int32)
                            assert np.array_equal(got[g], exp), (l, N, groups, "ragged", g, offsets)
                    finally:
                        dRg.free()
                finally:
                    for b in bufs + [dO, dP, dT]:
                        b.free()
    finally:
        ctx.close()
