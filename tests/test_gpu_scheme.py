"""GPU parity of the fused scheme cores and of the drop-in fusion.fusion surface against golden
data produced by running the reference (tests/golden/scheme_*.npz, scheme.json)."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def sha_str(s):
    return hashlib.sha256(s.encode("utf-8")).hexdigest()


@pytest.fixture(scope="module")
def meta():
    with open(os.path.join(G, "scheme.json")) as fh:
        return json.load(fh)


@pytest.mark.parametrize("secpar", [128, 256])
def test_fused_cores_match_reference_arrays(secpar, meta):
    import fusion_hip
    S = np.load(os.path.join(G, f"scheme_{secpar}.npz"))
    m = meta[str(secpar)]
    P = O.PARAMS[secpar]
    ctx = fusion_hip.Context(P["q"], P["d"], P["root"], P["inv_root"])
    sk, vk = ctx.keygen_core(S["A"], S["coef"])
    assert np.array_equal(sk, S["sk_hat"]) and np.array_equal(vk, S["vk"])
    assert np.array_equal(ctx.sign_core(S["sk_hat"], S["c_hat"]), S["sig"])
    for n in (1, 2, 4):
        order = m["agg"][str(n)]["order"]
        alpha = S[f"alpha_hat_{n}"]
        agg = ctx.aggregate_core(S["sig"][order], alpha)
        assert np.array_equal(agg, S[f"agg_{n}"])
        args = (S["vk"][order, 0], S["vk"][order, 1], S["c_hat"][order], alpha)
        assert ctx.verify_core(S["A"], agg, *args, m["beta_vf"], m["omega_vf"]) == 0
        bad = agg.copy()
        bad[0, 0] += 1
        assert ctx.verify_core(S["A"], bad, *args, m["beta_vf"], m["omega_vf"]) == 3
        assert ctx.verify_core(S["A"], agg, *args, 1, m["omega_vf"]) == 4
        assert ctx.verify_core(S["A"], agg, *args, m["beta_vf"], 1) == 5


@pytest.mark.parametrize("secpar", [128, 256])
def test_cores_on_distinct_random_keys(secpar, coracle):
    """The reference's keys repeat one polynomial rank times; the throughput configs use DISTINCT
    rows -- check those against the oracle, with a ragged signature count."""
    import fusion_hip
    P = O.PARAMS[secpar]
    q, d, l = P["q"], P["d"], P["rank"]
    ctx = fusion_hip.Context(q, d, P["root"], P["inv_root"])
    rng = np.random.default_rng(secpar + 1)
    N = 37
    A = O.splitmix_centered(3, l * d).reshape(l, d)
    coef = rng.integers(-52, 53, size=(N, 2, l, d)).astype(np.int32)
    sk, vk = ctx.keygen_core(A, coef)
    rsk, rvk = coracle.keygen_core(A, coef, q, P["root"])
    assert np.array_equal(sk, rsk) and np.array_equal(vk, rvk)
    c = np.zeros((N, d), np.int32)
    for i in range(N):
        c[i, rng.choice(d, P["omega_ch"], replace=False)] = rng.choice([-1, 1], P["omega_ch"])
    c_hat = ctx.ntt_forward(c)
    sig = ctx.sign_core(sk, c_hat)
    assert np.array_equal(sig, coracle.sign_core(sk, c_hat, q))
    alpha = ctx.ntt_forward(np.roll(c, 1, axis=1))
    agg = ctx.aggregate_core(sig, alpha)
    assert np.array_equal(agg, coracle.aggregate_core(sig, alpha, q))
    beta, omega = P["beta_vf"], d
    assert ctx.verify_core(A, agg, vk[:, 0], vk[:, 1], c_hat, alpha, beta, omega) == \
        coracle.verify_core(A, agg, vk[:, 0], vk[:, 1], c_hat, alpha, q, P["inv_root"], beta, omega) == 0
    # split partial sums (the multi-GPU exchange step) reduce to the same aggregate
    halves = []
    for sl in (slice(0, 20), slice(20, N)):
        ds = fusion_hip.DeviceBuffer.from_numpy(ctx, sig[sl])
        da = fusion_hip.DeviceBuffer.from_numpy(ctx, alpha[sl])
        dp = fusion_hip.DeviceBuffer(ctx, l * d * 8)
        ctx.aggregate_partial_dev(ds.ptr, da.ptr, dp.ptr, sig[sl].shape[0], l)
        halves.append(dp.to_numpy(np.int64, (l, d)))
    tot = halves[0] + halves[1]
    dt = fusion_hip.DeviceBuffer.from_numpy(ctx, tot)
    do = fusion_hip.DeviceBuffer(ctx, l * d * 4)
    ctx.reduce_i64_dev(dt.ptr, do.ptr, l * d)
    assert np.array_equal(do.to_numpy(np.int32, (l, d)), agg)


@pytest.mark.parametrize("secpar", [128, 256])
def test_dropin_end_to_end_matches_reference_strings(secpar, meta):
    """setup -> keygen -> sign -> aggregate -> verify through the drop-in API; every object's str()
    must hash to what the reference printed (these strings are hashed by the scheme itself)."""
    import fusion.fusion as F
    m = meta[str(secpar)]
    params = F.fusion_setup(secpar, m["setup_seed"])
    assert sha_str(str(params)) == m["sha256_str_params"]
    keys = [F.keygen(params, s) for s in m["key_seeds"]]
    assert [sha_str(str(k[1])) for k in keys] == m["sha256_str_vk"]
    assert [sha_str(str(k[0])) for k in keys] == m["sha256_str_sk"]
    msgs = m["messages"]
    assert [str(F.hash_message_to_int(params, x)) for x in msgs] == m["prehash"]
    vks = [k[1] for k in keys]
    assert [sha_str(str(F.hash_ch(params, v, x))) for v, x in zip(vks, msgs)] == m["sha256_str_chall"]
    sigs = [F.sign(params, k, x) for k, x in zip(keys, msgs)]
    assert [sha_str(str(s)) for s in sigs] == m["sha256_str_sig"]
    for n in (1, 2, 4):
        agg = F.aggregate(params, vks[:n], msgs[:n], sigs[:n])
        assert sha_str(str(agg)) == m["agg"][str(n)]["sha256_str_agg"]
        assert list(F.verify(params, vks[:n], msgs[:n], agg)) == m["agg"][str(n)]["verdict"] == [True, ""]
        # tamper test of the reference (tests/test_fusion.py:860-873): in-place edit of one value
        agg.signature_hat.matrix[0][0].values[0] += 1
        assert list(F.verify(params, vks[:n], msgs[:n], agg)) == m["agg"][str(n)]["tampered_verdict"]
    assert F.verify(params, vks, msgs[:1], sigs[0]) == (False, "Number of keys and messages must be equal.")
    assert F.verify(params, vks * (params.capacity // 4 + 1), msgs, sigs[0]) == (False, "Too many keys.")


def test_config1_demo_flow(meta):
    """BASELINE config 1 (misc/demo.py plumbing at secpar 128): two keys from the same seed."""
    import fusion.fusion as F
    m = meta["demo128"]
    a = F.fusion_setup(128, m["setup_seed"])
    keys = [F.keygen(a, m["key_seed"]) for _ in range(2)]
    sigs = [F.sign(a, k, x) for k, x in zip(keys, m["messages"])]
    assert [sha_str(str(s)) for s in sigs] == m["sha256_str_sig"]
    vks = [k[1] for k in keys]
    agg = F.aggregate(a, vks, m["messages"], sigs)
    assert sha_str(str(agg)) == m["sha256_str_agg"]
    assert list(F.verify(a, vks, m["messages"], agg)) == m["verdict"] == [True, ""]


def test_batched_aggregate_target_verify(coracle):
    """group-batched entry points (one launch for G aggregates) == the single-aggregate ones == oracle"""
    import fusion_hip
    P = O.PARAMS[256]
    q, d, l = P["q"], P["d"], 7
    ctx = fusion_hip.Context(q, d, P["root"], P["inv_root"])
    rng = np.random.default_rng(5)
    G, N = 3, 13
    A = O.splitmix_centered(3, l * d).reshape(l, d)
    coef = rng.integers(-52, 53, size=(G * N, 2, l, d)).astype(np.int32)
    sk, vk = ctx.keygen_core(A, coef)
    c = np.zeros((G * N, d), np.int32)
    for i in range(G * N):
        c[i, rng.choice(d, 60, replace=False)] = rng.choice([-1, 1], 60)
    c_hat = ctx.ntt_forward(c)
    al_hat = ctx.ntt_forward(np.roll(c, 3, axis=1))
    sig = ctx.sign_core(sk, c_hat)
    DB = fusion_hip.DeviceBuffer
    d_sig, d_al, d_c = DB.from_numpy(ctx, sig), DB.from_numpy(ctx, al_hat), DB.from_numpy(ctx, c_hat)
    d_vkL, d_vkR = DB.from_numpy(ctx, np.ascontiguousarray(vk[:, 0])), DB.from_numpy(ctx, np.ascontiguousarray(vk[:, 1]))
    d_A = DB.from_numpy(ctx, A)
    n_agg, n_tgt = G * l * d, G * d
    d_part = DB(ctx, (n_agg + n_tgt) * 8)
    ctx.aggregate_partial_batch_dev(d_sig.ptr, d_al.ptr, d_part.ptr, l * d, G, N, l)
    ctx.target_partial_batch_dev(d_vkL.ptr, d_vkR.ptr, d_c.ptr, d_al.ptr, d_part.ptr + n_agg * 8, d, G, N)
    d_red = DB(ctx, (n_agg + n_tgt) * 4)
    ctx.reduce_i64_dev(d_part.ptr, d_red.ptr, n_agg + n_tgt)
    red = d_red.to_numpy(np.int32, (n_agg + n_tgt,))
    agg = red[:n_agg].reshape(G, l, d)
    for g in range(G):
        sl = slice(g * N, (g + 1) * N)
        assert np.array_equal(agg[g], coracle.aggregate_core(sig[sl], al_hat[sl], q))
        assert np.array_equal(agg[g], ctx.aggregate_core(sig[sl], al_hat[sl]))
    v = ctx.verify_with_target_batch_dev(d_A.ptr, d_red.ptr, d_red.ptr + n_agg * 4, G, l, P["beta_vf"], d)
    assert v == [0, 0, 0]
    # corrupt the middle aggregate only
    bad = red.copy()
    bad[l * d + 5] += 1
    ctx.h2d(d_red.ptr, bad)
    assert ctx.verify_with_target_batch_dev(d_A.ptr, d_red.ptr, d_red.ptr + n_agg * 4, G, l, P["beta_vf"], d) == [0, 3, 0]
    assert ctx.verify_with_target_batch_dev(d_A.ptr, d_red.ptr, d_red.ptr + n_agg * 4, G, l, 1, d) == [4, 3, 4]

    # one-pass form: aggregate and target partials from the same two launches == the separate entry points
    part_sep = d_part.to_numpy(np.int64, (n_agg + n_tgt,))
    d_part2 = DB(ctx, (n_agg + n_tgt) * 8)
    ctx.aggregate_target_partial_batch_dev(d_sig.ptr, d_al.ptr, d_vkL.ptr, d_vkR.ptr, d_c.ptr, d_part2.ptr, l * d,
                                           d_part2.ptr + n_agg * 8, d, G, N, l)
    part_one = d_part2.to_numpy(np.int64, (n_agg + n_tgt,))
    half = q // 2
    assert np.array_equal((part_one + half) % q, (part_sep + half) % q)       # equal as residues (split points differ)
    assert np.array_equal(((part_one[:n_agg] + half) % q - half).astype(np.int32).reshape(G, l, d), agg)
    # verdicts straight from int64 sums (also shifted by multiples of q, as an all-reduce over ranks would leave them)
    d_verd = DB(ctx, G * 4)
    for shift in (0, 5 * q, -3 * q):
        ctx.h2d(d_part2.ptr, part_one + shift)
        ctx.verify_partials_batch_async_dev(d_A.ptr, d_part2.ptr, l * d, d_part2.ptr + n_agg * 8, d, G, l, P["beta_vf"], d,
                                            d_verd.ptr)
        assert d_verd.to_numpy(np.int32, (G,)).tolist() == [0, 0, 0]
    badp = part_one.copy()
    badp[2 * l * d + 9] += 1                      # last aggregate
    badp[n_agg + 3] -= 1                          # first aggregate's target
    ctx.h2d(d_part2.ptr, badp)
    ctx.verify_partials_batch_async_dev(d_A.ptr, d_part2.ptr, l * d, d_part2.ptr + n_agg * 8, d, G, l, P["beta_vf"], d,
                                        d_verd.ptr)
    assert d_verd.to_numpy(np.int32, (G,)).tolist() == [3, 0, 3]
    ctx.verify_partials_batch_async_dev(d_A.ptr, d_part2.ptr + l * d * 8, l * d, d_part2.ptr + (n_agg + d) * 8, d, G - 1, l,
                                        P["beta_vf"], d, d_verd.ptr)            # a rank's share: groups 1..2
    assert d_verd.to_numpy(np.int32, (G,)).tolist()[:2] == [0, 3]


@pytest.mark.parametrize("secpar,l", [(256, 83), (128, 195), (256, 5)])
@pytest.mark.parametrize("G,N", [(1, 1), (1, 7), (3, 13), (4, 256), (1, 1000), (2, 40)])
def test_sign_and_aggregate_in_one_pass_equals_the_two_calls(secpar, l, G, N, coracle, monkeypatch):
    """fz_sign_aggregate_target_partial_batch (aggregate_onepass<.., SIGN>: sigma written as it is computed, aggregated from
    registers) == fz_sign_core then fz_aggregate_target_partial_batch: the signatures bit for bit (and == the oracle's,
    fusion.py:557), the int64 sums as residues and after centring == the oracle's aggregate (fusion.py:670-676); with and
    without the verification target; record-strided outputs; FZ_UNFUSED=1 runs the two launches behind the same entry."""
    import fusion_hip
    P = O.PARAMS[secpar]
    q, d = P["q"], P["d"]
    rng = np.random.default_rng(1000 * G + N + l)
    A = O.splitmix_centered(3, l * d).reshape(l, d)
    sk = O.splitmix_centered(5, G * N * 2 * l * d).reshape(G * N, 2, l, d)       # any centred rows serve as key halves here
    c_hat = O.splitmix_centered(6, G * N * d).reshape(G * N, d)
    al_hat = O.splitmix_centered(7, G * N * d).reshape(G * N, d)
    vk = O.splitmix_centered(8, G * N * 2 * d).reshape(G * N, 2, d)
    rec = l * d + d + 8                                   # records [l*d sums | d target sums | padding]: any stride >= the row
    DB = fusion_hip.DeviceBuffer
    for env in ({}, {"FZ_UNFUSED": "1"}):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        ctx = fusion_hip.Context(q, d, P["root"], P["inv_root"])
        for k_ in env:
            monkeypatch.delenv(k_)
        d_sk, d_c, d_al = DB.from_numpy(ctx, sk), DB.from_numpy(ctx, c_hat), DB.from_numpy(ctx, al_hat)
        d_vkL, d_vkR = DB.from_numpy(ctx, np.ascontiguousarray(vk[:, 0])), DB.from_numpy(ctx, np.ascontiguousarray(vk[:, 1]))
        d_sig_a, d_sig_b = DB(ctx, G * N * l * d * 4), DB(ctx, G * N * l * d * 4)
        d_pa, d_pb = DB(ctx, G * rec * 8), DB(ctx, G * rec * 8)
        ctx.h2d(d_pa.ptr, np.full(G * rec, -7, np.int64))
        ctx.h2d(d_pb.ptr, np.full(G * rec, -7, np.int64))
        # the two calls
        ctx.sign_core_dev(d_sk.ptr, d_c.ptr, d_sig_a.ptr, G * N, l)
        ctx.aggregate_target_partial_batch_dev(d_sig_a.ptr, d_al.ptr, d_vkL.ptr, d_vkR.ptr, d_c.ptr, d_pa.ptr, rec,
                                               d_pa.ptr + l * d * 8, rec, G, N, l)
        # the one pass (twice: the accumulator words re-arm themselves)
        for _ in range(2):
            ctx.sign_aggregate_target_partial_batch_dev(d_sk.ptr, d_c.ptr, d_al.ptr, d_vkL.ptr, d_vkR.ptr, d_sig_b.ptr, d_pb.ptr, rec,
                                                        d_pb.ptr + l * d * 8, rec, G, N, l)
        sig_a, sig_b = d_sig_a.to_numpy(np.int32, (G * N, l, d)), d_sig_b.to_numpy(np.int32, (G * N, l, d))
        assert np.array_equal(sig_a, sig_b)
        if not env:
            assert np.array_equal(sig_b, coracle.sign_core(sk, c_hat, q))
        pa, pb = d_pa.to_numpy(np.int64, (G, rec)), d_pb.to_numpy(np.int64, (G, rec))
        half = q // 2
        assert np.array_equal((pa[:, :l * d + d] + half) % q, (pb[:, :l * d + d] + half) % q)
        assert np.all(pb[:, l * d + d:] == -7)            # the padding between records is nobody's
        if not env:
            for g in range(G):
                sl = slice(g * N, (g + 1) * N)
                want = coracle.aggregate_core(sig_b[sl], al_hat[sl], q)
                assert np.array_equal(((pb[g, :l * d] + half) % q - half).astype(np.int32).reshape(l, d), want)
        # without the target: the aggregate's sums alone, same values
        ctx.h2d(d_pb.ptr, np.full(G * rec, -7, np.int64))
        ctx.sign_aggregate_target_partial_batch_dev(d_sk.ptr, d_c.ptr, d_al.ptr, 0, 0, d_sig_b.ptr, d_pb.ptr, rec, 0, 0, G, N, l)
        pc = d_pb.to_numpy(np.int64, (G, rec))
        assert np.array_equal((pc[:, :l * d] + half) % q, (pa[:, :l * d] + half) % q) and np.all(pc[:, l * d:] == -7)
        assert np.array_equal(d_sig_b.to_numpy(np.int32, (G * N, l, d)), sig_a)
        for b in (d_sk, d_c, d_al, d_vkL, d_vkR, d_sig_a, d_sig_b, d_pa, d_pb):
            b.free()
        ctx.close()


def test_sign_and_aggregate_in_one_pass_edges(coracle):
    """the same entry where the one-launch form does not apply: rows that are not 16-byte aligned and a degree the one-pass
    kernel does not cover take the two launches behind it (same results); no signers / no aggregates is a no-op"""
    import fusion_hip
    q = O.PARAMS[256]["q"]
    for d in (256, 12):                                # 12: a ring-only context (any degree; scalar kernels)
        ctx = fusion_hip.Context(q, d, O.PARAMS[256]["root"], O.PARAMS[256]["inv_root"]) if d == 256 else fusion_hip.Context(q, d, 0, 0)
        G, N, l = 2, 9, 3
        sk = O.splitmix_centered(15, G * N * 2 * l * d).reshape(G * N, 2, l, d)
        c_hat = O.splitmix_centered(16, G * N * d).reshape(G * N, d)
        al = O.splitmix_centered(17, G * N * d).reshape(G * N, d)
        vk = O.splitmix_centered(18, G * N * 2 * d).reshape(G * N, 2, d)
        DB = fusion_hip.DeviceBuffer
        d_sk, d_c, d_al = DB.from_numpy(ctx, sk), DB.from_numpy(ctx, c_hat), DB.from_numpy(ctx, al)
        d_L, d_R = DB.from_numpy(ctx, np.ascontiguousarray(vk[:, 0])), DB.from_numpy(ctx, np.ascontiguousarray(vk[:, 1]))
        n_sig = G * N * l * d
        d_sig = DB(ctx, n_sig * 4 + 16)
        d_p = DB(ctx, G * (l * d + d) * 8)
        want = coracle.sign_core(sk, c_hat, q)
        half = q // 2
        for off in ((0, 4) if d == 256 else (0,)):          # +4 bytes: the signature rows are no longer 16-byte aligned
            ctx.sign_aggregate_target_partial_batch_dev(d_sk.ptr, d_c.ptr, d_al.ptr, d_L.ptr, d_R.ptr, d_sig.ptr + off, d_p.ptr, l * d,
                                                        d_p.ptr + G * l * d * 8, d, G, N, l)
            got = np.empty(n_sig, np.int32)
            ctx.d2h(got, d_sig.ptr + off)
            assert np.array_equal(got.reshape(G * N, l, d), want), (d, off)
            p = d_p.to_numpy(np.int64, (G * (l * d + d),))
            for g in range(G):
                sl = slice(g * N, (g + 1) * N)
                agg = ((p[g * l * d:(g + 1) * l * d] + half) % q - half).astype(np.int32).reshape(l, d)
                assert np.array_equal(agg, coracle.aggregate_core(want[sl], al[sl], q)), (d, off, g)
        ctx.h2d(d_p.ptr, np.full(G * (l * d + d), 5, np.int64))
        ctx.sign_aggregate_target_partial_batch_dev(d_sk.ptr, d_c.ptr, d_al.ptr, d_L.ptr, d_R.ptr, d_sig.ptr, d_p.ptr, l * d,
                                                    d_p.ptr + G * l * d * 8, d, G, 0, l)                # no signers
        ctx.sign_aggregate_target_partial_batch_dev(0, 0, 0, 0, 0, 0, d_p.ptr, l * d, 0, 0, 0, N, l)   # no aggregates
        assert np.all(d_p.to_numpy(np.int64, (G * (l * d + d),)) == 5)
        with pytest.raises(fusion_hip.FusionHipError):                                                 # keys without the target's buffer
            ctx.sign_aggregate_target_partial_batch_dev(d_sk.ptr, d_c.ptr, d_al.ptr, d_L.ptr, d_R.ptr, d_sig.ptr, d_p.ptr, l * d, 0, 0, G, N, l)
        for b in (d_sk, d_c, d_al, d_L, d_R, d_sig, d_p):
            b.free()
        ctx.close()


@pytest.mark.parametrize("secpar", [128, 256])
def test_fused_verify_equals_unfused_path(secpar, coracle, monkeypatch):
    """verify_fused (sigma read once) vs the four-kernel path (FZ_UNFUSED=1) vs the oracle, on every
    verdict branch, with non-centred raw int32 rows in the aggregate as well."""
    import fusion_hip
    P = O.PARAMS[secpar]
    q, d, l = P["q"], P["d"], P["rank"]
    ctx = fusion_hip.Context(q, d, P["root"], P["inv_root"])
    rng = np.random.default_rng(secpar + 9)
    N = 5
    A = O.splitmix_centered(8, l * d).reshape(l, d)
    coef = rng.integers(-52, 53, size=(N, 2, l, d)).astype(np.int32)
    sk, vk = ctx.keygen_core(A, coef)
    c = np.zeros((N, d), np.int32)
    for i in range(N):
        c[i, rng.choice(d, P["omega_ch"], replace=False)] = rng.choice([-1, 1], P["omega_ch"])
    c_hat, al_hat = ctx.ntt_forward(c), ctx.ntt_forward(np.roll(c, 2, axis=1))
    agg = ctx.aggregate_core(ctx.sign_core(sk, c_hat), al_hat)
    cases = [(agg, P["beta_vf"], d), (agg, 1, d), (agg, P["beta_vf"], 1)]
    bad = agg.copy()
    bad[l - 1, d - 1] += 1
    cases.append((bad, P["beta_vf"], d))
    raw = rng.integers(-2**31, 2**31, size=(l, d), dtype=np.int64).astype(np.int32)   # arbitrary int32 "aggregate"
    cases += [(raw, P["beta_vf"], d), (raw, 2**31, d)]
    for sigma, beta, omega in cases:
        want = coracle.verify_core(A, sigma, vk[:, 0], vk[:, 1], c_hat, al_hat, q, P["inv_root"], beta, omega)
        monkeypatch.delenv("FZ_UNFUSED", raising=False)
        fused = ctx.verify_core(A, sigma, vk[:, 0], vk[:, 1], c_hat, al_hat, beta, omega)
        monkeypatch.setenv("FZ_UNFUSED", "1")
        unfused = ctx.verify_core(A, sigma, vk[:, 0], vk[:, 1], c_hat, al_hat, beta, omega)
        assert fused == unfused == want, (beta, omega)


@pytest.mark.parametrize("secpar", [128, 256])
def test_fused_keygen_equals_unfused_path(secpar, coracle, monkeypatch):
    """keygen_fused (one launch; sk_hat never re-read) vs NTT + matvec launches vs the oracle; ragged key
    counts and arbitrary int32 coefficient rows."""
    import fusion_hip
    P = O.PARAMS[secpar]
    q, d, l = P["q"], P["d"], P["rank"]
    ctx = fusion_hip.Context(q, d, P["root"], P["inv_root"])
    rng = np.random.default_rng(secpar + 21)
    A = O.splitmix_centered(6, l * d).reshape(l, d)
    for n, lo, hi in ((1, -52, 53), (3, -52, 53), (2, -2**31, 2**31)):
        coef = rng.integers(lo, hi, size=(n, 2, l, d), dtype=np.int64).astype(np.int32)
        want_sk, want_vk = coracle.keygen_core(A, coef, q, P["root"])
        monkeypatch.delenv("FZ_UNFUSED", raising=False)
        sk, vk = ctx.keygen_core(A, coef)
        assert np.array_equal(sk, want_sk) and np.array_equal(vk, want_vk)
        monkeypatch.setenv("FZ_UNFUSED", "1")
        sk2, vk2 = ctx.keygen_core(A, coef)
        assert np.array_equal(sk2, want_sk) and np.array_equal(vk2, want_vk)


@pytest.mark.parametrize("d,l", [(256, 83), (64, 195), (32, 5)])
def test_broadcast_keygen_equals_replicated_rows(d, l, coracle, monkeypatch):
    """fz_keygen_core_bcast (one secret polynomial per key half, as the reference's seeded sampler produces) == the
    general entry on l replicated rows == oracle; fused kernel, multi-launch path and a generic degree"""
    import fusion_hip
    q = O.PRIME
    root = next(r for r in (pow(g, (q - 1) // (2 * d), q) for g in range(2, 200)) if pow(r, d, q) == q - 1)
    ctx = fusion_hip.Context(q, d, root, pow(root, q - 2, q))
    DB = fusion_hip.DeviceBuffer
    rng = np.random.default_rng(d + l)
    n = 5
    A = O.splitmix_centered(8, l * d).reshape(l, d)
    one = rng.integers(-52, 53, size=(n, 2, d), dtype=np.int64).astype(np.int32)
    coef = np.ascontiguousarray(np.broadcast_to(one[:, :, None, :], (n, 2, l, d)))
    want_sk, want_vk = coracle.keygen_core(A, coef, q, root)
    for unfused in (False, True):
        if unfused:
            monkeypatch.setenv("FZ_UNFUSED", "1")
        dA, dc = DB.from_numpy(ctx, A), DB.from_numpy(ctx, one)
        dsk, dvk = DB(ctx, coef.nbytes), DB(ctx, n * 2 * d * 4)
        ctx.keygen_core_bcast_dev(dA.ptr, dc.ptr, dsk.ptr, dvk.ptr, n, l)
        assert np.array_equal(dsk.to_numpy(np.int32, (n, 2, l, d)), want_sk)
        assert np.array_equal(dvk.to_numpy(np.int32, (n, 2, d)), want_vk)
        for b in (dA, dc, dsk, dvk):
            b.free()
    ctx.close()


def test_first_verify_on_a_fresh_context_with_its_own_stream(coracle):
    """regression (found by tools/soak.py): the verify accumulators are cleared on the context's stream -- a null-stream
    memset is not ordered with a non-blocking stream, and the first verification of a context raced with it"""
    import fusion_hip
    P = O.PARAMS[128]
    q, d, l, n, G = P["q"], P["d"], 1, 27, 3
    rng = np.random.default_rng(9)
    A = O.splitmix_centered(11, l * d).reshape(l, d)
    coef = rng.integers(-52, 53, size=(G * n, 2, l, d)).astype(np.int32)
    c = np.zeros((G * n, d), np.int32)
    for i in range(G * n):
        c[i, rng.choice(d, 60, replace=False)] = rng.choice([-1, 1], 60)
    DB = fusion_hip.DeviceBuffer
    for _ in range(6):
        ctx = fusion_hip.Context(q, d, P["root"], P["inv_root"])
        s = ctx.stream_create()
        ctx.set_stream(s)
        sk, vk = ctx.keygen_core(A, coef)
        c_hat, al_hat = ctx.ntt_forward(c), ctx.ntt_forward(np.roll(c, 5, axis=1))
        sig = ctx.sign_core(sk, c_hat)
        bufs = [DB.from_numpy(ctx, a) for a in (sig, al_hat, np.ascontiguousarray(vk[:, 0]), np.ascontiguousarray(vk[:, 1]),
                                                c_hat, A)]
        n_agg = G * l * d
        part, verd = DB(ctx, (n_agg + G * d) * 8), DB(ctx, G * 4)
        ctx.aggregate_target_partial_batch_dev(bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, bufs[4].ptr, part.ptr,
                                               l * d, part.ptr + n_agg * 8, d, G, n, l)
        ctx.verify_partials_batch_async_dev(bufs[5].ptr, part.ptr, l * d, part.ptr + n_agg * 8, d, G, l, P["beta_vf"], d,
                                            verd.ptr)
        assert verd.to_numpy(np.int32, (G,)).tolist() == [0] * G
        for b in bufs + [part, verd]:
            b.free()
        ctx.set_stream(0)
        ctx.stream_destroy(s)
        ctx.close()


@pytest.mark.parametrize("secpar,G", [(256, 700), (128, 1100), (256, 200)])
def test_many_aggregates_per_launch_one_workgroup_each(secpar, G, coracle):
    """G aggregates in one launch: above 2 x CUs each aggregate is ONE workgroup's (rows prefetched one ahead, no shared
    accumulators); G = 200 keeps the shared-accumulator path with two workgroups per aggregate.  Every verdict branch, the
    reference's order (target, norm, weight: fusion.py:718-727), int32 and int64 rows, twice (state re-arms itself)."""
    import fusion_hip
    P = O.PARAMS[secpar]
    q, d, l = P["q"], P["d"], P["rank"]
    ctx = fusion_hip.Context(q, d, P["root"], P["inv_root"])
    rng = np.random.default_rng(secpar + G)
    A = O.splitmix_centered(77, l * d).reshape(l, d)
    s = rng.integers(-1000, 1001, size=(G, l, d)).astype(np.int32)
    s[:, :, ::3] = 0                                        # weight < d
    big, heavy, wrong, both = 3, 11, G - 1, 17
    s[big, l - 1, d - 1] = 5000                             # norm 5000 > beta = 1000
    s[heavy, 0, :] = 1                                      # weight d > omega = d - 1
    s[both, 2, 5] = -4000                                   # norm failure AND wrong target: target is reported
    sig = ctx.ntt_forward(s.reshape(G * l, d)).reshape(G, l, d)
    DB = fusion_hip.DeviceBuffer
    dA, dsig = DB.from_numpy(ctx, A), DB.from_numpy(ctx, sig)
    dt, dv = DB(ctx, G * d * 4), DB(ctx, G * 4)
    try:
        ctx.matvec_dev(dA.ptr, dsig.ptr, dt.ptr, G, l)
        tgt = dt.to_numpy(np.int32, (G, d))
        for g in (0, big, G - 1):
            assert np.array_equal(tgt[g], coracle.matvec(A, sig[g:g + 1], q)[0])
        tgt[wrong, d - 1] += 1
        tgt[both, 0] -= 1
        ctx.h2d(dt.ptr, tgt)
        want = np.zeros(G, np.int32)
        want[big], want[heavy], want[wrong], want[both] = 4, 5, 3, 3
        for _ in range(2):
            ctx.verify_with_target_batch_async_dev(dA.ptr, dsig.ptr, dt.ptr, G, l, 1000, d - 1, dv.ptr)
            assert np.array_equal(dv.to_numpy(np.int32, (G,)), want)
        # the same rows as int64 partial sums (centred on load): sigma + k q, target + k q
        sig64 = sig.astype(np.int64) + rng.integers(-2**31, 2**31, size=sig.shape) * q
        tgt64 = tgt.astype(np.int64) - 5 * q
        d64, dt64 = DB.from_numpy(ctx, sig64), DB.from_numpy(ctx, tgt64)
        ctx.h2d(dv.ptr, np.full(G, -1, np.int32))
        ctx.verify_partials_batch_async_dev(dA.ptr, d64.ptr, l * d, dt64.ptr, d, G, l, 1000, d - 1, dv.ptr)
        assert np.array_equal(dv.to_numpy(np.int32, (G,)), want)
        d64.free()
        dt64.free()
    finally:
        for b in (dA, dsig, dt, dv):
            b.free()
        ctx.close()


def test_c_caller_runs_the_whole_scheme_flow(tmp_path):
    """examples/scheme_flow.c: keygen (device sampler) -> sign (device challenge pipeline) -> aggregate (host hash_ag) ->
    verify through the C ABI alone, from strict C99; exit code 0 = accepted, and the tampered aggregate rejected"""
    import subprocess
    from test_cabi_symbols import build_c_example
    r = subprocess.run([build_c_example(tmp_path, "scheme_flow")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "verdict 0 (0 = accepted), tampered aggregate: verdict 3" in r.stdout


def test_a_captured_aggregation_survives_later_scratch_growth(coracle):
    """a graph that holds a multi-slice aggregation and a multi-workgroup verification has the addresses of the
    context's accumulator scratch baked in; later, larger calls on the same context make that scratch grow.  The old
    allocations must stay valid (retired, not freed) and re-armed, so that replaying the graph afterwards still gives the
    oracle's aggregate and verdict (fz_retire, csrc/fz_capi.hip)."""
    import fusion_hip
    P = O.PARAMS[256]
    q, d, l = P["q"], P["d"], P["rank"]
    ctx = fusion_hip.Context(q, d, P["root"], P["inv_root"])
    s = ctx.stream_create()
    ctx.set_stream(s)
    rng = np.random.default_rng(77)
    DB = fusion_hip.DeviceBuffer

    def operands(groups, n):
        sig = rng.integers(-(q // 2), q // 2, size=(groups * n, l, d)).astype(np.int32)
        al = rng.integers(-(q // 2), q // 2, size=(groups * n, d)).astype(np.int32)
        return sig, al
    A = O.splitmix_centered(5, l * d).reshape(l, d)
    dA = DB.from_numpy(ctx, A)
    n = 300                                             # one aggregate of 300 signers: several slices -> accumulator words in use
    sig, al = operands(1, n)
    d_sig, d_al = DB.from_numpy(ctx, sig), DB.from_numpy(ctx, al)
    d_out, d_tgt, d_verd = DB(ctx, l * d * 4), DB(ctx, d * 4), DB(ctx, 4)
    want = coracle.aggregate_core(sig, al, q)
    ctx.h2d(d_tgt.ptr, coracle.matvec(A, want, q).astype(np.int32))
    ctx.aggregate_core_dev(d_sig.ptr, d_al.ptr, d_out.ptr, n, l)          # un-captured first: sizes the scratch
    ctx.verify_with_target_batch_async_dev(dA.ptr, d_out.ptr, d_tgt.ptr, 1, l, q, d, d_verd.ptr)
    ctx.synchronize()
    ctx.graph_begin()
    ctx.aggregate_core_dev(d_sig.ptr, d_al.ptr, d_out.ptr, n, l)
    ctx.verify_with_target_batch_async_dev(dA.ptr, d_out.ptr, d_tgt.ptr, 1, l, q, d, d_verd.ptr)
    g = ctx.graph_end()
    g.launch()
    assert np.array_equal(d_out.to_numpy(np.int32, (l, d)), want) and d_verd.to_numpy(np.int32, (1,)).tolist() in ([0], [4])
    first = d_verd.to_numpy(np.int32, (1,)).tolist()
    # larger calls on the same context: more aggregates with several slices each, more multi-workgroup verifications
    big_sig, big_al = operands(6, 40)
    b_sig, b_al, b_part = DB.from_numpy(ctx, big_sig), DB.from_numpy(ctx, big_al), DB(ctx, 6 * l * d * 8)
    ctx.aggregate_partial_batch_dev(b_sig.ptr, b_al.ptr, b_part.ptr, l * d, 6, 40, l)
    b_tgt, b_verd = DB(ctx, 40 * d * 4), DB(ctx, 40 * 4)
    many = rng.integers(-50, 50, size=(40, l, d)).astype(np.int32)
    b_many = DB.from_numpy(ctx, many)
    ctx.verify_with_target_batch_async_dev(dA.ptr, b_many.ptr, b_tgt.ptr, 40, l, q, d, b_verd.ptr)
    ctx.synchronize()
    for _ in range(3):                                   # the graph still works on its (retired) scratch
        ctx.h2d(d_out.ptr, np.zeros((l, d), np.int32))
        g.launch()
        assert np.array_equal(d_out.to_numpy(np.int32, (l, d)), want)
        assert d_verd.to_numpy(np.int32, (1,)).tolist() == first
    g.destroy()
    ctx.set_stream(0)
    ctx.stream_destroy(s)
    for b in (dA, d_sig, d_al, d_out, d_tgt, d_verd, b_sig, b_al, b_part, b_tgt, b_verd, b_many):
        b.free()
    ctx.close()


@pytest.mark.parametrize("args", [["--secpar", "128"], ["--secpar", "256", "--signatures", "5", "--distinct-seeds"]])
def test_demo_script_runs_the_reference_flow(args):
    """examples/demo.py = misc/demo.py's five calls on the drop-in package, as a user switching over would run them"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "demo.py")] + args, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.strip().endswith("Verification successful!"), r.stdout
