"""Many DISTINCT signers against the reference (tests/golden/scheme_many_*: 32 signers at secpar 128, 16 at 256, nested
sub-aggregates): everything that depends on the ORDER of many keys -- sorted(key=str(vk)) on signed decimals
(fusion.py:661-663, :693), the hash_ag text over N tuples (:586-591), the scatter of alpha back to the callers' order -- through
the array API (BatchScheme), the many-aggregates batch (aggregate_many / verify_many) and the drop-in object API."""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def load(secpar):
    S = np.load(os.path.join(G, f"scheme_many_{secpar}.npz"))
    with open(os.path.join(G, "scheme_many.json")) as fh:
        return S, json.load(fh)[str(secpar)]


def sha_i32(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<i4").tobytes()).hexdigest()


def sha_str(s):
    return hashlib.sha256(s.encode("utf-8")).hexdigest()


@pytest.mark.parametrize("secpar", [128, 256])
def test_batch_scheme_at_many_signers_equals_the_reference(secpar):
    import fusion.fusion as F
    from fusion_hip import hostpipe
    from fusion_hip.scheme import BatchScheme
    S, m = load(secpar)
    params = F.fusion_setup(secpar, m["setup_seed"])
    bs = BatchScheme(params)
    assert np.array_equal(bs.A, S["A"])
    sk, vk = bs.keygen_batch(m["key_seeds"])
    assert np.array_equal(vk, S["vk"])
    sig = bs.sign_batch(sk, vk, m["messages"])
    assert [sha_i32(r) for r in sig] == m["sha256_sig_rows"]
    for lo, hi in m["subsets"]:
        info = m["agg"][f"{lo}_{hi}"]
        v, ms, sg = vk[lo:hi], m["messages"][lo:hi], sig[lo:hi]
        order = hostpipe.sort_by_vk_string(bs.P, v[:, 0], v[:, 1], 3)
        assert order.tolist() == info["order"]                                   # the reference's sorted(key=str(vk))
        dC, dAl, order2, _, _ = bs.hash_ag_dev(v, ms)
        alpha = dAl.numpy()
        dC.free()
        dAl.free()
        assert order2.tolist() == info["order"]
        assert np.array_equal(alpha[order], S[f"alpha_hat_sorted_{lo}_{hi}"])     # hash_ag's output, in the reference's order
        agg = bs.aggregate(v, ms, sg)
        assert np.array_equal(agg, S[f"agg_{lo}_{hi}"])
        assert list(bs.verify(v, ms, agg)) == info["verdict"]
        agg1, verdict1 = bs.aggregate_verify(v, ms, sg)                          # one hash_ag, one pass over the signatures
        assert np.array_equal(agg1, agg) and list(verdict1) == info["verdict"]
        bad = agg.copy()
        bad[info["tampered_at"][0], info["tampered_at"][1]] += 1
        assert list(bs.verify(v, ms, bad)) == info["tampered_verdict"]
        sw = list(ms)
        sw[0], sw[-1] = sw[-1], sw[0]
        assert list(bs.verify(v, sw, agg)) == info["swapped_messages_verdict"]
    bs.close()


@pytest.mark.parametrize("secpar", [128, 256])
def test_many_aggregates_in_one_batch_equal_the_reference(secpar):
    """aggregate_many / verify_many: G = 3 aggregates of DIFFERENT sizes in one launch each"""
    import fusion.fusion as F
    from fusion_hip.scheme import BatchScheme
    S, m = load(secpar)
    params = F.fusion_setup(secpar, m["setup_seed"])
    bs = BatchScheme(params)
    sk, vk, vk_dev = bs.keygen_batch(m["key_seeds"], device=True, keep_vk=True)
    sig = bs.sign_batch(sk, vk_dev, m["messages"], device=True)                  # signatures stay on the device
    parts = m["subsets"][1:]                                                      # three consecutive blocks covering all signers
    assert parts[0][0] == 0 and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
    sizes = [hi - lo for lo, hi in parts]
    aggs = bs.aggregate_many(vk, m["messages"], sig, sizes)
    for g, (lo, hi) in enumerate(parts):
        assert np.array_equal(aggs[g], S[f"agg_{lo}_{hi}"]), f"aggregate {g} ({lo}:{hi})"
    assert bs.verify_many(vk, m["messages"], aggs, sizes) == [(True, "")] * 3
    bad = aggs.copy()
    bad[1, 2, 5] += 1
    want = [(True, ""), (False, "Target doesn't match image of aggregate signature."), (True, "")]
    assert bs.verify_many(vk, m["messages"], bad, sizes) == want
    # one aggregate as a "batch" of one, and the whole set again through the single-aggregate entry
    lo, hi = m["subsets"][0]
    assert np.array_equal(bs.aggregate_many(vk, m["messages"], sig, [hi - lo])[0], S[f"agg_{lo}_{hi}"])
    assert np.array_equal(bs.aggregate(vk, m["messages"], sig), S[f"agg_{lo}_{hi}"])
    sk.free()
    vk_dev.free()
    sig.free()


def test_more_aggregates_than_one_launch_holds():
    """70 small aggregates (the group table of a launch holds 64): equal to 70 single calls"""
    import fusion.fusion as F
    from fusion_hip.scheme import BatchScheme
    params = F.fusion_setup(128, 9)
    bs = BatchScheme(params, threads=4)
    sizes = [1 + (g * 7) % 5 for g in range(70)]
    n = sum(sizes)
    seeds = [40_000 + 3 * i for i in range(n)]
    msgs = [f"m{i}" for i in range(n)]
    sk, vk = bs.keygen_batch(seeds)
    sig = bs.sign_batch(sk, vk, msgs)
    aggs = bs.aggregate_many(vk, msgs, sig, sizes)
    off = np.concatenate([[0], np.cumsum(sizes)])
    for g in (0, 1, 33, 63, 64, 69):
        a, b = off[g], off[g + 1]
        assert np.array_equal(aggs[g], bs.aggregate(vk[a:b], msgs[a:b], sig[a:b])), g
    verdicts = bs.verify_many(vk, msgs, aggs, sizes)
    assert verdicts == [(True, "")] * 70
    aggs[64, 0, 0] += 1
    assert [v[0] for v in bs.verify_many(vk, msgs, aggs, sizes)] == [g != 64 for g in range(70)]


def test_drop_in_object_api_at_32_signers_prints_the_reference_strings():
    """fusion.fusion keygen / sign / aggregate / verify on 32 distinct signers: the hashed str() of every key, signature and
    of the aggregate equal the reference's (sorted(key=str(vk)) and hash_ag over 32 tuples included)"""
    import fusion.fusion as F
    S, m = load(128)
    params = F.fusion_setup(128, m["setup_seed"])
    keys = [F.keygen(params, s) for s in m["key_seeds"]]
    assert [sha_str(str(k[1])) for k in keys] == m["sha256_str_vk"]
    sigs = [F.sign(params, k, msg) for k, msg in zip(keys, m["messages"])]
    assert [sha_str(str(s)) for s in sigs] == m["sha256_str_sig"]
    vks = [k[1] for k in keys]
    for lo, hi in m["subsets"]:
        info = m["agg"][f"{lo}_{hi}"]
        agg = F.aggregate(params, vks[lo:hi], m["messages"][lo:hi], sigs[lo:hi])
        assert sha_str(str(agg)) == info["sha256_str_agg"]
        assert list(F.verify(params, vks[lo:hi], m["messages"][lo:hi], agg)) == info["verdict"]
        r, c = info["tampered_at"]
        agg.signature_hat.matrix[r][0].values[c] += 1
        assert list(F.verify(params, vks[lo:hi], m["messages"][lo:hi], agg)) == info["tampered_verdict"]


def _full(tag):
    p = os.path.join(G, f"scheme_full_{tag}.npz")
    if not os.path.exists(p):
        pytest.skip(f"tests/golden/scheme_full_{tag}.npz not generated (gen_golden.py full / full128 / full256cap)")
    with open(os.path.join(G, "scheme_full.json")) as fh:
        return np.load(p), json.load(fh)[tag]


@pytest.mark.parametrize("tag", ["256", "128", "256cap"])
def test_full_size_flows_equal_the_reference(tag):
    """AT FULL SIZE against the REFERENCE ITSELF: BASELINE configs[3] (1024 distinct signers at secpar 256) and both parameter sets
    at their CAPACITY (1796 signers at secpar 128, 2818 at secpar 256: fusion.py:24-25) -- the reference's keygen, sign, ONE aggregate() and ONE verify() over all of them
    (tests/golden/gen_golden.py full / full128: ~5 minutes each on 8 cores).  Keys, signatures and aggregation coefficients
    are compared by SHA-256 of the whole arrays, the sort order and the aggregate element by element, verdict, tamper verdict
    and the "Too many keys." verdict of capacity + 1 signers literally."""
    import fusion.fusion as F
    from fusion_hip.scheme import BatchScheme
    S, m = _full(tag)
    secpar = m["secpar"]
    params = F.fusion_setup(secpar, m["setup_seed"])
    bs = BatchScheme(params)
    sk, vk, vk_dev = bs.keygen_batch(m["key_seeds"], device=True, keep_vk=True)
    assert sha_i32(vk) == m["sha256_vk"]
    sig = bs.sign_batch(sk, vk_dev, m["messages"], device=True)
    sig_host = sig.numpy()
    assert sha_i32(sig_host) == m["sha256_sig"]
    assert [sha_i32(r) for r in sig_host[:8]] == m["sha256_sig_rows_first8"]
    dC, dAl, order, _, _ = bs.hash_ag_dev(vk, m["messages"])
    assert np.array_equal(order, S["order"])                                     # sorted(key=str(vk)) over 1024 keys
    assert sha_i32(dAl.numpy()[order]) == m["sha256_alpha_hat_sorted"]             # hash_ag over 1024 tuples
    # signing + aggregation in ONE pass over ALL signers (fz_sign_aggregate_target_partial_batch): the reference's signatures by
    # SHA-256 and the reference's aggregate element by element, from the keys, challenges and coefficients alone
    from fusion_hip.context import DeviceArray
    n_, l_, d_, q_ = m["n"], params.num_rows_sk, params.degree, params.modulus
    sig1, sums = DeviceArray(bs.ctx, (n_, l_, d_)), DeviceArray(bs.ctx, (l_ * d_,), np.int64)
    bs.ctx.sign_aggregate_target_partial_batch_dev(sk.ptr, dC.ptr, dAl.ptr, 0, 0, sig1.ptr, sums.ptr, l_ * d_, 0, 0, 1, n_, l_)
    assert sha_i32(sig1.numpy()) == m["sha256_sig"]
    assert np.array_equal(((sums.numpy() + q_ // 2) % q_ - q_ // 2).astype(np.int32).reshape(l_, d_), S["agg"])
    sig1.free()
    sums.free()
    dC.free()
    dAl.free()
    agg = bs.aggregate(vk, m["messages"], sig)
    assert np.array_equal(agg, S["agg"])
    assert list(bs.verify(vk, m["messages"], agg)) == m["verdict"] == [True, ""]
    agg1, v1 = bs.aggregate_verify(vk, m["messages"], sig)
    assert np.array_equal(agg1, S["agg"]) and list(v1) == m["verdict"]
    bad = agg.copy()
    bad[m["tampered_at"][0], m["tampered_at"][1]] += 1
    assert list(bs.verify(vk, m["messages"], bad)) == m["tampered_verdict"]
    if m.get("too_many_verdict"):                                                # capacity + 1 signers (fusion.py:686-687)
        assert list(bs.verify(np.concatenate([vk, vk[:1]]), m["messages"] + m["messages"][:1], agg)) == m["too_many_verdict"] \
            == [False, "Too many keys."]
    # the same signers as 4 aggregates in one batch: each equals a single call on its block
    q4 = m["n"] // 4
    sizes = [q4, q4, q4, m["n"] - 3 * q4]
    aggs = bs.aggregate_many(vk, m["messages"], sig, sizes)
    for g in (0, 3):
        a_, b_ = q4 * g, q4 * g + sizes[g]
        assert np.array_equal(aggs[g], bs.aggregate(vk[a_:b_], m["messages"][a_:b_], sig_host[a_:b_]))
    assert bs.verify_many(vk, m["messages"], aggs, sizes) == [(True, "")] * 4
    for b in (sk, vk_dev, sig):
        b.free()
