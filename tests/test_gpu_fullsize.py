"""GPU parity AT THE STATED SIZES of BASELINE.json configs[2], configs[3] and the SURVEY section 9 form of configs[4]:
the fused scheme kernels pick their grid shape from (CUs, aggregates, signers), so a grid that only ever ran at
N <= 64 signers in the tests says nothing about the N = 1024 launch of the bench.  Everything goes through the C ABI
(device-pointer entry points) and is compared with the CPU oracle on the same inputs.

Reference arithmetic: fusion/fusion.py:363-370 (keygen), :557 (sign), :670-676 (aggregate), :686-727 (verify)."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
Q = O.PRIME


def _cent(x, q=Q):
    y = np.asarray(x, dtype=np.int64) % q
    return np.where(y > q // 2, y - q, y)


def _sparse(rng, n, d, weight):
    c = np.zeros((n, d), np.int32)
    for i in range(n):
        c[i, rng.choice(d, weight, replace=False)] = rng.choice(np.array([-1, 1], np.int32), weight)
    return c


def _target_oracle(vkL, vkR, c_hat, alpha, q=Q):
    """sum_i (vkL_i * c_i + vkR_i) * alpha_i mod q, centred (fusion.py:706-714), in int64 numpy"""
    t = _cent(vkL.astype(np.int64) * c_hat.astype(np.int64)) + vkR.astype(np.int64)
    t = (_cent(t) * alpha.astype(np.int64)) % q
    return _cent(t.sum(axis=0))


class Dev:
    """a few device buffers with automatic release"""

    def __init__(self, ctx):
        self.ctx, self.bufs = ctx, []

    def put(self, arr):
        import fusion_hip
        b = fusion_hip.DeviceBuffer.from_numpy(self.ctx, np.ascontiguousarray(arr))
        self.bufs.append(b)
        return b

    def new(self, nbytes):
        import fusion_hip
        b = fusion_hip.DeviceBuffer(self.ctx, nbytes)
        self.bufs.append(b)
        return b

    def close(self):
        for b in self.bufs:
            b.free()


def _signers(ctx, P, n, seed, distinct_rows=False):
    """n valid (key, signature) pairs made on the device: -> dict of host arrays + device buffers"""
    q, d, l = P["q"], P["d"], P["rank"]
    rng = np.random.default_rng(seed)
    dev = Dev(ctx)
    A = O.splitmix_centered(seed, l * d).reshape(l, d)
    dA = dev.put(A)
    dsk = dev.new(n * 2 * l * d * 4)
    dvk = dev.new(n * 2 * d * 4)
    if distinct_rows:
        coef = (rng.integers(1, 53, size=(n, 2, l, d)) * rng.choice(np.array([-1, 1]), size=(n, 2, l, d))).astype(np.int32)
        dc = dev.put(coef)
        ctx.keygen_core_dev(dA.ptr, dc.ptr, dsk.ptr, dvk.ptr, n, l)
    else:   # what the reference's seeded sampler produces: one polynomial per (key, half)
        coef = (rng.integers(1, 53, size=(n, 2, d)) * rng.choice(np.array([-1, 1]), size=(n, 2, d))).astype(np.int32)
        dc = dev.put(coef)
        ctx.keygen_core_bcast_dev(dA.ptr, dc.ptr, dsk.ptr, dvk.ptr, n, l)
    vk = dvk.to_numpy(np.int32, (n, 2, d))
    c_hat = ctx.ntt_forward(_sparse(rng, n, d, P["omega_ch"]))
    al_hat = ctx.ntt_forward(_sparse(rng, n, d, P["omega_ag"]))
    dch, dal = dev.put(c_hat), dev.put(al_hat)
    dsig = dev.new(n * l * d * 4)
    ctx.sign_core_dev(dsk.ptr, dch.ptr, dsig.ptr, n, l)
    return dict(dev=dev, A=A, dA=dA, coef=coef, dsk=dsk, vk=vk, c_hat=c_hat, al_hat=al_hat, dch=dch, dal=dal, dsig=dsig,
                dvkL=dev.put(vk[:, 0]), dvkR=dev.put(vk[:, 1]))


def test_config2_keygen_and_sign_on_1024_distinct_keys(coracle):
    """BASELINE configs[2]: secpar 256, 1024 keygen + sign, whole arrays against the oracle"""
    import fusion_hip
    P = O.PARAMS[256]
    q, d, l = P["q"], P["d"], P["rank"]
    ctx = fusion_hip.Context(q, d, P["root"], P["inv_root"])
    n = 1024
    S = _signers(ctx, P, n, 2026, distinct_rows=True)
    try:
        sk = S["dsk"].to_numpy(np.int32, (n, 2, l, d))
        rsk, rvk = coracle.keygen_core(S["A"], S["coef"], q, P["root"])
        assert np.array_equal(sk, rsk), "sk_hat differs from the oracle"
        assert np.array_equal(S["vk"], rvk), "vk differs from the oracle"
        sig = S["dsig"].to_numpy(np.int32, (n, l, d))
        assert np.array_equal(sig, coracle.sign_core(rsk, S["c_hat"], q)), "signatures differ from the oracle"
    finally:
        S["dev"].close()


@pytest.mark.parametrize("secpar,n", [(256, 128), (256, 256), (256, 1024), (256, 2048), (256, 2818), (128, 1796), (128, 1024), (128, 128)])
def test_config3_aggregate_and_verify_at_full_size(secpar, n, coracle):
    """BASELINE configs[3] sizes from one rank's share at 8 GPUs (128 signers: the slice-free aggregation kernel takes the
    launch) up to the capacity (fusion.py:24-25): aggregate_core, the one-pass int64 partials of
    aggregate + target, fz_reduce_i64, verification from the int64 partials -- each against the oracle, with a tampered
    aggregate per size and the two-shard form of the multi-GPU exchange (partials of two signer blocks added on the host)."""
    import fusion_hip
    P = O.PARAMS[secpar]
    q, d, l = P["q"], P["d"], P["rank"]
    ctx = fusion_hip.Context(q, d, P["root"], P["inv_root"])
    S = _signers(ctx, P, n, 7 * n + secpar)
    dev = S["dev"]
    try:
        sig = S["dsig"].to_numpy(np.int32, (n, l, d))
        ref_agg = coracle.aggregate_core(sig, S["al_hat"], q)
        ref_tgt = _target_oracle(S["vk"][:, 0], S["vk"][:, 1], S["c_hat"], S["al_hat"], q)
        # (1) centred aggregate in one call
        dout = dev.new(l * d * 4)
        ctx.aggregate_core_dev(S["dsig"].ptr, S["dal"].ptr, dout.ptr, n, l)
        agg = dout.to_numpy(np.int32, (l, d))
        assert np.array_equal(agg, ref_agg), "aggregate differs from the oracle"
        # (2) int64 partials of aggregate and target in one pass, then centring
        dpart, dtp = dev.new(l * d * 8), dev.new(d * 8)
        ctx.aggregate_target_partial_batch_dev(S["dsig"].ptr, S["dal"].ptr, S["dvkL"].ptr, S["dvkR"].ptr, S["dch"].ptr,
                                               dpart.ptr, l * d, dtp.ptr, d, 1, n, l)
        part = dpart.to_numpy(np.int64, (l, d))
        tp = dtp.to_numpy(np.int64, (d,))
        assert np.abs(part).max() < n * (q // 2) and np.array_equal(_cent(part, q), ref_agg)
        assert np.array_equal(_cent(tp, q), ref_tgt), "target partial differs from the oracle"
        dred = dev.new(l * d * 4)
        ctx.reduce_i64_dev(dpart.ptr, dred.ptr, l * d)
        assert np.array_equal(dred.to_numpy(np.int32, (l, d)), ref_agg)
        # (3) verdict straight from the int64 sums, and the oracle's verdict on the same aggregate
        dver = dev.put(np.full(1, -1, np.int32))
        ctx.verify_partials_batch_async_dev(S["dA"].ptr, dpart.ptr, l * d, dtp.ptr, d, 1, l, P["beta_vf"], d, dver.ptr)
        ref_v = coracle.verify_core(S["A"], ref_agg, S["vk"][:, 0], S["vk"][:, 1], S["c_hat"], S["al_hat"], q, P["inv_root"],
                                    P["beta_vf"], d)
        assert dver.to_numpy(np.int32, (1,))[0] == ref_v == 0
        assert ctx.verify_core_dev(S["dA"].ptr, dout.ptr, S["dvkL"].ptr, S["dvkR"].ptr, S["dch"].ptr, S["dal"].ptr, n, l,
                                   P["beta_vf"], d) == 0
        # (4) tampered aggregate: one coefficient off by one (tests/test_fusion.py:860-873)
        bad = part.copy()
        bad[l // 2, d // 3] += 1
        dbad = dev.put(bad)
        ctx.verify_partials_batch_async_dev(S["dA"].ptr, dbad.ptr, l * d, dtp.ptr, d, 1, l, P["beta_vf"], d, dver.ptr)
        bad32 = ref_agg.copy()
        bad32[l // 2, d // 3] = int(_cent(int(bad32[l // 2, d // 3]) + 1, q))
        assert dver.to_numpy(np.int32, (1,))[0] == 3 == coracle.verify_core(
            S["A"], bad32, S["vk"][:, 0], S["vk"][:, 1], S["c_hat"], S["al_hat"], q, P["inv_root"], P["beta_vf"], d)
        # (5) two shards, as two ranks would hold them: partials of each block, added (the all-reduce), verified
        cut = n // 2 + 3
        parts, tps = [], []
        for lo_, hi_ in ((0, cut), (cut, n)):
            m = hi_ - lo_
            off = lo_ * d * 4
            ctx.aggregate_target_partial_batch_dev(S["dsig"].ptr + lo_ * l * d * 4, S["dal"].ptr + off, S["dvkL"].ptr + off,
                                                   S["dvkR"].ptr + off, S["dch"].ptr + off, dpart.ptr, l * d, dtp.ptr, d, 1, m, l)
            parts.append(dpart.to_numpy(np.int64, (l, d)))
            tps.append(dtp.to_numpy(np.int64, (d,)))
        dsum, dtsum = dev.put(parts[0] + parts[1]), dev.put(tps[0] + tps[1])
        ctx.verify_partials_batch_async_dev(S["dA"].ptr, dsum.ptr, l * d, dtsum.ptr, d, 1, l, P["beta_vf"], d, dver.ptr)
        assert dver.to_numpy(np.int32, (1,))[0] == 0
        ctx.reduce_i64_dev(dsum.ptr, dred.ptr, l * d)
        assert np.array_equal(dred.to_numpy(np.int32, (l, d)), ref_agg)
    finally:
        dev.close()


def test_config3_four_aggregates_of_256_like_the_bench(coracle):
    """the bench's sign_verify leg at N = 1 rank: 4 aggregates of 256 signers in one groups=4 launch"""
    import fusion_hip
    P = O.PARAMS[256]
    q, d, l = P["q"], P["d"], P["rank"]
    ctx = fusion_hip.Context(q, d, P["root"], P["inv_root"])
    G, per = 4, 256
    S = _signers(ctx, P, G * per, 99)
    dev = S["dev"]
    try:
        sig = S["dsig"].to_numpy(np.int32, (G * per, l, d))
        dpart, dtp = dev.new(G * l * d * 8), dev.new(G * d * 8)
        ctx.aggregate_target_partial_batch_dev(S["dsig"].ptr, S["dal"].ptr, S["dvkL"].ptr, S["dvkR"].ptr, S["dch"].ptr,
                                               dpart.ptr, l * d, dtp.ptr, d, G, per, l)
        part, tp = dpart.to_numpy(np.int64, (G, l, d)), dtp.to_numpy(np.int64, (G, d))
        for g in range(G):
            sl = slice(g * per, (g + 1) * per)
            assert np.array_equal(_cent(part[g], q), coracle.aggregate_core(sig[sl], S["al_hat"][sl], q)), g
            assert np.array_equal(_cent(tp[g], q), _target_oracle(S["vk"][sl, 0], S["vk"][sl, 1], S["c_hat"][sl], S["al_hat"][sl], q))
        dver = dev.put(np.full(G, -1, np.int32))
        ctx.verify_partials_batch_async_dev(S["dA"].ptr, dpart.ptr, l * d, dtp.ptr, d, G, l, P["beta_vf"], d, dver.ptr)
        assert dver.to_numpy(np.int32, (G,)).tolist() == [0] * G
        # the bench's step as it runs since round 4: signing + aggregation + target sums in ONE pass into records
        # [l*d sums | d target sums] per aggregate -- the same signatures, sums equal as residues, the same verdicts
        rec = l * d + d
        dsig2, drec = dev.new(G * per * l * d * 4), dev.put(np.full(G * rec, -1, np.int64))
        ctx.sign_aggregate_target_partial_batch_dev(S["dsk"].ptr, S["dch"].ptr, S["dal"].ptr, S["dvkL"].ptr, S["dvkR"].ptr, dsig2.ptr,
                                                    drec.ptr, rec, drec.ptr + l * d * 8, rec, G, per, l)
        assert np.array_equal(dsig2.to_numpy(np.int32, (G * per, l, d)), sig)
        r = drec.to_numpy(np.int64, (G, rec))
        assert np.array_equal(_cent(r[:, :l * d].reshape(G, l, d), q), _cent(part, q)) and np.array_equal(_cent(r[:, l * d:], q), _cent(tp, q))
        ctx.verify_partials_batch_async_dev(S["dA"].ptr, drec.ptr, rec, drec.ptr + l * d * 8, rec, G, l, P["beta_vf"], d, dver.ptr)
        assert dver.to_numpy(np.int32, (G,)).tolist() == [0] * G
        bad = r.copy()
        bad[2, 17] += 1
        ctx.h2d(drec.ptr, bad)
        ctx.verify_partials_batch_async_dev(S["dA"].ptr, drec.ptr, rec, drec.ptr + l * d * 8, rec, G, l, P["beta_vf"], d, dver.ptr)
        assert dver.to_numpy(np.int32, (G,)).tolist() == [0, 0, 3, 0]
    finally:
        dev.close()


def test_config5_as_survey_section9_two_aggregates_of_2048(coracle):
    """BASELINE configs[4] names secpar=512 / N=4096, which the reference cannot express (fusion.py:71,97: parameter sets
    128 and 256 only; q - 1 = 2^9 * odd admits no degree-512 negacyclic NTT; verify rejects N > 2818, :686-687).
    SURVEY section 9 resolves it as secpar-256 parameters with N = 4096 split into two aggregates of 2048 -- a
    NON-REFERENCE signer count, labelled as such: full keygen -> sign -> aggregate -> verify, one groups=2 launch."""
    import fusion_hip
    P = O.PARAMS[256]
    q, d, l = P["q"], P["d"], P["rank"]
    ctx = fusion_hip.Context(q, d, P["root"], P["inv_root"])
    G, per = 2, 2048
    assert per <= 2818 < G * per
    S = _signers(ctx, P, G * per, 4096)
    dev = S["dev"]
    try:
        dpart, dtp = dev.new(G * l * d * 8), dev.new(G * d * 8)
        ctx.aggregate_target_partial_batch_dev(S["dsig"].ptr, S["dal"].ptr, S["dvkL"].ptr, S["dvkR"].ptr, S["dch"].ptr,
                                               dpart.ptr, l * d, dtp.ptr, d, G, per, l)
        dver = dev.put(np.full(G, -1, np.int32))
        ctx.verify_partials_batch_async_dev(S["dA"].ptr, dpart.ptr, l * d, dtp.ptr, d, G, l, P["beta_vf"], d, dver.ptr)
        assert dver.to_numpy(np.int32, (G,)).tolist() == [0, 0]
        part = dpart.to_numpy(np.int64, (G, l, d))
        sig = S["dsig"].to_numpy(np.int32, (G * per, l, d))
        for g in range(G):
            sl = slice(g * per, (g + 1) * per)
            ref = coracle.aggregate_core(sig[sl], S["al_hat"][sl], q)
            assert np.array_equal(_cent(part[g], q), ref)
            assert coracle.verify_core(S["A"], ref, S["vk"][sl, 0], S["vk"][sl, 1], S["c_hat"][sl], S["al_hat"][sl], q,
                                       P["inv_root"], P["beta_vf"], d) == 0
        # the second aggregate checked against the FIRST one's signers must fail, as the oracle says
        dtp2 = dev.put(dtp.to_numpy(np.int64, (G, d))[::-1].copy())
        ctx.verify_partials_batch_async_dev(S["dA"].ptr, dpart.ptr, l * d, dtp2.ptr, d, G, l, P["beta_vf"], d, dver.ptr)
        assert dver.to_numpy(np.int32, (G,)).tolist() == [3, 3]
    finally:
        dev.close()


def test_reduce_i64_is_exact_for_any_int64(coracle):
    """sums that crossed an all-reduce are not bounded by 2^53: fz_reduce_i64 and the int64 verify loads reduce in
    integer-exact steps (the int64 -> double conversion was exact only below 2^53)"""
    import fusion_hip
    P = O.PARAMS[256]
    q, d = P["q"], P["d"]
    ctx = fusion_hip.Context(q, d, P["root"], P["inv_root"])
    rng = np.random.default_rng(5)
    v = np.concatenate([rng.integers(-2**63, 2**63 - 1, size=4000, dtype=np.int64),
                        np.array([2**63 - 1, -2**63, 2**53 + 1, -(2**53) - 1, 2**62 + 12345, q * (2**31) + 7, 0, -1], np.int64)])
    expect = np.array([int(_cent(int(x) % q, q)) for x in v], dtype=np.int32)
    dev = Dev(ctx)
    try:
        din, dout = dev.put(v), dev.new(v.size * 4)
        ctx.reduce_i64_dev(din.ptr, dout.ptr, v.size)
        assert np.array_equal(dout.to_numpy(np.int32, v.shape), expect)
    finally:
        dev.close()


@pytest.mark.parametrize("knob", ["FZ_AGG_DIRECT=-1", "FZ_AGG_DIRECT=2", "FZ_AGG_DIRECT=4", "FZ_VERIFY_ORDERED=1"])
def test_knobbed_paths_agree_with_the_oracle(knob, coracle, monkeypatch):
    """the A/B paths kept behind environment knobs (read once at context creation) compute the same integers"""
    import fusion_hip
    name, val = knob.split("=")
    monkeypatch.setenv(name, val)
    P = O.PARAMS[256]
    q, d, l = P["q"], P["d"], P["rank"]
    ctx = fusion_hip.Context(q, d, P["root"], P["inv_root"])       # a fresh context: the knob is read here
    monkeypatch.delenv(name)
    n = 203
    S = _signers(ctx, P, n, 31)
    dev = S["dev"]
    try:
        sig = S["dsig"].to_numpy(np.int32, (n, l, d))
        dout = dev.new(l * d * 4)
        ctx.aggregate_core_dev(S["dsig"].ptr, S["dal"].ptr, dout.ptr, n, l)
        assert np.array_equal(dout.to_numpy(np.int32, (l, d)), coracle.aggregate_core(sig, S["al_hat"], q))
        for _ in range(3):          # repeated launches: accumulators and tickets re-arm themselves
            assert ctx.verify_core_dev(S["dA"].ptr, dout.ptr, S["dvkL"].ptr, S["dvkR"].ptr, S["dch"].ptr, S["dal"].ptr, n, l,
                                       P["beta_vf"], d) == 0
            ctx.aggregate_core_dev(S["dsig"].ptr, S["dal"].ptr, dout.ptr, n, l)
            assert np.array_equal(dout.to_numpy(np.int32, (l, d)), coracle.aggregate_core(sig, S["al_hat"], q))
    finally:
        dev.close()
        ctx.close()


def test_config2_sweep_maximum_2_22_rows(coracle):
    """BASELINE configs[1]'s sweep ends at B = 2^22 rows (4 GiB in, 4 GiB out at degree 256).  At that size: sampled rows
    against the C oracle (first, middle, last 2048 rows of the seeded stream), the round trip INTT(NTT(x)) == x over ALL
    rows (difference reduced on the device: max |.| and weight per row), and linearity NTT(x + y) == NTT(x) + NTT(y)."""
    import fusion_hip
    P = O.PARAMS[256]
    q, d = P["q"], P["d"]
    ctx = fusion_hip.Context(q, d, P["root"], P["inv_root"])
    rows = 1 << 22
    count = rows * d
    DB = fusion_hip.DeviceBuffer
    GAMMA = 0x9E3779B97F4A7C15

    def host_rows(seed, first, n):          # the same stream as fz_fill_synthetic, from element first * d on
        return O.splitmix_centered((seed + first * d * GAMMA) % (1 << 64), n * d).reshape(n, d)
    dx, dy, dz = DB(ctx, count * 4), DB(ctx, count * 4), DB(ctx, count * 4)
    dm, dw = DB(ctx, rows * 8), DB(ctx, rows * 4)          # fz_norm_weight: int64 maxima, int32 weights
    try:
        ctx.fill_synthetic_dev(dx.ptr, count, 5)
        ctx.ntt_forward_dev(dx.ptr, dy.ptr, rows)
        for first in (0, 1234567, rows - 2048):
            want = coracle.ntt_forward(host_rows(5, first, 2048), q, P["root"]).reshape(2048, d)
            got = np.empty((2048, d), np.int32)
            ctx.d2h(got, dy.ptr + first * d * 4)
            assert np.array_equal(got, want), first

        def all_zero(ptr):
            ctx.norm_weight_dev(ptr, rows, dm.ptr, dw.ptr)
            return not dm.to_numpy(np.int64, (rows,)).any() and not dw.to_numpy(np.int32, (rows,)).any()
        ctx.ntt_inverse_dev(dy.ptr, dy.ptr, rows)                       # in place
        ctx.pw_dev(fusion_hip.OP_SUB, dy.ptr, dx.ptr, dy.ptr, count)
        assert all_zero(dy.ptr), "INTT(NTT(x)) != x somewhere in 2^22 rows"
        ctx.fill_synthetic_dev(dz.ptr, count, 6)
        ctx.pw_dev(fusion_hip.OP_ADD, dx.ptr, dz.ptr, dy.ptr, count)    # x + y
        ctx.ntt_forward_dev(dy.ptr, dy.ptr, rows)
        ctx.ntt_forward_dev(dx.ptr, dx.ptr, rows)
        ctx.ntt_forward_dev(dz.ptr, dz.ptr, rows)
        ctx.pw_dev(fusion_hip.OP_ADD, dx.ptr, dz.ptr, dx.ptr, count)
        ctx.pw_dev(fusion_hip.OP_SUB, dy.ptr, dx.ptr, dy.ptr, count)
        assert all_zero(dy.ptr), "NTT(x + y) != NTT(x) + NTT(y) somewhere in 2^22 rows"
    finally:
        for b in (dx, dy, dz, dm, dw):
            b.free()
        ctx.close()


@pytest.mark.parametrize("secpar", [256, 128])
def test_poly_mul_at_2_20_products_equals_the_composed_launches(secpar, coracle):
    """fz_poly_mul at 2^20 products (the 16-per-lane kernel at degree 256, the radix-4 kernel at degree 64): over ALL rows equal
    to INTT(NTT(f) (.) NTT(g)) made of the transform and pointwise launches (difference reduced on the device), f * g == g * f,
    and sampled rows (first, middle, last 512) against the C oracle (algebra/ntt.py:380-484)."""
    import fusion_hip
    P = O.PARAMS[secpar]
    q, d = P["q"], P["d"]
    ctx = fusion_hip.Context(q, d, P["root"], P["inv_root"])
    rows = 1 << 20
    count = rows * d
    DB = fusion_hip.DeviceBuffer
    GAMMA = 0x9E3779B97F4A7C15

    def host_rows(seed, first, n):
        return O.splitmix_centered((seed + first * d * GAMMA) % (1 << 64), n * d).reshape(n, d)
    df, dg, dp, dc = (DB(ctx, count * 4) for _ in range(4))
    dm, dw = DB(ctx, rows * 8), DB(ctx, rows * 4)
    try:
        ctx.fill_synthetic_dev(df.ptr, count, 15)
        ctx.fill_synthetic_dev(dg.ptr, count, 16)
        ctx.poly_mul_dev(df.ptr, dg.ptr, dp.ptr, rows)
        for first in (0, 345678, rows - 512):
            f, g = host_rows(15, first, 512), host_rows(16, first, 512)
            want = coracle.ntt_inverse(coracle.pw_mul(coracle.ntt_forward(f, q, P["root"]), coracle.ntt_forward(g, q, P["root"]), q), q, P["inv_root"])
            got = np.empty((512, d), np.int32)
            ctx.d2h(got, dp.ptr + first * d * 4)
            assert np.array_equal(got, want.reshape(512, d)), first

        def all_zero(ptr):
            ctx.norm_weight_dev(ptr, rows, dm.ptr, dw.ptr)
            return not dm.to_numpy(np.int64, (rows,)).any() and not dw.to_numpy(np.int32, (rows,)).any()
        ctx.poly_mul_dev(dg.ptr, df.ptr, dc.ptr, rows)                   # g * f
        ctx.pw_dev(fusion_hip.OP_SUB, dc.ptr, dp.ptr, dc.ptr, count)
        assert all_zero(dc.ptr), "f * g != g * f somewhere in 2^20 products"
        ctx.ntt_forward_dev(df.ptr, df.ptr, rows)                        # the composed launches, in place
        ctx.ntt_forward_dev(dg.ptr, dg.ptr, rows)
        ctx.pw_dev(fusion_hip.OP_MUL, df.ptr, dg.ptr, df.ptr, count)
        ctx.ntt_inverse_dev(df.ptr, df.ptr, rows)
        ctx.pw_dev(fusion_hip.OP_SUB, df.ptr, dp.ptr, df.ptr, count)
        assert all_zero(df.ptr), "fz_poly_mul != INTT(NTT f (.) NTT g) somewhere in 2^20 products"
    finally:
        for b in (df, dg, dp, dc, dm, dw):
            b.free()
        ctx.close()
