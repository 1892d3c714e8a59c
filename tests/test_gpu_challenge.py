"""GPU parity of the device challenge pipeline (fz_challenge_*_dev: text of str(vk) -> SHAKE-256 -> decoder -> NTT) against
 * the reference's own KAT rows (intermediate_hash_ch_KAT_128.csv, via tests/golden/kat.json),
 * the reference's golden challenges of tests/golden/scheme_{128,256}.npz,
 * the host pipeline (fz_challenge_coefficients, itself pinned by CPython's hashlib and the KATs) on 1024+ random keys.
Reference: fusion/fusion.py:412-419 (hash), :422-481 (decoder), :484-531 (parse_challenge / hash_ch)."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _setup(secpar):
    import fusion.fusion as F
    import fusion_hip
    from fusion_hip import hostpipe
    params = F.fusion_setup(secpar, 7)
    P = hostpipe.scheme_params(params)
    ctx = fusion_hip.get_context(params.modulus, params.degree, params.root, params.inv_root)
    return params, P, ctx


def _device_challenges(ctx, P, vk, prehash, transform):
    import fusion_hip
    n, d = vk.shape[0], vk.shape[2]
    dvk = fusion_hip.DeviceBuffer.from_numpy(ctx, np.ascontiguousarray(vk, dtype=np.int32))
    dout = fusion_hip.DeviceBuffer(ctx, n * d * 4)
    ctx.challenge_dev(P, dvk.ptr, prehash, n, dout.ptr, transform=transform)
    out = dout.to_numpy(np.int32, (n, d))
    dvk.free()
    dout.free()
    return out


@pytest.mark.parametrize("secpar,n", [(128, 1), (128, 65), (256, 1), (256, 33), (256, 1024), (128, 1500)])
def test_device_pipeline_equals_host_pipeline(secpar, n, coracle):
    from fusion_hip import hostpipe
    params, P, ctx = _setup(secpar)
    d, q = params.degree, params.modulus
    rng = np.random.default_rng(secpar + n)
    vk = rng.integers(-(q // 2), q // 2 + 1, size=(n, 2, d)).astype(np.int32)
    # every length of decimal text: tiny, zero and extreme values in some keys
    vk[0, 0, :8] = [0, 1, -1, 9, -10, 99999, -100000, q // 2]
    vk[-1, 1, -4:] = [-(q // 2), 1000000000, -999999999, 0]
    msgs = [f"message {i}" * (1 + i % 3) for i in range(n)]
    coefs, pre = hostpipe.challenge_coefficients(P, vk[:, 0], vk[:, 1], msgs)
    assert np.array_equal(pre, hostpipe.hash_messages(P, msgs))
    pre[0] = 0                                               # str(int) of the pre-hash: 1 digit ...
    if n > 1:
        pre[1] = 255                                         # ... and the full 78
    # host pipeline with the edited pre-hashes, through its generic pieces
    for i in range(min(n, 2)):
        text = (bytes(P.sign_hash_dst) + b"," + hostpipe.format_vk(P, vk[i, 0], vk[i, 1]).encode() + b"," +
                str(int.from_bytes(bytes(pre[i]), "little")).encode())
        nbytes = 8 * 1024 * 2
        coefs[i] = hostpipe.decode_coefficients(hostpipe.shake256(text, nbytes), P.secpar, q, d, params.beta_ch, params.omega_ch)
    got = _device_challenges(ctx, P, vk, pre, transform=False)
    assert np.array_equal(got, coefs), np.argwhere((got != coefs).any(axis=1))[:5]
    assert (np.abs(got).sum(axis=1) == params.omega_ch).all()
    hat = _device_challenges(ctx, P, vk, pre, transform=True)
    assert np.array_equal(hat, coracle.ntt_forward(coefs, q, params.root))


def test_reference_kat_rows_hash_ch_128():
    """the 20 rows of the reference's intermediate_hash_ch_KAT_128.csv (the in-tree KAT that pins hash_ch bit-exactly)"""
    import fusion.fusion as F
    import fusion_hip
    from fusion_hip import hostpipe
    with open(os.path.join(G, "kat.json")) as fh:
        rows = json.load(fh)["hash_ch"]
    params = F.fusion_setup(128, 1)
    P = hostpipe.scheme_params(params)
    ctx = fusion_hip.get_context(params.modulus, params.degree, params.root, params.inv_root)
    vk = np.array([[r["vk_left"], r["vk_right"]] for r in rows], dtype=np.int64).astype(np.int32)
    pre = hostpipe.hash_messages(P, [r["message"] for r in rows])
    hat = _device_challenges(ctx, P, vk, pre, transform=True)
    assert hat.tolist() == [r["c_hat"] for r in rows]


@pytest.mark.parametrize("secpar", [128, 256])
def test_golden_scheme_challenges(secpar):
    """c_hat of the reference's setup -> keygen -> sign flow (tests/golden/scheme_*.npz)"""
    import fusion.fusion as F
    import fusion_hip
    from fusion_hip import hostpipe
    with open(os.path.join(G, "scheme.json")) as fh:
        m = json.load(fh)[str(secpar)]
    S = np.load(os.path.join(G, f"scheme_{secpar}.npz"))
    params = F.fusion_setup(secpar, m["setup_seed"])
    P = hostpipe.scheme_params(params)
    ctx = fusion_hip.get_context(params.modulus, params.degree, params.root, params.inv_root)
    pre = hostpipe.hash_messages(P, m["messages"])
    hat = _device_challenges(ctx, P, S["vk"], pre, transform=True)
    assert np.array_equal(hat, S["c_hat"])


@pytest.mark.parametrize("secpar", [128, 256])
def test_messages_hashed_on_the_device(secpar, coracle):
    """fz_challenge_hat_msgs_dev: hash_message_to_int (fusion.py:405-409) on the device too.  The digests it returns are
    CPython's hashlib.sha3_256 of dst + "," + message for every block-boundary length (rate 136: 3 prefix bytes, one
    suffix byte), empty and multi-block messages and non-ASCII text; the challenges equal the host pipeline's."""
    import hashlib
    import fusion_hip
    from fusion_hip import hostpipe
    params, P, ctx = _setup(secpar)
    d, q = params.degree, params.modulus
    lengths = list(range(0, 8)) + list(range(128, 140)) + [264, 265, 266, 267, 268, 269, 270, 271, 272, 273, 400, 1000, 5000]
    msgs = ["".join(chr(33 + (7 * i + k) % 90) for k in range(n)) for i, n in enumerate(lengths)]
    msgs += ["café ✓ \U0001f511" * 9, "ü" * 66, "ü" * 67]          # UTF-8: 2- to 4-byte characters, 132 / 134 bytes
    msgs += [f"synthetic message {i:06d}" for i in range(70)]
    n = len(msgs)
    rng = np.random.default_rng(secpar)
    vk = rng.integers(-(q // 2), q // 2 + 1, size=(n, 2, d)).astype(np.int32)
    blob, off = hostpipe._pack_messages(msgs)
    dvk = fusion_hip.DeviceBuffer.from_numpy(ctx, vk)
    dout = fusion_hip.DeviceBuffer(ctx, n * d * 4)
    try:
        pre = ctx.challenge_msgs_dev(P, dvk.ptr, blob, off, n, dout.ptr, want_prehash=True)
        dst = bytes(P.sign_pre_hash_dst)
        for i, m in enumerate(msgs):
            assert bytes(pre[i]) == hashlib.sha3_256(dst + b"," + m.encode("utf-8")).digest(), (i, len(m))
        assert np.array_equal(pre, hostpipe.hash_messages(P, msgs))
        coefs, _ = hostpipe.challenge_coefficients(P, vk[:, 0], vk[:, 1], msgs)
        assert np.array_equal(dout.to_numpy(np.int32, (n, d)), coracle.ntt_forward(coefs, q, params.root))
        # offsets with a non-zero origin (a slice of a larger packing), no digests requested
        ctx.h2d(dout.ptr, np.zeros((n, d), np.int32))
        k = 40
        assert ctx.challenge_msgs_dev(P, dvk.ptr + k * 2 * d * 4, blob, off[k:], n - k, dout.ptr + k * d * 4) is None
        assert np.array_equal(dout.to_numpy(np.int32, (n, d))[k:], coracle.ntt_forward(coefs[k:], q, params.root))
    finally:
        dvk.free()
        dout.free()


@pytest.mark.parametrize("form", [1, 2, 3])
@pytest.mark.parametrize("secpar,n", [(128, 3), (128, 200), (256, 5), (256, 1024), (256, 1027)])
def test_every_form_of_the_device_pipeline_agrees_with_the_host(form, secpar, n, coracle, monkeypatch):
    """FZ_SHAKE_FORM = 1 (a Keccak state on a lane pair), 2 (on a lane), 3 (on a wave, the whole pipeline of a signer fused in
    one kernel: fz_keccak_wave.h): the same digests (hashlib) and the same challenges as the host pipeline, from messages and
    from digests, for signer counts that leave the last workgroup / wave partly empty."""
    import hashlib
    import fusion.fusion as F
    import fusion_hip
    from fusion_hip import hostpipe
    params = F.fusion_setup(secpar, 11)
    P = hostpipe.scheme_params(params)
    monkeypatch.setenv("FZ_SHAKE_FORM", str(form))
    ctx = fusion_hip.Context(params.modulus, params.degree, params.root, params.inv_root)      # a fresh context: the knob is read here
    monkeypatch.delenv("FZ_SHAKE_FORM")
    d, q = params.degree, params.modulus
    rng = np.random.default_rng(1000 * form + n)
    vk = rng.integers(-(q // 2), q // 2 + 1, size=(n, 2, d)).astype(np.int32)
    vk[0, 0, :8] = [0, 1, -1, 9, -10, 99999, -100000, q // 2]
    vk[-1, 1, -4:] = [-(q // 2), 1000000000, -999999999, 0]
    vk[n // 2] = 0                                           # the shortest text: every value one digit
    vk[n // 3] = -(q // 2)                                   # the longest: every value a sign and ten digits
    lens = [0, 1, 131, 132, 133, 134, 267, 268, 269, 700]
    msgs = ["".join(chr(33 + (5 * i + k) % 90) for k in range(lens[i % len(lens)])) for i in range(n)]
    coefs, pre = hostpipe.challenge_coefficients(P, vk[:, 0], vk[:, 1], msgs)
    want = coracle.ntt_forward(coefs, q, params.root).reshape(n, d)
    blob, off = hostpipe._pack_messages(msgs)
    dvk = fusion_hip.DeviceBuffer.from_numpy(ctx, vk)
    dout = fusion_hip.DeviceBuffer(ctx, n * d * 4)
    try:
        got_pre = ctx.challenge_msgs_dev(P, dvk.ptr, blob, off, n, dout.ptr, want_prehash=True)
        dst = bytes(P.sign_pre_hash_dst)
        for i in (0, 1, 2, n - 1):
            assert bytes(got_pre[i]) == hashlib.sha3_256(dst + b"," + msgs[i].encode()).digest()
        assert np.array_equal(got_pre, pre)
        got = dout.to_numpy(np.int32, (n, d))
        assert np.array_equal(got, want), np.argwhere((got != want).any(axis=1))[:5]
        ctx.h2d(dout.ptr, np.zeros((n, d), np.int32))
        ctx.challenge_dev(P, dvk.ptr, pre, n, dout.ptr, transform=False)
        assert np.array_equal(dout.to_numpy(np.int32, (n, d)), coefs)
    finally:
        dvk.free()
        dout.free()
        ctx.close()


def test_message_entry_rejects_bad_arguments():
    import fusion_hip
    from fusion_hip._lib import FZ_E_BADARG
    params, P, ctx = _setup(256)
    d = params.degree
    dvk = fusion_hip.DeviceBuffer(ctx, 2 * 2 * d * 4)
    dout = fusion_hip.DeviceBuffer(ctx, 2 * d * 4)
    ctx.h2d(dvk.ptr, np.zeros((2, 2, d), np.int32))
    try:
        with pytest.raises(fusion_hip.FusionHipError) as e:
            ctx.challenge_msgs_dev(P, dvk.ptr, b"abcdef", np.array([0, 4, 2], dtype=np.uintp), 2, dout.ptr)     # offsets decrease
        assert e.value.code == FZ_E_BADARG
        with pytest.raises(fusion_hip.FusionHipError):
            ctx.challenge_msgs_dev(P, dvk.ptr, b"abcdef", np.array([0, 4], dtype=np.uintp), 2, dout.ptr)        # N + 1 offsets needed
        # empty batch and all-empty messages are fine
        assert ctx.challenge_msgs_dev(P, dvk.ptr, b"", np.array([0], dtype=np.uintp), 0, dout.ptr) is None
        pre = ctx.challenge_msgs_dev(P, dvk.ptr, b"", np.array([0, 0, 0], dtype=np.uintp), 2, dout.ptr, want_prehash=True)
        import hashlib
        assert bytes(pre[0]) == bytes(pre[1]) == hashlib.sha3_256(bytes(P.sign_pre_hash_dst) + b",").digest()
    finally:
        dvk.free()
        dout.free()


def test_more_signers_than_one_pass_holds(coracle):
    """66 313 signers: the pipeline runs in passes of 65 536 (scratch bound), the second pass starts at a non-zero base in the
    keys, the messages, the offsets and the output.  Rows around the pass boundary, the last rows and a random sample
    against the host pipeline, for both entry points (messages / pre-hashed messages)."""
    import fusion_hip
    from fusion_hip import hostpipe
    params, P, ctx = _setup(256)
    d, q = params.degree, params.modulus
    n = 65536 + 777
    rng = np.random.default_rng(99)
    vk = rng.integers(-(q // 2), q // 2 + 1, size=(n, 2, d), dtype=np.int64).astype(np.int32)
    msgs = [f"m{i}" * (1 + i % 5) for i in range(n)]
    pick = np.unique(np.concatenate([np.arange(0, 4), np.arange(65530, 65542), np.arange(n - 4, n), rng.integers(0, n, 40)]))
    coefs, pre_pick = hostpipe.challenge_coefficients(P, vk[pick, 0], vk[pick, 1], [msgs[i] for i in pick])
    want = coracle.ntt_forward(coefs, q, params.root).reshape(len(pick), d)
    blob, off = hostpipe._pack_messages(msgs)
    dvk = fusion_hip.DeviceBuffer.from_numpy(ctx, vk)
    dout = fusion_hip.DeviceBuffer(ctx, n * d * 4)
    try:
        pre = ctx.challenge_msgs_dev(P, dvk.ptr, blob, off, n, dout.ptr, want_prehash=True)
        assert np.array_equal(pre[pick], pre_pick)
        got = dout.to_numpy(np.int32, (n, d))
        assert np.array_equal(got[pick], want)
        ctx.h2d(dout.ptr + 65530 * d * 4, np.zeros((20, d), np.int32))
        ctx.challenge_dev(P, dvk.ptr, pre, n, dout.ptr, transform=True)
        assert np.array_equal(dout.to_numpy(np.int32, (n, d)), got)
    finally:
        dvk.free()
        dout.free()


@pytest.mark.parametrize("form", [1, 3])
@pytest.mark.parametrize("d,omega,secpar", [(16, 5, 128), (64, 63, 128), (64, 64, 128), (256, 1, 256), (128, 64, 256), (8, 3, 64), (4, 1, 40),
                                            (32, 31, 200), (256, 64, 96)])
def test_device_pipeline_on_parameters_beyond_the_two_sets(form, d, omega, secpar, monkeypatch):
    """The device pipeline takes any degree 4..256, any weight (the wave form: up to 64) and any secpar the decoder's chunk
    buffers hold: weights that leave no shuffle draw (omega = d - 1), no shuffle at all (omega = d) or one coefficient, index chunks
    of 6 to 33 bytes, texts from 8 to 512 values -- lane-pair and wave forms against the host pipeline (fusion.py:422-481, :511-531)"""
    import types
    import fusion_hip
    from fusion_hip import hostpipe
    q = O.PRIME
    root = pow(O.PARAMS[256]["root"], 256 // d, q)            # a primitive 2d-th root of unity
    inv = pow(root, q - 2, q)
    prm = types.SimpleNamespace(modulus=q, root=root, inv_root=inv, degree=d, root_order=2 * d, secpar=secpar, omega_ch=omega, omega_ag=omega,
                                beta_ch=1, beta_ag=1, bytes_for_one_coef_bdd_by_beta_ch=0, bytes_for_poly_shuffle=0,
                                sign_pre_hash_dst=bytes([9, 0]), sign_hash_dst=bytes([9, 1]), agg_xof_dst=bytes([9, 2]))
    P = hostpipe.scheme_params(prm)
    monkeypatch.setenv("FZ_SHAKE_FORM", str(form))
    ctx = fusion_hip.Context(q, d, root, inv)
    monkeypatch.delenv("FZ_SHAKE_FORM")
    n = 37
    rng = np.random.default_rng(d * 1000 + omega + secpar + form)
    vk = rng.integers(-(q // 2), q // 2 + 1, size=(n, 2, d)).astype(np.int32)
    vk[3] = 0
    msgs = [f"m{i}" * (i % 7) for i in range(n)]
    coefs, pre = hostpipe.challenge_coefficients(P, vk[:, 0], vk[:, 1], msgs)
    assert (np.abs(coefs).sum(axis=1) == min(omega, d)).all() and set(np.unique(coefs)) <= {-1, 0, 1}
    blob, off = hostpipe._pack_messages(msgs)
    dvk = fusion_hip.DeviceBuffer.from_numpy(ctx, vk)
    dout = fusion_hip.DeviceBuffer(ctx, n * d * 4)
    try:
        got_pre = ctx.challenge_msgs_dev(P, dvk.ptr, blob, off, n, dout.ptr, want_prehash=True)
        assert np.array_equal(got_pre, pre)
        assert np.array_equal(dout.to_numpy(np.int32, (n, d)), ctx.ntt_forward(coefs))
        ctx.challenge_dev(P, dvk.ptr, pre, n, dout.ptr, transform=False)
        assert np.array_equal(dout.to_numpy(np.int32, (n, d)), coefs)
    finally:
        dvk.free()
        dout.free()
        ctx.close()
