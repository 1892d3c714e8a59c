"""Array-backed batch API of the scheme (SURVEY.md 8f rows N1 + N2): keys, challenges and signatures
are int32 arrays instead of lists of polynomial objects, hashing/decoding runs in the C host pipeline
(hostpipe) and all algebra on the device.  Results are the same integers the object API
(fusion.fusion.keygen / sign / aggregate / verify) produces -- tests/test_gpu_batch_scheme.py checks both
against the reference's golden arrays.

Layouts: sk_hat [N][2][l][d] (left rows then right rows), vk [N][2][d], c_hat / alpha_hat [N][d],
sig [N][l][d], aggregate [l][d]; all int32, centred.
"""
import numpy as np

from . import hostpipe
from ._lib import FZ_E_UNSUPPORTED, FusionHipError
from .context import DeviceArray, VERDICT_REASONS, get_context


class BatchScheme:
    def __init__(self, params, device=0, threads=None):
        """params: a fusion.fusion.Params (or any object with the same attributes)."""
        self.params = params
        self.P = hostpipe.scheme_params(params)
        self.threads = threads or hostpipe.default_threads()
        self.d, self.l, self.q = params.degree, params.num_rows_sk, params.modulus
        self.ctx = get_context(params.modulus, params.degree, params.root, params.inv_root, device)
        self.device_hash = True          # per-signer challenge pipeline on the device (falls back to the host if unsupported)
        self.device_sampler = True       # secret polynomials sampled on the device (the same fallback)
        self.A = np.array([z.values for row in params.public_challenge.matrix for z in row], dtype=np.int32)
        self._dA = None                  # the public challenge, resident on the device after its first use

    def _A_dev(self):
        if self._dA is None:
            self._dA = DeviceArray.from_numpy(self.ctx, self.A)
        return self._dA

    def close(self):
        """release the device copy of the public challenge (the context itself is shared and stays)"""
        if self._dA is not None:
            self._dA.free()
            self._dA = None

    # ---- keygen ------------------------------------------------------------------------------------
    def _dev(self, a, shape):
        """numpy array or DeviceArray -> (DeviceArray, owned?)"""
        if isinstance(a, DeviceArray):
            return a, False
        return DeviceArray.from_numpy(self.ctx, np.ascontiguousarray(a, dtype=np.int32).reshape(shape)), True

    def keygen_batch(self, seeds, device=False, keep_vk=False):
        """-> (sk_hat [N][2][l][d], vk [N][2][d]); key i equals keygen(params, seeds[i]).  With device=True
        sk_hat stays in device memory (a DeviceArray) -- vk always comes back (hash_ag hashes it on the host); with
        keep_vk=True a third value is returned, the verification keys as a DeviceArray [N][2][d] (what sign_batch's
        device challenge pipeline reads: pass it as `vk` and the keys are not uploaded again).
        Sampling: the reference draws every entry of a secret matrix with the SAME seed (fusion.py:156-173),
        so a matrix is one polynomial repeated l times; the polynomial itself comes from a clone of CPython's
        MT19937 `random` -- on the device (self.device_sampler) or in C on the host (hostpipe.sample_secret_polys); both
        are pinned against `random` in the tests.
        Unlike the reference this does not leave the process-global `random` generator re-seeded."""
        p = self.params
        sd = np.array([int(s) for s in seeds], dtype=np.uint64)
        n = sd.size
        # the reference's sampler yields ONE polynomial per (key, half) for all l rows (same seed for every matrix
        # entry): 2 KiB per key instead of 166 KiB.  Sampled on the device (an exact MT19937 per lane, fz_sample.hip) when
        # the parameter set allows (weight bound = degree), else by the C clone on the host threads and uploaded.
        coef = None
        if self.device_sampler:
            coef = DeviceArray(self.ctx, (n, 2, self.d))
            try:
                self.ctx.sample_secret_polys_dev(sd, p.modulus, p.degree, p.beta_sk, p.omega_sk, coef.ptr)
            except FusionHipError as e:
                coef.free()
                coef = None
                if e.code != FZ_E_UNSUPPORTED:
                    raise
                self.device_sampler = False
        if coef is None:
            polys = hostpipe.sample_secret_polys(sd, p.modulus, p.degree, p.beta_sk, p.omega_sk, self.threads)   # [N][2][d]
            coef = DeviceArray.from_numpy(self.ctx, polys)
        dA = self._A_dev()
        sk = DeviceArray(self.ctx, (n, 2, self.l, self.d))
        vk = DeviceArray(self.ctx, (n, 2, self.d))
        try:
            self.ctx.keygen_core_bcast_dev(dA.ptr, coef.ptr, sk.ptr, vk.ptr, n, self.l)
            vk_host = vk.numpy()
            res_sk = sk
            if not device:
                res_sk = sk.numpy()
                sk.free()
            if keep_vk:
                keep, vk = vk, None
                return res_sk, vk_host, keep
            return res_sk, vk_host
        finally:
            for b in (coef, vk):
                if b is not None:
                    b.free()

    # ---- sign --------------------------------------------------------------------------------------
    def challenges_dev(self, vk, messages, want_prehash=True):
        """hash_ch for every (key, message) ON THE DEVICE (fz_challenge_hat_msgs_dev: SHA3-256 of the messages, text of
        str(vk), SHAKE-256, decoder, forward NTT): -> (c_hat DeviceArray [N][d], prehash [N][32] or None).  vk: numpy
        [N][2][d] or a DeviceArray of that shape.  Raises FusionHipError(FZ_E_UNSUPPORTED) for parameter sets the device
        pipeline does not cover."""
        blob, off = hostpipe._pack_messages(messages)
        n = len(messages)
        dV, own = self._dev(vk, (n, 2, self.d))
        dC = DeviceArray(self.ctx, (n, self.d))
        try:
            pre = self.ctx.challenge_msgs_dev(self.P, dV.ptr, blob, off, n, dC.ptr, want_prehash)
        except Exception:
            dC.free()
            raise
        finally:
            if own:
                self.ctx.synchronize()
                dV.free()
        return dC, pre

    def challenges(self, vk, messages):
        """hash_ch for every (key, message): -> (c_hat [N][d], prehash [N][32]).  The per-signer pipeline runs on the
        device when it covers the parameter set (self.device_hash), else in the C host pipeline."""
        if self.device_hash:
            try:
                dC, pre = self.challenges_dev(vk, messages)
                out = dC.numpy()
                dC.free()
                return out, pre
            except FusionHipError as e:
                if e.code != FZ_E_UNSUPPORTED:
                    raise
                self.device_hash = False
        vk = np.ascontiguousarray(vk.numpy() if isinstance(vk, DeviceArray) else vk, dtype=np.int32).reshape(-1, 2, self.d)
        coefs, pre = hostpipe.challenge_coefficients(self.P, np.ascontiguousarray(vk[:, 0]),
                                                     np.ascontiguousarray(vk[:, 1]), messages, self.threads)
        return self.ctx.ntt_forward(coefs), pre

    def sign_batch(self, sk_hat, vk, messages, device=False):
        """-> sig [N][l][d]; row i equals sign(params, key_i, messages[i]).signature_hat.
        sk_hat may be a numpy array or a DeviceArray; with device=True the signatures stay on the device.
        With the device challenge pipeline the challenges never visit the host."""
        dC = None
        if self.device_hash:
            try:
                dC, _ = self.challenges_dev(vk, messages, want_prehash=False)
            except FusionHipError as e:
                if e.code != FZ_E_UNSUPPORTED:
                    raise
                self.device_hash = False
        if dC is None:
            c_hat, _ = self.challenges(vk, messages)
            dC = DeviceArray.from_numpy(self.ctx, c_hat)
        n = dC.shape[0]
        dK, own = self._dev(sk_hat, (n, 2, self.l, self.d))
        dS = DeviceArray(self.ctx, (n, self.l, self.d))
        try:
            self.ctx.sign_core_dev(dK.ptr, dC.ptr, dS.ptr, n, self.l)
            if device:
                self.ctx.synchronize()
                return dS
            out = dS.numpy()
            dS.free()
            return out
        finally:
            dC.free()
            if own:
                dK.free()

    # ---- aggregate / verify ----------------------------------------------------------------------------
    def _sorted_inputs(self, vk, messages):
        vk = np.ascontiguousarray(vk, dtype=np.int32).reshape(-1, 2, self.d)
        L, R = np.ascontiguousarray(vk[:, 0]), np.ascontiguousarray(vk[:, 1])
        order = hostpipe.sort_by_vk_string(self.P, L, R, self.threads)
        L, R = L[order], R[order]
        msgs = [messages[i] for i in order]
        c_hat, pre = self.challenges(np.stack([L, R], axis=1), msgs)
        alpha = hostpipe.aggregation_coefficients(self.P, L, R, pre, c_hat, self.threads)
        return order, L, R, c_hat, self.ctx.ntt_forward(alpha)

    def aggregate(self, vk, messages, sig):
        """-> aggregate [l][d] == aggregate(params, keys, messages, signatures).signature_hat.
        sig may be a numpy array or a DeviceArray.  The aggregation coefficients are derived in sorted key
        order (as the reference does) and scattered back to the callers' order: the aggregate is a sum, so
        the signatures themselves never have to be permuted."""
        order, _, _, _, alpha_sorted = self._sorted_inputs(vk, messages)
        n = alpha_sorted.shape[0]
        alpha = np.empty_like(alpha_sorted)
        alpha[order] = alpha_sorted
        dS, own = self._dev(sig, (n, self.l, self.d))
        dA = DeviceArray.from_numpy(self.ctx, alpha)
        dO = DeviceArray(self.ctx, (self.l, self.d))
        try:
            self.ctx.aggregate_core_dev(dS.ptr, dA.ptr, dO.ptr, n, self.l)
            return dO.numpy()
        finally:
            dA.free()
            dO.free()
            if own:
                dS.free()

    def verify(self, vk, messages, aggregate):
        """-> (bool, reason) with the reference's reason strings (fusion.py:680-728)"""
        n = np.asarray(vk).reshape(-1, 2, self.d).shape[0]
        if n > self.params.capacity:
            return False, VERDICT_REASONS[1]
        if n != len(messages):
            return False, VERDICT_REASONS[2]
        _, L, R, c_hat, alpha_hat = self._sorted_inputs(vk, messages)
        code = self.ctx.verify_core(self.A, aggregate, L, R, c_hat, alpha_hat, self.params.beta_vf,
                                    self.params.omega_vf)
        return code == 0, VERDICT_REASONS[code]


# ---- conversions between the array face and the drop-in object face ----------------------------------------
def vk_to_object(params, vk_row):
    """vk [2][d] -> fusion.fusion.OneTimeVerificationKey"""
    import fusion.fusion as F
    return F.OneTimeVerificationKey(left_vk_hat=F._column(params, np.asarray(vk_row[0:1])),
                                    right_vk_hat=F._column(params, np.asarray(vk_row[1:2])))


def sk_to_object(params, seed, sk_rows):
    """sk_hat [2][l][d] -> fusion.fusion.OneTimeSigningKey"""
    import fusion.fusion as F
    return F.OneTimeSigningKey(seed=seed, left_sk_hat=F._column(params, np.asarray(sk_rows[0])),
                               right_sk_hat=F._column(params, np.asarray(sk_rows[1])))


def signature_to_object(params, sig_rows):
    import fusion.fusion as F
    return F.Signature(signature_hat=F._column(params, np.asarray(sig_rows)))


def signature_from_object(params, sig):
    import fusion.fusion as F
    return F._rows_of(sig.signature_hat, params.modulus)


def vk_from_object(params, vk):
    import fusion.fusion as F
    return np.concatenate([F._rows_of(vk.left_vk_hat, params.modulus), F._rows_of(vk.right_vk_hat, params.modulus)])
