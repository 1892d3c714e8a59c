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
from .context import Context, DeviceArray, VERDICT_REASONS, get_context


class BatchScheme:
    def __init__(self, params, device=0, threads=None, private_context=False):
        """params: a fusion.fusion.Params (or any object with the same attributes).
        private_context: give this object a context and a HIP stream of its own instead of the process-wide one per device.
        A context serves one host thread at a time (include/fusion_hip.h), so several BatchSchemes that are to work
        CONCURRENTLY -- one per worker thread -- each need their own.  That is how batches of the BASELINE size keep the chip
        busy: a 1024-signature sign_batch is a latency chain (108 Keccak permutations per signer on 32 waves: 0.63 of its
        0.87 ms) that leaves the other 250 CUs idle, and calls on separate streams overlap (tools/probes/concurrent_batches.py).
        close() releases the private context."""
        self.params = params
        self.P = hostpipe.scheme_params(params)
        self.threads = threads or hostpipe.default_threads()
        self._pool = None
        self.d, self.l, self.q = params.degree, params.num_rows_sk, params.modulus
        self._private = bool(private_context)
        if self._private:
            self.ctx = Context(params.modulus, params.degree, params.root % params.modulus, params.inv_root % params.modulus, device)
            self._stream = self.ctx.stream_create()
            self.ctx.set_stream(self._stream)
        else:
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
        """release the device copy of the public challenge, and the context if it is this object's own (the shared one stays)"""
        if self._dA is not None:
            self._dA.free()
            self._dA = None
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
        if self._private and self.ctx is not None:
            self.ctx.set_stream(0)
            self.ctx.stream_destroy(self._stream)
            self.ctx.close()
            self.ctx = None

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
        # random.seed(int) uses abs(seed), so negative seeds are legal in the reference; the C / device samplers take the
        # non-negative seeds below 2^64 - 1
        # (keygen seeds the right half with seed + 1: for a negative seed that is abs(seed) - 1, not abs(seed) + 1)
        sd = _seed_array(seeds)
        if sd is None:
            return self._keygen_batch_python_sampler(seeds, device, keep_vk)
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

    def _keygen_batch_python_sampler(self, seeds, device, keep_vk):
        """keygen_batch for seeds the C / device MT19937 clones do not take (negative seeds, where seed + 1 is not
        abs(seed) + 1, and seeds >= 2^64 - 1, whose keys have more words than the clones' two): the secret polynomials come
        from the drop-in's own sampler, i.e. CPython's `random` exactly as the reference drives it (polynomials.py:436-467);
        everything after the sampling is the same device path."""
        from algebra.polynomials import sample_polynomial_coefficient_representation as sample
        p = self.params
        polys = np.empty((len(seeds), 2, self.d), dtype=np.int32)
        for i, s in enumerate(seeds):
            for h in (0, 1):
                polys[i, h] = sample(modulus=p.modulus, degree=p.degree, root_order=p.root_order, root=p.root, inv_root=p.inv_root,
                                     norm_bound=p.beta_sk, weight_bound=p.omega_sk, seed=int(s) + h).coefficients
        n = len(seeds)
        coef = DeviceArray.from_numpy(self.ctx, polys)
        sk, vk = DeviceArray(self.ctx, (n, 2, self.l, self.d)), DeviceArray(self.ctx, (n, 2, self.d))
        try:
            self.ctx.keygen_core_bcast_dev(self._A_dev().ptr, coef.ptr, sk.ptr, vk.ptr, n, self.l)
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
    def _split_vk(self, vk):
        vk = np.ascontiguousarray(vk.numpy() if isinstance(vk, DeviceArray) else vk, dtype=np.int32).reshape(-1, 2, self.d)
        return vk, np.ascontiguousarray(vk[:, 0]), np.ascontiguousarray(vk[:, 1])

    def _challenges_both(self, vk, messages):
        """hash_ch for every (key, message) in the CALLERS' order: -> (dC DeviceArray [N][d], c_hat host copy, prehash
        [N][32]).  The device pipeline leaves c_hat where the kernels need it; hash_ag needs a host copy of it as text input."""
        if self.device_hash:
            try:
                dC, pre = self.challenges_dev(vk, messages)
                return dC, dC.numpy(), pre
            except FusionHipError as e:
                if e.code != FZ_E_UNSUPPORTED:
                    raise
                self.device_hash = False
        c_hat, pre = self.challenges(vk, messages)
        return DeviceArray.from_numpy(self.ctx, c_hat), c_hat, pre

    def _sort_async(self, L, R):
        """sort_by_vk_string on a worker thread (the C call releases the GIL): it needs only the keys, so it runs beside the
        device challenge pipeline instead of after it; -> a future of the order"""
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="fz-sort")
        return self._pool.submit(hostpipe.sort_by_vk_string, self.P, L, R, self.threads)

    def _alpha_coefficients(self, L, R, pre, c_hat, threads=None, order=None):
        """hash_ag without the transforms (fusion.py:632-652) for ONE aggregate: sort by str(vk) (fusion.py:661-663, :693), the
        one serial SHAKE-256 over the sorted list, decode; -> (order, alpha coefficient rows scattered back to the CALLERS'
        order).  The aggregate and the target are sums over signers, so nothing else ever has to be permuted."""
        threads = threads or self.threads
        if order is None:
            order = hostpipe.sort_by_vk_string(self.P, L, R, threads)
        alpha_sorted = hostpipe.aggregation_coefficients(self.P, L[order], R[order], pre[order], c_hat[order], threads)
        alpha = np.empty_like(alpha_sorted)
        alpha[order] = alpha_sorted
        return order, alpha

    def hash_ag_dev(self, vk, messages):
        """Everything aggregate() and verify() derive from the keys and messages, device-resident and in the callers' order:
        -> (dC [N][d] challenges c_hat, dAl [N][d] aggregation coefficients alpha_hat, order, vkL, vkR host rows)."""
        vk, L, R = self._split_vk(vk)
        order_f = self._sort_async(L, R)                    # beside the challenge pipeline: neither needs the other
        try:
            dC, c_hat, pre = self._challenges_both(vk, messages)
        except Exception:
            order_f.result()
            raise
        try:
            order, alpha = self._alpha_coefficients(L, R, pre, c_hat, order=order_f.result())
            dAl = DeviceArray.from_numpy(self.ctx, alpha)
            self.ctx.ntt_forward_dev(dAl.ptr, dAl.ptr, alpha.shape[0])          # in place
        except Exception:
            dC.free()
            raise
        return dC, dAl, order, L, R

    def aggregate(self, vk, messages, sig):
        """-> aggregate [l][d] == aggregate(params, keys, messages, signatures).signature_hat.
        sig may be a numpy array or a DeviceArray.  Challenges and aggregation coefficients stay on the device; only the
        keys' text input of hash_ag and the result cross PCIe."""
        dC, dAl, _, _, _ = self.hash_ag_dev(vk, messages)
        n = dAl.shape[0]
        dS, own = self._dev(sig, (n, self.l, self.d))
        dO = DeviceArray(self.ctx, (self.l, self.d))
        try:
            self.ctx.aggregate_core_dev(dS.ptr, dAl.ptr, dO.ptr, n, self.l)
            return dO.numpy()
        finally:
            for b in (dC, dAl, dO):
                b.free()
            if own:
                dS.free()

    def aggregate_verify(self, vk, messages, sig):
        """aggregate() followed by verify() of the result on the same keys and messages, as an aggregator that checks its own
        output does: -> (aggregate [l][d], (ok, reason)).  hash_ag -- the one serial SHAKE-256 over all signers that bounds both
        calls end to end -- runs ONCE instead of twice, the signatures are read once (aggregate and verification target in one
        pass), and the verdict comes straight from the int64 sums.  Same values as aggregate(...) and verify(..., aggregate)."""
        from .dist import LocalCollective, ShardedScheme
        return ShardedScheme(self, 0, 1, LocalCollective(self.ctx)).aggregate_verify_sharded(vk, messages, sig)

    def verify(self, vk, messages, aggregate):
        """-> (bool, reason) with the reference's reason strings (fusion.py:680-728)"""
        n = (vk.shape[0] if isinstance(vk, DeviceArray) else np.asarray(vk).reshape(-1, 2, self.d).shape[0])
        if n > self.params.capacity:
            return False, VERDICT_REASONS[1]
        if n != len(messages):
            return False, VERDICT_REASONS[2]
        dC, dAl, _, L, R = self.hash_ag_dev(vk, messages)
        dL, dR = DeviceArray.from_numpy(self.ctx, L), DeviceArray.from_numpy(self.ctx, R)
        dS, own = self._dev(aggregate, (self.l, self.d))
        try:
            code = self.ctx.verify_core_dev(self._A_dev().ptr, dS.ptr, dL.ptr, dR.ptr, dC.ptr, dAl.ptr, n, self.l,
                                            int(self.params.beta_vf), int(self.params.omega_vf))
            return code == 0, VERDICT_REASONS[code]
        finally:
            for b in (dC, dAl, dL, dR):
                b.free()
            if own:
                dS.free()

    # ---- many aggregates at once ------------------------------------------------------------------------
    def _hash_ag_many(self, vk, messages, sizes):
        """hash_ag for G independent aggregates whose signers are concatenated (aggregate g = rows off[g]:off[g+1]): ONE device
        pass for all challenges, then one host thread per aggregate for its sort + serial SHAKE-256 (the sponges are what
        bounds a lone aggregate: independent aggregates hide each other's), ONE upload + transform of all coefficients.
        -> (dC, dAl [sum N][d], offsets, L, R)"""
        from concurrent.futures import ThreadPoolExecutor
        vk, L, R = self._split_vk(vk)
        sizes = [int(x) for x in sizes]
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        if off[-1] != vk.shape[0] or vk.shape[0] != len(messages):
            raise FusionHipError(-1, f"sizes sum to {off[-1]} but {vk.shape[0]} keys and {len(messages)} messages were given")
        if any(x < 1 for x in sizes):
            raise FusionHipError(-1, "every aggregate needs at least one signer")
        dC, c_hat, pre = self._challenges_both(vk, messages)
        try:
            per = max(1, self.threads // max(1, len(sizes)))

            def one(g):
                a, b = off[g], off[g + 1]
                return self._alpha_coefficients(L[a:b], R[a:b], pre[a:b], c_hat[a:b], per)[1]
            with ThreadPoolExecutor(max_workers=min(len(sizes), self.threads)) as pool:      # the C calls release the GIL
                alpha = np.concatenate(list(pool.map(one, range(len(sizes)))))
            dAl = DeviceArray.from_numpy(self.ctx, alpha)
            self.ctx.ntt_forward_dev(dAl.ptr, dAl.ptr, alpha.shape[0])
        except Exception:
            dC.free()
            raise
        return dC, dAl, off, L, R

    def aggregate_many(self, vk, messages, sig, sizes):
        """G independent aggregate() calls (fusion.py:655; the reference is called once per aggregate) as one batch:
        vk [sum N][2][d], messages (sum N of them), sig [sum N][l][d] (numpy or DeviceArray) hold the signers of aggregate 0,
        then of aggregate 1, ...; sizes = signers per aggregate (they may differ).  -> [G][l][d]; row g equals
        aggregate(params, keys_g, messages_g, signatures_g).signature_hat.  One launch for all aggregates."""
        dC, dAl, off, _, _ = self._hash_ag_many(vk, messages, sizes)
        g = len(off) - 1
        dS, own = self._dev(sig, (int(off[-1]), self.l, self.d))
        dO = DeviceArray(self.ctx, (g, self.l, self.d))
        try:
            self.ctx.aggregate_core_ragged_dev(dS.ptr, dAl.ptr, off, self.l, dO.ptr)
            return dO.numpy()
        finally:
            for b in (dC, dAl, dO):
                b.free()
            if own:
                dS.free()

    def verify_many(self, vk, messages, aggregates, sizes):
        """G independent verify() calls (fusion.py:680-728) as one batch; aggregates [G][l][d].  -> list of (bool, reason).
        One launch for all verification targets, one for all verifications, verdicts read once."""
        sizes = [int(x) for x in sizes]
        g = len(sizes)
        out = [None] * g
        live = []
        for i, n in enumerate(sizes):                 # the reference's capacity check comes first (fusion.py:686-687)
            if n > self.params.capacity:
                out[i] = (False, VERDICT_REASONS[1])
            else:
                live.append(i)
        if len(live) != g:                            # rare: verify the rest one by one, keeping the batch path simple
            vkh, _, _ = self._split_vk(vk)
            off = np.concatenate([[0], np.cumsum(sizes)])
            aggs = aggregates.numpy() if isinstance(aggregates, DeviceArray) else np.asarray(aggregates)
            for i in live:
                out[i] = self.verify(vkh[off[i]:off[i + 1]], messages[off[i]:off[i + 1]], aggs.reshape(g, self.l, self.d)[i])
            return out
        dC, dAl, off, L, R = self._hash_ag_many(vk, messages, sizes)
        dL, dR = DeviceArray.from_numpy(self.ctx, L), DeviceArray.from_numpy(self.ctx, R)
        dS, own = self._dev(aggregates, (g, self.l, self.d))
        dT64 = DeviceArray(self.ctx, (g, self.d), np.int64)
        dT = DeviceArray(self.ctx, (g, self.d))
        dV = DeviceArray(self.ctx, (g,))
        try:
            self.ctx.aggregate_target_partial_ragged_dev(0, dAl.ptr, dL.ptr, dR.ptr, dC.ptr, off, self.l, 0, 0, dT64.ptr, self.d)
            self.ctx.reduce_i64_dev(dT64.ptr, dT.ptr, g * self.d)
            self.ctx.verify_with_target_batch_async_dev(self._A_dev().ptr, dS.ptr, dT.ptr, g, self.l, int(self.params.beta_vf),
                                                        int(self.params.omega_vf), dV.ptr)
            return [(int(c) == 0, VERDICT_REASONS[int(c)]) for c in dV.numpy()]
        finally:
            for b in (dC, dAl, dL, dR, dT64, dT, dV):
                b.free()
            if own:
                dS.free()


def _seed_array(seeds):
    """the seeds as a uint64 array when every one is a non-negative integer below 2^64 - 1 (what the C and device samplers
    take), else None.  One numpy conversion instead of a Python loop: 0.25 us per seed was most of what a 1024-key
    keygen_batch spent on the host."""
    try:
        a = np.asarray(seeds)
    except (OverflowError, ValueError, TypeError):
        a = None
    if a is not None and a.ndim == 1 and a.dtype.kind == "u":
        a = a.astype(np.uint64, copy=False)
        return a if a.size == 0 or int(a.max()) < 2 ** 64 - 1 else None
    if a is not None and a.ndim == 1 and a.dtype.kind == "i":
        return a.astype(np.uint64) if a.size == 0 or int(a.min()) >= 0 else None
    # objects, floats (numpy's choice for lists that mix negative and huge ints), nested input: element by element
    vals = [int(s) for s in seeds]
    if any(v < 0 or v >= 2 ** 64 - 1 for v in vals):
        return None
    return np.array(vals, dtype=np.uint64)


# ---- conversions between the array face and the drop-in object face ----------------------------------------
def _own(rows):
    """a private int32 copy: the objects keep their rows as they are until somebody reads their lists
    (algebra/polynomials.py, storage note), so they must not alias an array the caller may write to afterwards"""
    return np.array(rows, dtype=np.int32, copy=True)


def vk_to_object(params, vk_row):
    """vk [2][d] -> fusion.fusion.OneTimeVerificationKey"""
    import fusion.fusion as F
    vk_row = _own(vk_row)
    return F.OneTimeVerificationKey(left_vk_hat=F._column(params, vk_row[0:1]), right_vk_hat=F._column(params, vk_row[1:2]))


def sk_to_object(params, seed, sk_rows):
    """sk_hat [2][l][d] -> fusion.fusion.OneTimeSigningKey"""
    import fusion.fusion as F
    sk_rows = _own(sk_rows)
    return F.OneTimeSigningKey(seed=seed, left_sk_hat=F._column(params, sk_rows[0]), right_sk_hat=F._column(params, sk_rows[1]))


def signature_to_object(params, sig_rows):
    import fusion.fusion as F
    return F.Signature(signature_hat=F._column(params, _own(sig_rows)))


def signature_from_object(params, sig):
    import fusion.fusion as F
    return F._rows_of(sig.signature_hat, params.modulus)


def vk_from_object(params, vk):
    import fusion.fusion as F
    return np.concatenate([F._rows_of(vk.left_vk_hat, params.modulus), F._rows_of(vk.right_vk_hat, params.modulus)])
