"""BatchQueue -- the asynchronous batch queue of the C ABI (fz_queue_*, csrc/fz_queue.hip) from Python.

The reference is called once per key / signature (fusion/fusion.py:338-373 keygen, :534-557 sign).  BatchScheme batches such
calls, but a call of BASELINE's size (1024 keys + 1024 signatures) is a latency chain that leaves most of the chip idle, and
overlapping such calls from Python needs a thread per call.  With the queue ONE Python thread submits calls and gets tickets
back at once; worker threads below Python (each with a context and a stream of its own) run whatever is pending as one batch
-- rows are independent, so every call's keys and signatures are bit-identical to keygen_batch + sign_batch of that call alone
(tests/test_gpu_queue.py).

    with BatchQueue(params) as bq:
        t = bq.submit_keygen_sign(seeds, messages)          # returns immediately
        ...
        res = bq.wait(t)                                    # res.vk (numpy, [n][2][d]), res.sig_ptr / res.sk_ptr (device)
        sig = res.signatures()                              # numpy copy [n][l][d]
        res.release()
        ta = bq.submit_aggregate_verify(vk, messages, sig)  # aggregate() + verify() of one aggregate, queued the same way
        agg, (ok, reason) = bq.wait_aggregate(ta)           # many such calls pending = one ragged launch for all of them
        tv = bq.submit_verify(vk, messages, agg)
        ok, reason = bq.wait_verdict(tv)
"""
import ctypes
from ctypes import byref, c_uint64, c_void_p

import numpy as np

from . import hostpipe
from ._lib import FZ_QUEUE_DISCARD, FZ_QUEUE_KEEP_SK, FZ_QUEUE_ROWS_ON_DEVICE, FusionHipError, QueueResult, check, load_library


class PackedMessages:
    """messages packed once for repeated submission: their UTF-8 bytes back to back + offsets (what the C ABI takes)"""

    def __init__(self, messages):
        self.blob, self.off = hostpipe._pack_messages(messages)
        self.n = len(messages)


class _Pinned:
    """a page-locked host buffer viewed as a numpy array (fz_pinned_alloc): the workers' device-to-host copies into it are
    asynchronous and run at PCIe speed"""

    def __init__(self, lib, shape, dtype=np.int32):
        self._lib, self._p = lib, c_void_p()
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        check(lib, lib.fz_pinned_alloc(nbytes, byref(self._p)))
        buf = (ctypes.c_char * nbytes).from_address(self._p.value)
        self.array = np.frombuffer(buf, dtype=dtype).reshape(shape)

    def free(self):
        if self._p:
            self.array = None
            self._lib.fz_pinned_free(self._p)
            self._p = c_void_p()


class QueueCallResult:
    """one finished call: vk [n][2][d] on the host; signatures (and, if asked for, secret keys) as device pointers owned by
    the queue until release()"""

    def __init__(self, queue, ticket, raw, vk_buf):
        self._q, self.ticket, self.n = queue, ticket, int(raw.n)
        self.sig_ptr, self.vk_ptr, self.sk_ptr = raw.d_sig, raw.d_vk, raw.d_sk_hat
        self._vk_buf = vk_buf
        self.vk = vk_buf.array if vk_buf is not None else None          # [n][2][d]: the buffer was sized for this call

    def _copy(self, ptr, shape):
        if not ptr:
            raise FusionHipError(-1, "this call's device rows were not kept (discard=True or keep_sk=False), or were released")
        out = np.empty(shape, dtype=np.int32)
        ctx = self._q._copy_ctx()
        ctx.d2h(out, ptr)
        return out

    def signatures(self):
        return self._copy(self.sig_ptr, (self.n, self._q.l, self._q.d))

    def secret_keys(self):
        return self._copy(self.sk_ptr, (self.n, 2, self._q.l, self._q.d))

    def release(self, copy_vk=False, after=None):
        """give the device rows back to the queue and the pinned vk buffer to its free list (a later call will overwrite it:
        copy_vk=True leaves `vk` as a private copy, otherwise it becomes None).
        The rows are recycled at once.  after=None: every read of sig_ptr / vk_ptr / sk_ptr must have COMPLETED (synchronise the
        context that read them first).  after=<Context>: reads still queued on that context's stream are waited for by the queue
        itself (fz_queue_release_after) -- the form for `ctx.aggregate_*_dev(res.sig_ptr, ...); res.release(after=ctx)`.
        A no-op once the queue has been closed (close() released everything)."""
        q = self._q
        if q is None:
            return
        self._q = None
        self.sig_ptr = self.vk_ptr = self.sk_ptr = None
        if not q._h:                                        # closed: the rows are gone, `vk` was detached by close()
            self._vk_buf = None
            return
        if after is not None:
            check(q._lib, q._lib.fz_queue_release_after(q._h, self.ticket, after._h))
        else:
            check(q._lib, q._lib.fz_queue_release(q._h, self.ticket))
        if self._vk_buf is not None:
            self.vk = self.vk.copy() if copy_vk else None
            q._recycle(self._vk_buf)
            self._vk_buf = None
        q._results.discard(self)


class BatchQueue:
    def __init__(self, params, device=0, workers=3, max_rows=16384, max_call=None, host_threads=8):
        """params: a fusion.fusion.Params.  workers: threads below Python, each with a context + stream of its own (3 stay
        within the HIP runtime's default 4 hardware queues).  max_rows: keys (or signers) per coalesced batch.  host_threads:
        threads a batch of aggregate / verify calls may use for its aggregates' hash_ag sponges (one aggregate per thread)."""
        self._lib = load_library()
        self.params, self.device, self.workers, self.max_rows = params, device, int(workers), int(max_rows)
        self.d, self.l = params.degree, params.num_rows_sk
        self.P = hostpipe.scheme_params(params)
        A = np.ascontiguousarray(np.array([z.values for row in params.public_challenge.matrix for z in row], dtype=np.int32))
        self._h = c_void_p()
        check(self._lib, self._lib.fz_queue_create(device, byref(self.P), self.l, int(params.beta_sk), int(params.omega_sk),
                                                   A.ctypes.data_as(c_void_p), self.workers, self.max_rows, byref(self._h)))
        # aggregate / verify calls (fusion.py:655-677, :680-728): the parameter set's verification bounds and capacity
        check(self._lib, self._lib.fz_queue_enable_aggregate(self._h, int(params.beta_vf), int(params.omega_vf), int(params.capacity),
                                                             max(1, min(32, int(host_threads)))))
        self._agg = {}                 # ticket -> (kept-alive inputs, agg output array, verdict cell)
        self._vk_free = {}             # rows -> [pinned buffers]
        self._vk_out = {}              # ticket -> pinned buffer
        self._results = set()          # results handed out and not released: close() detaches their views of pinned memory
        self._ctx = None

    # ---- lifetime -----------------------------------------------------------------------------------------
    def close(self):
        """finishes what was submitted, releases every call and joins the workers.  Results still held by the caller stay safe to
        touch: their `vk` becomes a private copy (the pinned buffer it viewed is freed here), their device pointers None, and
        release() on them is a no-op."""
        if self._h:
            self._lib.fz_queue_destroy(self._h)
            self._h = c_void_p()
            for res in list(self._results):
                if res._vk_buf is not None and res.vk is not None:
                    res.vk = res.vk.copy()
                    res._vk_buf.free()
                res._vk_buf = None
                res.sig_ptr = res.vk_ptr = res.sk_ptr = None
            self._results.clear()
            for bufs in list(self._vk_free.values()) + [list(self._vk_out.values())]:
                for b in bufs:
                    b.free()
            self._vk_free, self._vk_out = {}, {}

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _copy_ctx(self):
        if self._ctx is None:
            from .context import get_context
            p = self.params
            self._ctx = get_context(p.modulus, p.degree, p.root, p.inv_root, self.device)
        return self._ctx

    def _recycle(self, buf):
        self._vk_free.setdefault(buf.array.shape[0], []).append(buf)

    # ---- the calls ----------------------------------------------------------------------------------------
    def submit_keygen_sign(self, seeds, messages, keep_sk=False, discard=False, want_vk=True):
        """keygen(params, seeds[i]) + sign(params, key_i, messages[i]) for every i, asynchronously -> ticket.
        seeds: ints in [0, 2^64 - 1) (list or uint64 array); messages: list of str / bytes, or a PackedMessages.
        discard: drop the device results when the call has finished (throughput runs; vk still comes back)."""
        sd = seeds if isinstance(seeds, np.ndarray) and seeds.dtype == np.uint64 else None
        if sd is None:
            from .scheme import _seed_array
            sd = _seed_array(seeds)
            if sd is None:
                raise ValueError("the queue takes seeds in [0, 2^64 - 1): use BatchScheme.keygen_batch for the others")
        sd = np.ascontiguousarray(sd)
        pm = messages if isinstance(messages, PackedMessages) else PackedMessages(messages)
        n = sd.size
        if pm.n != n:
            raise ValueError("Number of seeds and messages must be equal.")
        vk_buf = None
        if want_vk:
            free = self._vk_free.get(n)
            vk_buf = free.pop() if free else _Pinned(self._lib, (n, 2, self.d))
        t = c_uint64()
        flags = (FZ_QUEUE_KEEP_SK if keep_sk else 0) | (FZ_QUEUE_DISCARD if discard else 0)
        try:
            check(self._lib, self._lib.fz_queue_submit_keygen_sign(
                self._h, sd.ctypes.data_as(c_void_p), n, pm.blob, pm.off.ctypes.data_as(c_void_p),
                vk_buf.array.ctypes.data_as(c_void_p) if vk_buf is not None else None, flags, byref(t)))
        except Exception:
            if vk_buf is not None:
                self._recycle(vk_buf)
            raise
        if vk_buf is not None:
            self._vk_out[t.value] = vk_buf
        return t.value

    def wait(self, ticket):
        """-> QueueCallResult of a finished call (blocks until it has finished)"""
        raw = QueueResult()
        try:
            check(self._lib, self._lib.fz_queue_wait(self._h, ticket, byref(raw)))
        except Exception:
            buf = self._vk_out.pop(ticket, None)
            if buf is not None:
                self._recycle(buf)
            self._lib.fz_queue_release(self._h, ticket)        # a failed call keeps nothing: forget its record
            raise
        res = QueueCallResult(self, ticket, raw, self._vk_out.pop(ticket, None))
        self._results.add(res)
        return res

    # ---- aggregate() + verify(), and verify() alone, as queued calls -------------------------------------
    def _agg_inputs(self, vk, messages):
        v = np.ascontiguousarray(np.asarray(vk, dtype=np.int32).reshape(-1, 2, self.d))
        pm = messages if isinstance(messages, PackedMessages) else PackedMessages(messages)
        if pm.n != v.shape[0]:
            raise ValueError("Number of keys and messages must be equal.")
        return v, pm

    @staticmethod
    def _rows(a, shape):
        """-> (pointer, flags, keep-alive object) of caller rows given as a numpy array, a DeviceArray / DeviceBuffer, or a raw
        device pointer (int)"""
        if isinstance(a, int):
            return c_void_p(a), FZ_QUEUE_ROWS_ON_DEVICE, None
        if hasattr(a, "ptr"):
            return c_void_p(a.ptr), FZ_QUEUE_ROWS_ON_DEVICE, a
        h = np.ascontiguousarray(np.asarray(a, dtype=np.int32).reshape(shape))
        return h.ctypes.data_as(c_void_p), 0, h

    def submit_aggregate_verify(self, vk, messages, sig):
        """aggregate(params, keys, messages, signatures) + verify(params, keys, messages, aggregate) of ONE aggregate,
        asynchronously -> ticket.  vk [n][2][d]; sig [n][l][d] as numpy (read by the worker: do not modify it before
        wait_aggregate), a DeviceArray, or a device pointer (e.g. QueueCallResult.sig_ptr of a call that has not been released)."""
        v, pm = self._agg_inputs(vk, messages)
        n = v.shape[0]
        ptr, flags, keep = self._rows(sig, (n, self.l, self.d))
        out = np.empty((self.l, self.d), dtype=np.int32)
        verdict = ctypes.c_int(-1)
        t = c_uint64()
        check(self._lib, self._lib.fz_queue_submit_aggregate_verify(self._h, v.ctypes.data_as(c_void_p), pm.blob, pm.off.ctypes.data_as(c_void_p), n,
                                                                    ptr, out.ctypes.data_as(c_void_p), byref(verdict), flags, byref(t)))
        self._agg[t.value] = (keep, out, verdict)
        return t.value

    def submit_verify(self, vk, messages, aggregate):
        """verify(params, keys, messages, aggregate_signature) asynchronously -> ticket; aggregate [l][d] (numpy / DeviceArray / pointer)"""
        v, pm = self._agg_inputs(vk, messages)
        ptr, flags, keep = self._rows(aggregate, (self.l, self.d))
        verdict = ctypes.c_int(-1)
        t = c_uint64()
        check(self._lib, self._lib.fz_queue_submit_verify(self._h, v.ctypes.data_as(c_void_p), pm.blob, pm.off.ctypes.data_as(c_void_p), v.shape[0],
                                                          ptr, byref(verdict), flags, byref(t)))
        self._agg[t.value] = (keep, None, verdict)
        return t.value

    def _wait_agg(self, ticket):
        from .context import VERDICT_REASONS
        if ticket not in self._agg:
            raise FusionHipError(-1, "not a pending aggregate / verify ticket of this queue")
        try:
            check(self._lib, self._lib.fz_queue_wait(self._h, ticket, None))
        finally:
            keep, out, verdict = self._agg.pop(ticket)
            self._lib.fz_queue_release(self._h, ticket)
        code = int(verdict.value)
        return out, (code == 0, VERDICT_REASONS[code])

    def wait_aggregate(self, ticket):
        """-> (aggregate [l][d] int32, (ok, reason)) of a submit_aggregate_verify call (blocks until it has finished)"""
        return self._wait_agg(ticket)

    def wait_verdict(self, ticket):
        """-> (ok, reason) of a submit_verify call"""
        return self._wait_agg(ticket)[1]

    def drain(self):
        """block until everything submitted has finished; raises if a discarded call failed (collect_discarded() then hands
        the verification-key buffers of calls nobody waited for back to the free list)"""
        check(self._lib, self._lib.fz_queue_drain(self._h))

    def collect_discarded(self):
        """after drain(): hand the vk buffers of discarded calls back (their tickets are then forgotten)"""
        for t in list(self._vk_out):
            self._recycle(self._vk_out.pop(t))

    def stats(self):
        """-> (calls finished, batches run, rows processed): calls / batches is the coalescing factor"""
        a, b, c = c_uint64(), c_uint64(), c_uint64()
        check(self._lib, self._lib.fz_queue_stats(self._h, byref(a), byref(b), byref(c)))
        return a.value, b.value, c.value
