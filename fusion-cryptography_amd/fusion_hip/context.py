"""Context: one (device, modulus, degree, root) instance of the HIP library.

Two faces:
  * device face  -- methods taking raw device pointers (ints, e.g. ``tensor.data_ptr()``);
    nothing is copied, kernels are enqueued on the context's stream (throughput path).
  * host face    -- methods taking/returning numpy int32 arrays; staged through the
    library's own scratch (used by the drop-in object API in ``algebra`` / ``fusion``).
"""
import collections
import ctypes
from ctypes import byref, c_double, c_float, c_int, c_int32, c_int64, c_uint32, c_void_p

import numpy as np

from ._lib import FZ_E_BADARG, FusionHipError, NttJob, UniqueId, check, load_library

_I32P = ctypes.POINTER(c_int32)
_I64P = ctypes.POINTER(c_int64)

OP_MUL, OP_ADD, OP_SUB, OP_NEG = 0, 1, 2, 3

VERDICT_REASONS = {
    0: "",
    1: "Too many keys.",
    2: "Number of keys and messages must be equal.",
    3: "Target doesn't match image of aggregate signature.",
    4: "Norm of aggregate signature too large.",
    5: "Weight of aggregate signature too large.",
}


def _as_i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a


def _p(a):
    return a.ctypes.data_as(_I32P)


class Context:
    def __init__(self, modulus, degree, root, inv_root, device=0, tables=None):
        """tables=(forward, inverse): the context's twiddle tables are these lists (fz_ctx_create_tables) instead of the bit-reversed
        powers of root / inv_root"""
        self._lib = load_library()
        self._h = c_void_p()
        self.modulus, self.degree, self.root, self.inv_root = modulus, degree, root, inv_root
        self.device = device
        if not (0 < modulus < 2 ** 32):
            raise FusionHipError(FZ_E_BADARG, f"modulus {modulus} outside (0, 2^32)")
        if tables is not None:
            fwd, inv = (np.ascontiguousarray(np.asarray(t, dtype=np.uint64) % np.uint64(modulus), dtype=np.uint32) for t in tables)
            if fwd.shape != (degree,) or inv.shape != (degree,):
                raise FusionHipError(FZ_E_BADARG, f"twiddle tables must have {degree} entries each")
            u32p = ctypes.POINTER(ctypes.c_uint32)
            check(self._lib, self._lib.fz_ctx_create_tables(device, modulus, degree, fwd.ctypes.data_as(u32p), inv.ctypes.data_as(u32p),
                                                            byref(self._h)))
            return
        check(self._lib, self._lib.fz_ctx_create(device, modulus, degree, root % modulus, inv_root % modulus,
                                                 byref(self._h)))

    # -- lifetime -------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.fz_ctx_destroy(self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        check(self._lib, self._lib.fz_ctx_set_stream(self._h, c_void_p(stream_ptr)))

    def synchronize(self):
        check(self._lib, self._lib.fz_ctx_synchronize(self._h))

    def stream_create(self, priority=None):
        """-> raw hipStream_t (int) on this context's device; priority "high" / "low": the device's highest / lowest stream
        priority (small latency-bound launches beside chip-filling ones)"""
        p = c_void_p()
        if priority is None:
            check(self._lib, self._lib.fz_stream_create(self._h, byref(p)))
        else:
            check(self._lib, self._lib.fz_stream_create_priority(self._h, 1 if priority == "high" else 0, byref(p)))
        return p.value

    def stream_destroy(self, stream_ptr):
        check(self._lib, self._lib.fz_stream_destroy(self._h, c_void_p(stream_ptr)))

    # -- graph capture ---------------------------------------------------------------------------
    def graph_begin(self):
        """Start recording the device-pointer calls issued on this context (needs a non-default stream)."""
        check(self._lib, self._lib.fz_graph_begin(self._h))

    def graph_end(self):
        """-> Graph: the recorded sequence, replayable with .launch()"""
        g = c_void_p()
        check(self._lib, self._lib.fz_graph_end(self._h, byref(g)))
        return Graph(self, g)

    def runtime_info(self):
        """-> dict(build_hip_version, runtime_hip_version, arch): what the library was built with / is bound to"""
        b, r = c_int(), c_int()
        arch = ctypes.create_string_buffer(64)
        check(self._lib, self._lib.fz_runtime_info(self._h, byref(b), byref(r), arch, 64))
        return dict(build_hip_version=b.value, runtime_hip_version=r.value, arch=arch.value.decode())

    def twiddles(self):
        f = np.empty(self.degree, dtype=np.uint32)
        i = np.empty(self.degree, dtype=np.uint32)
        U32P = ctypes.POINTER(c_uint32)
        check(self._lib, self._lib.fz_ctx_twiddles(self._h, f.ctypes.data_as(U32P), i.ctypes.data_as(U32P)))
        return f, i

    # -- device memory helpers ----------------------------------------------------------------
    def malloc(self, nbytes):
        p = c_void_p()
        check(self._lib, self._lib.fz_malloc(self._h, nbytes, byref(p)))
        return p.value

    def free(self, ptr):
        check(self._lib, self._lib.fz_free(self._h, c_void_p(ptr)))

    def pool_trim(self, keep_bytes=0):
        """hand the idle blocks fz_free kept for reuse back to the runtime (down to keep_bytes)"""
        check(self._lib, self._lib.fz_pool_trim(self._h, keep_bytes))

    def h2d(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        check(self._lib, self._lib.fz_memcpy_h2d(self._h, c_void_p(dptr), c_void_p(arr.ctypes.data), arr.nbytes))
        self.synchronize()   # the numpy buffer may be a temporary

    def d2h(self, arr, dptr):
        assert arr.flags["C_CONTIGUOUS"]
        check(self._lib, self._lib.fz_memcpy_d2h(self._h, c_void_p(arr.ctypes.data), c_void_p(dptr), arr.nbytes))
        return arr

    def timer_start(self):
        check(self._lib, self._lib.fz_timer_start(self._h))

    def timer_stop_ms(self):
        ms = c_float()
        check(self._lib, self._lib.fz_timer_stop_ms(self._h, byref(ms)))
        return ms.value

    def profile_begin(self, max_launches, sample_every=1):
        check(self._lib, self._lib.fz_profile_begin(self._h, max_launches, sample_every))

    def profile_end(self):
        """-> dict(fwd_avg_us, fwd_count, inv_avg_us, inv_count): kernel begin->end per dispatch"""
        fa, ia = ctypes.c_double(), ctypes.c_double()
        fc, ic = c_int(), c_int()
        check(self._lib, self._lib.fz_profile_end(self._h, byref(fa), byref(fc), byref(ia), byref(ic)))
        return dict(fwd_avg_us=fa.value, fwd_count=fc.value, inv_avg_us=ia.value, inv_count=ic.value)

    def profile_end_samples(self, cap):
        """-> (us [n] float64, kind [n] int32: 0 forward / 1 inverse): every instrumented launch since profile_begin"""
        us, kind, n = np.empty(cap, dtype=np.float64), np.empty(cap, dtype=np.int32), c_int()
        check(self._lib, self._lib.fz_profile_end_samples(self._h, us.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                                          kind.ctypes.data_as(ctypes.POINTER(c_int)), cap, byref(n)))
        return us[:n.value], kind[:n.value]

    # -- device face (raw pointers) -------------------------------------------------------------
    def ntt_forward_dev(self, d_in, d_out, batch):
        check(self._lib, self._lib.fz_ntt_forward(self._h, c_void_p(d_in), c_void_p(d_out), batch))

    def ntt_inverse_dev(self, d_in, d_out, batch):
        check(self._lib, self._lib.fz_ntt_inverse(self._h, c_void_p(d_in), c_void_p(d_out), batch))

    def ntt_multi_dev(self, jobs):
        """ONE dispatch over independent transform jobs: jobs = [(d_in, d_out, rows, inverse), ...]"""
        arr = (NttJob * len(jobs))(*[NttJob(c_void_p(i), c_void_p(o), int(r), 1 if inv else 0) for i, o, r, inv in jobs])
        check(self._lib, self._lib.fz_ntt_multi(self._h, arr, len(jobs)))

    def diag_empty_launch(self):
        check(self._lib, self._lib.fz_diag_empty_launch(self._h))

    def diag_copy_dev(self, d_src, d_dst, nbytes):
        check(self._lib, self._lib.fz_diag_copy(self._h, c_void_p(d_src), c_void_p(d_dst), nbytes))

    def diag_shader_clock(self, microseconds=500):
        """MHz the shader clock holds while the work already queued on this context's stream runs (a one-wave probe on a
        private stream; returns after `microseconds`)"""
        mhz = c_double()
        check(self._lib, self._lib.fz_diag_shader_clock(self._h, int(microseconds), byref(mhz)))
        return mhz.value

    def diag_ntt_schedule(self, rows):
        """4 | 16 | 0: the transform schedule a launch of `rows` rows in all takes on this context (radix-4 wave-tasks, 16
        coefficients per lane, another kernel)"""
        fam = c_int(0)
        check(self._lib, self._lib.fz_diag_ntt_schedule(self._h, int(rows), byref(fam)))
        return fam.value

    def diag_delay(self, microseconds):
        """one wave that occupies this context's stream for `microseconds` (asynchronous, capturable)"""
        check(self._lib, self._lib.fz_diag_delay(self._h, int(microseconds)))

    def diag_stamps_begin(self, max_launches, max_workgroups):
        """device-side launch timestamps of fz_ntt_multi (include/fusion_hip_diag.h): slots for launches issued or captured from now on"""
        check(self._lib, self._lib.fz_diag_stamps_begin(self._h, int(max_launches), int(max_workgroups)))

    def diag_stamps_stop(self):
        check(self._lib, self._lib.fz_diag_stamps_stop(self._h))

    def diag_stamps_reset(self):
        check(self._lib, self._lib.fz_diag_stamps_reset(self._h))

    def diag_stamps_read(self, cap):
        """-> (start, end, last_start, workgroups): uint64 ticks of the chip's 100 MHz reference counter per recorded launch"""
        st, en, ls = (np.zeros(cap, dtype=np.uint64) for _ in range(3))
        wg = np.zeros(cap, dtype=np.uint32)
        n = ctypes.c_size_t()
        u64p, u32p = ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)
        check(self._lib, self._lib.fz_diag_stamps_read(self._h, st.ctypes.data_as(u64p), en.ctypes.data_as(u64p), ls.ctypes.data_as(u64p),
                                                       wg.ctypes.data_as(u32p), cap, byref(n)))
        return st[:n.value], en[:n.value], ls[:n.value], wg[:n.value]

    def reduce_scatter_i64_dev(self, comm, d_buf, count_per_rank):
        """in-place ncclReduceScatter(int64, sum) on this context's stream: block `rank` of the summed buffer arrives in block
        `rank` of d_buf (nranks blocks of count_per_rank elements); the other blocks are undefined afterwards"""
        check(self._lib, self._lib.fz_reduce_scatter_i64(self._h, comm._c, c_void_p(d_buf), count_per_rank))

    def allreduce_i64_dev(self, comm, d_buf, count):
        """in-place ncclAllReduce(int64, sum) on this context's stream (comm: a Comm)"""
        check(self._lib, self._lib.fz_allreduce_i64(self._h, comm._c, c_void_p(d_buf), count))

    def broadcast_i32_dev(self, comm, d_buf, count, root=0):
        """rank root's int32 buffer to every rank's (ncclBroadcast through the C ABI, in place, on the context's stream)"""
        check(self._lib, self._lib.fz_broadcast_i32(self._h, comm._c, c_void_p(d_buf), count, root))

    def challenge_dev(self, P, d_vk, prehash, N, d_out, transform=True):
        """hash_ch on the device for N (key, message) pairs: P a SchemeParams, d_vk [N][2][degree] (device), prehash
        [N][32] uint8 (host, hostpipe.hash_messages), d_out [N][degree] (device): c_hat, or the coefficient rows
        with transform=False"""
        pre = np.ascontiguousarray(prehash, dtype=np.uint8).reshape(-1, 32)
        if pre.shape[0] != N:
            raise FusionHipError(FZ_E_BADARG, f"{pre.shape[0]} pre-hashed messages for {N} keys")
        fn = self._lib.fz_challenge_hat_dev if transform else self._lib.fz_challenge_coefficients_dev
        check(self._lib, fn(self._h, byref(P), c_void_p(d_vk), pre.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), N,
                            c_void_p(d_out)))

    def sample_secret_polys_dev(self, seeds, modulus, degree, norm_bound, weight_bound, d_out):
        """the two secret polynomials of keygen(params, seed) for every seed, sampled ON THE DEVICE by an exact MT19937
        (fz_sample_secret_polys_dev): d_out [N][2][degree] int32 (device)"""
        sd = np.ascontiguousarray(seeds, dtype=np.uint64)
        check(self._lib, self._lib.fz_sample_secret_polys_dev(self._h, sd.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), sd.size,
                                                              modulus, degree, norm_bound, weight_bound, c_void_p(d_out)))

    def challenge_msgs_dev(self, P, d_vk, blob, offsets, N, d_out, want_prehash=False):
        """hash_ch on the device INCLUDING hash_message_to_int (fz_challenge_hat_msgs_dev): blob = the N messages' bytes back
        to back, offsets [N + 1] uintp (hostpipe._pack_messages); -> the [N][32] digests if want_prehash else None"""
        off = np.ascontiguousarray(offsets, dtype=np.uintp)
        if off.shape[0] != N + 1:
            raise FusionHipError(FZ_E_BADARG, f"{off.shape[0]} offsets for {N} messages")
        pre = np.empty((N, 32), dtype=np.uint8) if want_prehash else None
        u8p = ctypes.POINTER(ctypes.c_uint8)
        check(self._lib, self._lib.fz_challenge_hat_msgs_dev(
            self._h, byref(P), c_void_p(d_vk), blob, off.ctypes.data_as(ctypes.POINTER(ctypes.c_size_t)), N, c_void_p(d_out),
            pre.ctypes.data_as(u8p) if want_prehash else ctypes.cast(None, u8p)))
        return pre

    def pw_dev(self, op, d_a, d_b, d_out, count):
        fn = {OP_MUL: self._lib.fz_pw_mul, OP_ADD: self._lib.fz_pw_add, OP_SUB: self._lib.fz_pw_sub}[op]
        check(self._lib, fn(self._h, c_void_p(d_a), c_void_p(d_b), c_void_p(d_out), count))

    def fill_synthetic_dev(self, d_out, count, seed):
        """d_out[0:count] = the seeded centred-uniform stream (same values as the tests' host generator)"""
        check(self._lib, self._lib.fz_fill_synthetic(self._h, c_void_p(d_out), count, seed))

    def poly_mul_dev(self, d_f, d_g, d_out, batch):
        """negacyclic products of `batch` coefficient-domain rows (device pointers; d_out may alias an input)"""
        check(self._lib, self._lib.fz_poly_mul(self._h, c_void_p(d_f), c_void_p(d_g), c_void_p(d_out), batch))

    def pw_neg_dev(self, d_a, d_out, count):
        check(self._lib, self._lib.fz_pw_neg(self._h, c_void_p(d_a), c_void_p(d_out), count))

    def pw_mulacc_dev(self, d_acc, d_a, d_b, count):
        check(self._lib, self._lib.fz_pw_mulacc(self._h, c_void_p(d_acc), c_void_p(d_a), c_void_p(d_b), count))

    def pw_mul_bcast_dev(self, d_a, d_s, d_out, rows):
        check(self._lib, self._lib.fz_pw_mul_bcast(self._h, c_void_p(d_a), c_void_p(d_s), c_void_p(d_out), rows))

    def matvec_dev(self, d_A, d_S, d_out, batch, l):
        check(self._lib, self._lib.fz_matvec(self._h, c_void_p(d_A), c_void_p(d_S), c_void_p(d_out), batch, l))

    def keygen_core_dev(self, d_A, d_coef, d_sk_hat, d_vk, batch, l):
        check(self._lib, self._lib.fz_keygen_core(self._h, c_void_p(d_A), c_void_p(d_coef), c_void_p(d_sk_hat),
                                                  c_void_p(d_vk), batch, l))

    def keygen_core_bcast_dev(self, d_A, d_coef, d_sk_hat, d_vk, batch, l):
        """d_coef [batch][2][degree]: one secret polynomial per (key, half), used for all l rows"""
        check(self._lib, self._lib.fz_keygen_core_bcast(self._h, c_void_p(d_A), c_void_p(d_coef), c_void_p(d_sk_hat),
                                                        c_void_p(d_vk), batch, l))

    def sign_core_dev(self, d_sk_hat, d_c_hat, d_sig, batch, l):
        check(self._lib, self._lib.fz_sign_core(self._h, c_void_p(d_sk_hat), c_void_p(d_c_hat), c_void_p(d_sig),
                                                batch, l))

    def aggregate_core_dev(self, d_sig, d_alpha, d_out, N, l):
        check(self._lib, self._lib.fz_aggregate_core(self._h, c_void_p(d_sig), c_void_p(d_alpha), c_void_p(d_out), N, l))

    def aggregate_partial_dev(self, d_sig, d_alpha, d_partial, N, l):
        check(self._lib, self._lib.fz_aggregate_partial(self._h, c_void_p(d_sig), c_void_p(d_alpha),
                                                        c_void_p(d_partial), N, l))

    def target_partial_dev(self, d_vkL, d_vkR, d_c, d_alpha, d_partial, N):
        check(self._lib, self._lib.fz_target_partial(self._h, c_void_p(d_vkL), c_void_p(d_vkR), c_void_p(d_c),
                                                     c_void_p(d_alpha), c_void_p(d_partial), N))

    def aggregate_partial_batch_dev(self, d_sig, d_alpha, d_partial, partial_stride, groups, N, l):
        check(self._lib, self._lib.fz_aggregate_partial_batch(self._h, c_void_p(d_sig), c_void_p(d_alpha),
                                                              c_void_p(d_partial), partial_stride, groups, N, l))

    def aggregate_target_partial_batch_dev(self, d_sig, d_alpha, d_vkL, d_vkR, d_c, d_partial, partial_stride,
                                           d_target_partial, target_stride, groups, N, l):
        """aggregate partials and the verification target's partials in one pass over the signers"""
        check(self._lib, self._lib.fz_aggregate_target_partial_batch(
            self._h, c_void_p(d_sig), c_void_p(d_alpha), c_void_p(d_vkL), c_void_p(d_vkR), c_void_p(d_c),
            c_void_p(d_partial), partial_stride, c_void_p(d_target_partial), target_stride, groups, N, l))

    def sign_aggregate_target_partial_batch_dev(self, d_sk_hat, d_c, d_alpha, d_vkL, d_vkR, d_sig, d_partial, partial_stride,
                                                d_target_partial, target_stride, groups, N, l):
        """sign_core and aggregate_target_partial_batch in ONE pass: sigma is written to d_sig as it is computed and aggregated
        from registers (d_vkL / d_vkR / d_target_partial: together, or all 0 for the aggregate's sums alone)"""
        check(self._lib, self._lib.fz_sign_aggregate_target_partial_batch(
            self._h, c_void_p(d_sk_hat), c_void_p(d_c), c_void_p(d_alpha), c_void_p(d_vkL or None), c_void_p(d_vkR or None),
            c_void_p(d_sig), c_void_p(d_partial), partial_stride, c_void_p(d_target_partial or None), target_stride, groups, N, l))

    def aggregate_core_ragged_dev(self, d_sig, d_alpha, offsets, l, d_out):
        """len(offsets) - 1 aggregates of different sizes in one launch: aggregate g = rows [offsets[g], offsets[g+1]) of the
        concatenated signatures / coefficients; d_out [groups][l][degree] int32 (centred)"""
        off = (ctypes.c_size_t * len(offsets))(*[int(x) for x in offsets])
        check(self._lib, self._lib.fz_aggregate_core_ragged(self._h, c_void_p(d_sig), c_void_p(d_alpha), off, len(offsets) - 1, l,
                                                            c_void_p(d_out)))

    def aggregate_target_partial_ragged_dev(self, d_sig, d_alpha, d_vkL, d_vkR, d_c, offsets, l, d_partial, partial_stride,
                                            d_target_partial, target_stride):
        """int64 partial sums of aggregates (d_sig / d_partial may both be 0: targets only) and verification targets for
        aggregates of different sizes, one launch"""
        off = (ctypes.c_size_t * len(offsets))(*[int(x) for x in offsets])
        check(self._lib, self._lib.fz_aggregate_target_partial_ragged(
            self._h, c_void_p(d_sig or None), c_void_p(d_alpha), c_void_p(d_vkL), c_void_p(d_vkR), c_void_p(d_c), off,
            len(offsets) - 1, l, c_void_p(d_partial or None), partial_stride, c_void_p(d_target_partial), target_stride))

    def verify_partials_batch_async_dev(self, d_A, d_partial, partial_stride, d_target_partial, target_stride, groups, l,
                                        beta_vf, omega_vf, d_verdicts):
        """verdict codes straight from int64 partial sums (asynchronous, verdicts stay on the device)"""
        check(self._lib, self._lib.fz_verify_partials_batch_async(
            self._h, c_void_p(d_A), c_void_p(d_partial), partial_stride, c_void_p(d_target_partial), target_stride,
            groups, l, beta_vf, omega_vf, c_void_p(d_verdicts)))

    def target_partial_batch_dev(self, d_vkL, d_vkR, d_c, d_alpha, d_partial, partial_stride, groups, N):
        check(self._lib, self._lib.fz_target_partial_batch(self._h, c_void_p(d_vkL), c_void_p(d_vkR), c_void_p(d_c),
                                                           c_void_p(d_alpha), c_void_p(d_partial), partial_stride,
                                                           groups, N))

    def verify_with_target_batch_dev(self, d_A, d_sig, d_target, groups, l, beta_vf, omega_vf):
        v = (c_int * groups)()
        check(self._lib, self._lib.fz_verify_with_target_batch(self._h, c_void_p(d_A), c_void_p(d_sig),
                                                               c_void_p(d_target), groups, l, beta_vf, omega_vf, v))
        return list(v)

    def verify_with_target_batch_async_dev(self, d_A, d_sig, d_target, groups, l, beta_vf, omega_vf, d_verdicts):
        check(self._lib, self._lib.fz_verify_with_target_batch_async(self._h, c_void_p(d_A), c_void_p(d_sig),
                                                                     c_void_p(d_target), groups, l, beta_vf,
                                                                     omega_vf, c_void_p(d_verdicts)))

    def reduce_i64_dev(self, d_in, d_out, count):
        check(self._lib, self._lib.fz_reduce_i64(self._h, c_void_p(d_in), c_void_p(d_out), count))

    def verify_core_dev(self, d_A, d_sig, d_vkL, d_vkR, d_c, d_alpha, N, l, beta_vf, omega_vf):
        v = c_int(-1)
        check(self._lib, self._lib.fz_verify_core(self._h, c_void_p(d_A), c_void_p(d_sig), c_void_p(d_vkL),
                                                  c_void_p(d_vkR), c_void_p(d_c), c_void_p(d_alpha), N, l,
                                                  beta_vf, omega_vf, byref(v)))
        return v.value

    def verify_with_target_dev(self, d_A, d_sig, d_target, l, beta_vf, omega_vf):
        v = c_int(-1)
        check(self._lib, self._lib.fz_verify_with_target(self._h, c_void_p(d_A), c_void_p(d_sig), c_void_p(d_target),
                                                         l, beta_vf, omega_vf, byref(v)))
        return v.value

    def norm_weight_dev(self, d_coef, batch, d_max_abs, d_weight):
        check(self._lib, self._lib.fz_norm_weight(self._h, c_void_p(d_coef), batch, c_void_p(d_max_abs),
                                                  c_void_p(d_weight)))

    # -- host face (numpy in, numpy out) --------------------------------------------------------
    def _rows(self, a):
        a = _as_i32(a)
        if a.size % self.degree:
            raise FusionHipError(FZ_E_BADARG, f"array of {a.size} values is not a whole number of degree-{self.degree} rows")
        return a, a.size // self.degree

    def ntt_forward(self, x):
        """cooley_tukey_ntt of every row; returns a new array."""
        a, rows = self._rows(x)
        out = a.copy()
        check(self._lib, self._lib.fz_ntt_forward_host(self._h, _p(out), rows))
        return out

    def ntt_inverse(self, x):
        """gentleman_sande_intt of every row; returns a new array."""
        a, rows = self._rows(x)
        out = a.copy()
        check(self._lib, self._lib.fz_ntt_inverse_host(self._h, _p(out), rows))
        return out

    def _pw(self, op, a, b):
        a = _as_i32(a)
        out = np.empty_like(a)
        if op == OP_NEG:
            check(self._lib, self._lib.fz_pw_binary_host(self._h, op, _p(a), None, _p(out), a.size))
            return out
        b = _as_i32(b)
        if a.shape != b.shape:
            raise FusionHipError(FZ_E_BADARG, f"shape mismatch {a.shape} vs {b.shape}")
        check(self._lib, self._lib.fz_pw_binary_host(self._h, op, _p(a), _p(b), _p(out), a.size))
        return out

    def pw_mul(self, a, b):
        return self._pw(OP_MUL, a, b)

    def poly_mul(self, f, g):
        """INTT(NTT(f) * NTT(g)) for host rows [batch][degree] (or one row): ntt_poly_mult, ntt.py:380-484"""
        f, g = _as_i32(f), _as_i32(g)
        if f.shape != g.shape or f.shape[-1] != self.degree:
            raise FusionHipError(FZ_E_BADARG, f"shape mismatch {f.shape} vs {g.shape} (degree {self.degree})")
        out = np.empty_like(f)
        check(self._lib, self._lib.fz_poly_mul_host(self._h, _p(f), _p(g), _p(out), f.size // self.degree))
        return out

    def pw_add(self, a, b):
        return self._pw(OP_ADD, a, b)

    def pw_sub(self, a, b):
        return self._pw(OP_SUB, a, b)

    def pw_neg(self, a):
        return self._pw(OP_NEG, a, None)

    def matvec(self, A, S):
        """A: [l][d]; S: [batch][l][d] -> [batch][d]."""
        A = _as_i32(A)
        S = _as_i32(S)
        l = A.shape[0]
        S3 = S.reshape(-1, l, self.degree)
        out = np.empty((S3.shape[0], self.degree), dtype=np.int32)
        check(self._lib, self._lib.fz_matvec_host(self._h, _p(A), _p(S3), _p(out), S3.shape[0], l))
        return out

    def norm_weight(self, coef):
        a, rows = self._rows(coef)
        mx = np.empty(rows, dtype=np.int64)
        wt = np.empty(rows, dtype=np.int32)
        check(self._lib, self._lib.fz_norm_weight_host(self._h, _p(a), rows, mx.ctypes.data_as(_I64P), _p(wt)))
        return mx, wt


    # -- host face of the fused scheme cores (object API of fusion.fusion) -----------------------
    def _staged(self, arrays):
        return [DeviceBuffer.from_numpy(self, _as_i32(a)) for a in arrays]

    def keygen_core(self, A, coef):
        """A [l][d]; coef [batch][2][l][d] -> (sk_hat [batch][2][l][d], vk [batch][2][d])."""
        A, coef = _as_i32(A), _as_i32(coef)
        l, d = A.shape
        c4 = coef.reshape(-1, 2, l, d)
        batch = c4.shape[0]
        dA, dC = self._staged([A, c4])
        dS = DeviceBuffer(self, c4.nbytes)
        dV = DeviceBuffer(self, batch * 2 * d * 4)
        try:
            self.keygen_core_dev(dA.ptr, dC.ptr, dS.ptr, dV.ptr, batch, l)
            return dS.to_numpy(np.int32, c4.shape), dV.to_numpy(np.int32, (batch, 2, d))
        finally:
            for b in (dA, dC, dS, dV):
                b.free()

    def keygen_core_bcast(self, A, polys):
        """A [l][d]; polys [batch][2][d]: ONE secret polynomial per (key, half), standing for all l rows of that half
        (every entry of a seeded half is the same polynomial: fusion.py:156-173) -> (sk_hat [batch][2][l][d], vk [batch][2][d])."""
        A, polys = _as_i32(A), _as_i32(polys)
        l, d = A.shape
        p3 = polys.reshape(-1, 2, d)
        batch = p3.shape[0]
        dA, dC = self._staged([A, p3])
        dS = DeviceBuffer(self, batch * 2 * l * d * 4)
        dV = DeviceBuffer(self, batch * 2 * d * 4)
        try:
            self.keygen_core_bcast_dev(dA.ptr, dC.ptr, dS.ptr, dV.ptr, batch, l)
            return dS.to_numpy(np.int32, (batch, 2, l, d)), dV.to_numpy(np.int32, (batch, 2, d))
        finally:
            for b in (dA, dC, dS, dV):
                b.free()

    def sign_core(self, sk_hat, c_hat):
        """sk_hat [batch][2][l][d]; c_hat [batch][d] -> sig [batch][l][d]."""
        sk, c = _as_i32(sk_hat), _as_i32(c_hat)
        batch, _, l, d = sk.shape
        dK, dC = self._staged([sk, c.reshape(batch, d)])
        dS = DeviceBuffer(self, batch * l * d * 4)
        try:
            self.sign_core_dev(dK.ptr, dC.ptr, dS.ptr, batch, l)
            return dS.to_numpy(np.int32, (batch, l, d))
        finally:
            for b in (dK, dC, dS):
                b.free()

    def aggregate_core(self, sig, alpha_hat):
        """sig [N][l][d]; alpha_hat [N][d] -> [l][d]."""
        sig, al = _as_i32(sig), _as_i32(alpha_hat)
        N, l, d = sig.shape
        dS, dA = self._staged([sig, al.reshape(N, d)])
        dO = DeviceBuffer(self, l * d * 4)
        try:
            self.aggregate_core_dev(dS.ptr, dA.ptr, dO.ptr, N, l)
            return dO.to_numpy(np.int32, (l, d))
        finally:
            for b in (dS, dA, dO):
                b.free()

    def verify_core(self, A, sig, vkL, vkR, c_hat, alpha_hat, beta_vf, omega_vf):
        """-> verdict code (VERDICT_REASONS)."""
        A, sig = _as_i32(A), _as_i32(sig)
        l, d = A.shape
        vkL, vkR, c_hat, alpha_hat = (_as_i32(x).reshape(-1, d) for x in (vkL, vkR, c_hat, alpha_hat))
        N = vkL.shape[0]
        bufs = self._staged([A, sig.reshape(l, d), vkL, vkR, c_hat, alpha_hat])
        try:
            return self.verify_core_dev(*(b.ptr for b in bufs), N, l, int(beta_vf), int(omega_vf))
        finally:
            for b in bufs:
                b.free()


def comm_unique_id():
    """-> 128 bytes (ncclUniqueId) that rank 0 hands to the other ranks"""
    lib = load_library()
    uid = UniqueId()
    check(lib, lib.fz_comm_unique_id(byref(uid)))
    return ctypes.string_at(ctypes.addressof(uid), 128)


def rccl_version():
    """-> the version code of the RCCL the library bound (ncclGetVersion), e.g. 22703"""
    lib = load_library()
    v = c_int()
    check(lib, lib.fz_rccl_version(byref(v)))
    return v.value


def rccl_library():
    """-> dict(path, how, copies_mapped): which RCCL file serves fz_comm_* in this process, by which rule it was found
    ("already mapped (shared)" | "beside the HIP runtime" | "default search path") and how many different librccl files the
    process has mapped (2 = two RCCLs in one process).  Binds RCCL if nothing has yet."""
    lib = load_library()
    path, how, n = ctypes.create_string_buffer(512), ctypes.create_string_buffer(64), c_int()
    check(lib, lib.fz_rccl_library(path, 512, how, 64, byref(n)))
    return {"path": path.value.decode(), "how": how.value.decode(), "copies_mapped": n.value}


class Comm:
    """An RCCL communicator owned through the C ABI (fz_comm_*).  destroy() is idempotent (so is fz_comm_destroy itself)."""

    def __init__(self, ctx, nranks, rank, unique_id):
        self._lib = ctx._lib
        self._c = c_void_p()
        uid = UniqueId()
        ctypes.memmove(ctypes.addressof(uid), bytes(unique_id), 128)
        check(self._lib, self._lib.fz_comm_create(ctx._h, nranks, rank, byref(uid), byref(self._c)))

    def info(self):
        """-> (nranks as RCCL reports it, rank)"""
        n, r = c_int(), c_int()
        check(self._lib, self._lib.fz_comm_info(self._c, byref(n), byref(r)))
        return n.value, r.value

    def destroy(self):
        if self._c:
            self._lib.fz_comm_destroy(self._c)
            self._c = c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class Graph:
    """A captured sequence of device calls (fz_graph_*)."""

    def __init__(self, ctx, handle):
        self.ctx, self._g = ctx, handle

    def launch(self):
        check(self.ctx._lib, self.ctx._lib.fz_graph_launch(self.ctx._h, self._g))

    def destroy(self):
        if self._g:
            self.ctx._lib.fz_graph_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class Event:
    """A point on a context's stream that another context's stream can wait for (fz_event_*): record(ctx) marks what ctx's
    stream has been given so far, wait(ctx) makes ctx's stream wait for the marked point without blocking the host.  Usable
    inside a capture (the waiting context's stream joins the capture; the capturing context must wait for an event recorded
    on it again before graph_end)."""

    def __init__(self, ctx):
        self._lib, self._e = ctx._lib, c_void_p()
        check(self._lib, self._lib.fz_event_create(ctx._h, byref(self._e)))

    def record(self, ctx):
        check(self._lib, self._lib.fz_event_record(ctx._h, self._e))

    def wait(self, ctx):
        check(self._lib, self._lib.fz_event_wait(ctx._h, self._e))

    def destroy(self):
        if self._e:
            self._lib.fz_event_destroy(self._e)
            self._e = c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class DeviceBuffer:
    """Minimal owning wrapper over fz_malloc for callers that do not use torch."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, nbytes
        self.ptr = ctx.malloc(nbytes)

    @classmethod
    def from_numpy(cls, ctx, arr):
        arr = np.ascontiguousarray(arr)
        buf = cls(ctx, arr.nbytes)
        ctx.h2d(buf.ptr, arr)
        return buf

    def to_numpy(self, dtype, shape):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        return self.ctx.d2h(out, self.ptr)

    def free(self):
        if self.ptr:
            self.ctx.free(self.ptr)
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceArray:
    """An int32/int64 array that lives in device memory (shape + owning DeviceBuffer): the device-resident
    face of BatchScheme, so that keys and signatures do not cross PCIe between calls."""

    def __init__(self, ctx, shape, dtype=np.int32):
        self.ctx, self.shape, self.dtype = ctx, tuple(int(x) for x in shape), np.dtype(dtype)
        self.buf = DeviceBuffer(ctx, max(1, int(np.prod(self.shape)) * self.dtype.itemsize))

    @classmethod
    def from_numpy(cls, ctx, arr):
        arr = np.ascontiguousarray(arr)
        out = cls(ctx, arr.shape, arr.dtype)
        if arr.size:
            ctx.h2d(out.buf.ptr, arr)
        return out

    @property
    def ptr(self):
        return self.buf.ptr

    def numpy(self):
        out = np.empty(self.shape, dtype=self.dtype)
        return self.ctx.d2h(out, self.buf.ptr) if out.size else out

    def free(self):
        self.buf.free()


_CTX_CACHE = {}
# Contexts built from caller-supplied TABLES are memoised in a small LRU instead: every distinct table is a context with its
# own streams, events, device tables and pool, and a caller iterating over tables would otherwise keep all of them alive for the
# life of the process.  The evicted context is closed (the drop-in functions hold one for the duration of a call only).
_TABLE_CTX_MAX = 8
_TABLE_CTX_CACHE = collections.OrderedDict()


def get_table_context(modulus, degree, fwd_table, inv_table, device=0):
    """memoised contexts built from caller-supplied twiddle tables (tuples): the `_TABLE_CTX_MAX` most recently used"""
    key = (modulus, degree, fwd_table, inv_table, device)
    ctx = _TABLE_CTX_CACHE.get(key)
    if ctx is not None:
        _TABLE_CTX_CACHE.move_to_end(key)
        return ctx
    ctx = _TABLE_CTX_CACHE[key] = Context(modulus, degree, 0, 0, device, tables=(fwd_table, inv_table))
    while len(_TABLE_CTX_CACHE) > _TABLE_CTX_MAX:
        _, old = _TABLE_CTX_CACHE.popitem(last=False)
        old.close()
    return ctx


def get_context(modulus, degree, root, inv_root, device=0):
    """Memoised contexts: the drop-in object API creates one per parameter tuple."""
    key = (modulus, degree, root % modulus, inv_root % modulus, device)
    ctx = _CTX_CACHE.get(key)
    if ctx is None:
        ctx = _CTX_CACHE[key] = Context(modulus, degree, root, inv_root, device)
    return ctx
