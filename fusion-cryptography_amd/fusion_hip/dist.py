"""Multi-GPU sharding of the path: one process per GPU, torch.distributed (backend "nccl" is RCCL
on ROCm; "gloo" for the CPU rehearsal).

What shards and what is exchanged
  * NTT batches, keygen and sign are independent per polynomial / signature: ranks own contiguous
    blocks, no data-path collective.
  * aggregate() is a sum over signatures and verify()'s target a sum over signers: each rank sums
    its block exactly into int64 (fz_aggregate_partial / fz_target_partial), then ONE all-reduce
    (ncclSum on int64; sums of centred int32 products need > 32 bits) and a centring kernel
    (fz_reduce_i64).  Integer sums are associative, so the result is bit-identical for any number
    of ranks and any reduction order.

The functions take the per-rank partial computation as a callable so that the same code is
exercised on CPU (gloo, partials from the oracle -- tests/test_dist_cpu.py) and on GPUs (partials
from the HIP kernels -- bench.py).
"""
from typing import Callable, Tuple


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of `total` items owned by `rank`: the first total % world ranks
    get one extra item.  Empty blocks are allowed (total < world)."""
    if world < 1 or not (0 <= rank < world) or total < 0:
        raise ValueError(f"bad shard request total={total} rank={rank} world={world}")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def allreduce_sum_i64(t, group=None):
    """In-place sum of an int64 tensor over the ranks of `group`; a no-op without an initialised
    process group (single-GPU runs)."""
    import torch
    import torch.distributed as dist
    if t.dtype != torch.int64:
        raise TypeError("partial sums must be int64: 8 centred int32 values already overflow int32")
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def sharded_sum(partial_fn: Callable[[int, int], "object"], total: int, rank: int, world: int, group=None):
    """partial_fn(lo, hi) -> int64 tensor holding this rank's exact partial sums over items
    [lo, hi) (zeros for an empty block); returns the all-reduced tensor."""
    lo, hi = shard_range(total, rank, world)
    return allreduce_sum_i64(partial_fn(lo, hi), group)


def sharded_alpha(rank: int, world: int, mode: str, coll, n: int, d: int, compute):
    """hash_ag's aggregation-coefficient rows [n][d] (callers' order) on this rank.  `compute()` runs the global sort by
    str(vk) and the ONE serial SHAKE-256 over the whole sorted list (fusion.py:586-591, :632-652, :661-663) and returns them.
      mode "replicated": every rank calls compute() -- no exchange, every rank pays the sponge;
      mode "root":       rank 0 alone calls compute() and broadcasts the rows (n * d int32: 1 MiB at 1024 signers) while the
                         other ranks do their own block's work; they join the broadcast when they need the rows.
    Pure host logic (tests/test_dist_cpu.py runs it at world 8 over gloo); the wall time is the sponge's in both modes --
    it is serial by construction -- but in "root" mode it costs ONE core of the node instead of one per rank."""
    if mode not in ("replicated", "root"):
        raise ValueError(f"alpha mode {mode!r}: 'replicated' or 'root'")
    if mode == "replicated" or world == 1:
        return compute()
    if rank == 0:
        alpha = compute()
        coll.broadcast_i32(alpha, (n, d), 0)
        return alpha
    return coll.broadcast_i32(None, (n, d), 0)


def resolve_alpha_mode(mode: str, world: int) -> str:
    """"auto": the sponge on rank 0 alone from four ranks on (it then occupies one core of the node instead of world cores;
    measured over gloo, profiles/r04_sharded_alpha_modes.txt), on every rank below (no broadcast in the path)."""
    if mode == "auto":
        return "root" if world >= 4 else "replicated"
    return mode


# ---------------------------------------------------------------------------------------------------------------
# End-to-end sharded aggregate() / verify() (BASELINE configs[3]; fusion/fusion.py:655-677, :680-728)
# ---------------------------------------------------------------------------------------------------------------
class TorchCollective:
    """The exchange step over torch.distributed (backend "nccl" = RCCL over xGMI with one GPU per rank; "gloo" when ranks
    share a GPU, as in the tests).  Buffers are torch tensors; the library's kernels read and write them by pointer.
    device=None: host tensors only (the CPU rehearsal of the host logic: broadcast_i32)."""
    name = "torch.distributed all_reduce"

    def __init__(self, ctx, device, group=None):
        import torch
        self.ctx, self.group = ctx, group
        self.device = torch.device("cuda", device) if device is not None else None

    def broadcast_i32(self, arr, shape, root=0):
        """rank `root` passes the int32 array, the others None: -> the array on every rank (host memory)"""
        import numpy as np
        import torch
        import torch.distributed as dist
        on_dev = self.device is not None and dist.get_backend(self.group) == "nccl"
        if arr is not None:
            t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.int32).reshape(shape))
            t = t.to(self.device) if on_dev else t
        else:
            t = torch.empty(shape, dtype=torch.int32, device=self.device if on_dev else "cpu")
        dist.broadcast(t, src=root, group=self.group)
        return t.cpu().numpy() if arr is None else np.asarray(arr, dtype=np.int32).reshape(shape)

    def alloc_i64(self, count):
        import torch
        buf = torch.zeros(count, dtype=torch.int64, device=self.device)
        if self.device is None:
            return buf                                   # host tensors (the CPU rehearsal): nothing to order
        # the fill runs on torch's current stream, the kernels that add into the buffer on the context's -- which need not be
        # the same stream, and a hipStreamNonBlocking one (BatchScheme(private_context=True)) is not even ordered against the
        # legacy default stream: the zeros must have landed before the pointer is handed to the library
        torch.cuda.current_stream(self.device).synchronize()
        return buf

    @staticmethod
    def ptr(buf):
        return buf.data_ptr()

    def allreduce(self, buf):
        import torch
        if self.device is None:
            allreduce_sum_i64(buf, self.group)
            return
        self.ctx.synchronize()                 # the library's stream and torch's need not be the same stream
        allreduce_sum_i64(buf, self.group)
        torch.cuda.synchronize(self.device)

    @staticmethod
    def to_numpy(buf):
        return buf.cpu().numpy()

    def free(self, buf):
        pass


class CommCollective:
    """The exchange step through the C ABI: fz_allreduce_i64 = ncclAllReduce(int64, sum) on the context's own stream, ordered
    with the kernels around it (one GPU per rank; `comm` a fusion_hip.Comm)."""
    name = "fz_allreduce_i64 (RCCL through the C ABI)"

    def __init__(self, ctx, comm):
        self.ctx, self.comm = ctx, comm

    def alloc_i64(self, count):
        from .context import DeviceArray
        import numpy as np
        return DeviceArray.from_numpy(self.ctx, np.zeros(count, dtype=np.int64))

    @staticmethod
    def ptr(buf):
        return buf.ptr

    def allreduce(self, buf):
        self.ctx.allreduce_i64_dev(self.comm, buf.ptr, buf.shape[0])

    def broadcast_i32(self, arr, shape, root=0):
        """ncclBroadcast through the C ABI (fz_broadcast_i32) on the context's stream; host array in, host array out"""
        from .context import DeviceArray
        import numpy as np
        if arr is not None:
            buf = DeviceArray.from_numpy(self.ctx, np.ascontiguousarray(arr, dtype=np.int32).reshape(shape))
        else:
            buf = DeviceArray(self.ctx, shape)
        try:
            self.ctx.broadcast_i32_dev(self.comm, buf.ptr, int(np.prod(shape)), root)
            return buf.numpy() if arr is None else np.asarray(arr, dtype=np.int32).reshape(shape)
        finally:
            buf.free()

    @staticmethod
    def to_numpy(buf):
        return buf.numpy()

    def free(self, buf):
        buf.free()


class LocalCollective(CommCollective):
    """No exchange: one rank holds every signer (world size 1).  Lets the one-pass form -- partial sums of aggregate and
    verification target in ONE pass over the signatures, verification straight from the sums, ONE hash_ag for both --
    serve a single GPU as well (BatchScheme.aggregate_verify)."""
    name = "none (single rank)"

    def __init__(self, ctx):
        self.ctx, self.comm = ctx, None

    def allreduce(self, buf):
        pass

    def broadcast_i32(self, arr, shape, root=0):
        return arr


class HipSteps:
    """The DEVICE steps of the sharded flow on libfusion_hip.so -- the product path, and ShardedScheme's default.  Everything
    ShardedScheme does to data goes through this interface (buffers are opaque to it), so that its own control flow -- blocks,
    the two alpha modes, the offset of a rank's challenges, ranks without signers, the verdicts that need no arithmetic -- can
    be driven at world 8 on a machine without a GPU by a stand-in with the same methods (tests/_shard_host_worker.py:
    the C oracle behind them, which is what the GPU tests check these kernels against).  There is no CPU implementation in
    the product: a ShardedScheme built without `steps` uses this class and fails loudly without a device."""

    def __init__(self, scheme):
        self.bs, self.ctx, self.params, self.d, self.l = scheme, scheme.ctx, scheme.params, scheme.d, scheme.l

    # -- host-side hashing (hostpipe / the device challenge pipeline), as BatchScheme runs it
    def split_vk(self, vk_all):
        return self.bs._split_vk(vk_all)

    def challenges(self, vk, messages):
        """hash_ch of every (key, message): -> (rows on the device [m][d], the same rows on the host, prehash [m][32])"""
        return self.bs._challenges_both(vk, messages)

    def alpha_rows(self, L, R, pre, c_hat):
        """hash_ag without the transforms: sort by str(vk), ONE serial SHAKE-256, decode -> rows [n][d] in the callers' order"""
        return self.bs._alpha_coefficients(L, R, pre, c_hat)[1]

    # -- buffers
    def rows(self, arr):
        import numpy as np
        from .context import DeviceArray
        return DeviceArray.from_numpy(self.ctx, np.ascontiguousarray(arr))

    def empty_rows(self, shape):
        from .context import DeviceArray
        return DeviceArray(self.ctx, shape)

    def take(self, a, shape):
        """the caller's rows (numpy or device) as a device buffer -> (buffer, whether this call owns it)"""
        return self.bs._dev(a, shape)

    def free(self, *bufs):
        for b in bufs:
            b.free()

    def synchronize(self):
        self.ctx.synchronize()

    # -- arithmetic
    def transform_rows(self, buf, m):
        self.ctx.ntt_forward_dev(buf.ptr, buf.ptr, m)                     # in place (fusion.py:614-624: alpha -> alpha_hat)

    def partial_sums(self, sig, al, L, R, c, c_row0, coll, part, m):
        """exact int64 partial sums of the aggregate [l][d] and of the verification target [d] over m signers, ONE pass"""
        l, d, pp = self.l, self.d, coll.ptr(part)
        self.ctx.aggregate_target_partial_batch_dev(sig.ptr, al.ptr, L.ptr, R.ptr, c.ptr + c_row0 * d * 4, pp, l * d, pp + l * d * 8, d, 1, m, l)

    def target_partial(self, L, R, c, c_row0, al, coll, part, m):
        self.ctx.target_partial_dev(L.ptr, R.ptr, c.ptr + c_row0 * self.d * 4, al.ptr, coll.ptr(part), m)

    def centred(self, coll, part, count):
        """the first `count` sums, centred: -> int32 array on the host"""
        from .context import DeviceArray
        out = DeviceArray(self.ctx, (count,))
        try:
            self.ctx.reduce_i64_dev(coll.ptr(part), out.ptr, count)
            return out.numpy()
        finally:
            out.free()

    def verdict_from_sums(self, coll, part):
        """verify() straight from the int64 sums (aggregate [l][d], then target [d]) -> verdict code"""
        from .context import DeviceArray
        l, d, pp = self.l, self.d, coll.ptr(part)
        dV = DeviceArray(self.ctx, (1,))
        try:
            self.ctx.verify_partials_batch_async_dev(self.bs._A_dev().ptr, pp, l * d, pp + l * d * 8, d, 1, l, int(self.params.beta_vf),
                                                     int(self.params.omega_vf), dV.ptr)
            return int(dV.numpy()[0])
        finally:
            dV.free()

    def verdict_with_target(self, agg, target):
        """verify() of an aggregate [l][d] (device buffer) against a centred target [d] (host) -> verdict code"""
        dT = self.rows(target)
        try:
            return self.ctx.verify_with_target_dev(self.bs._A_dev().ptr, agg.ptr, dT.ptr, self.l, int(self.params.beta_vf), int(self.params.omega_vf))
        finally:
            dT.free()


class ShardedScheme:
    """aggregate() and verify() of the reference with the SIGNERS sharded over ranks -- one process per GPU, rank r holds
    the signatures of its contiguous block [lo, hi) of the callers' list (shard_range) in its GPU's memory.

    Who computes what, and why:
      * Verification keys and messages are public and small (2 KiB per signer); every rank has all of them.
      * The global sort by str(vk) (fusion.py:661-663, :693) and hash_ag's ONE serial SHAKE-256 over the whole sorted list
        (fusion.py:586-591, :632-652) are serial by construction and bound the wall time wherever they run (alpha_mode):
        "replicated" -- every rank runs them (bit-identical alpha without any exchange; world cores busy with the same
        sponge); "root" -- rank 0 alone runs them and broadcasts the coefficient rows (1 MiB at 1024 signers) while the other
        ranks compute only THEIR block's challenges and upload their rows; "auto" = root from four ranks on (sharded_alpha).
        Neither form scales with the rank count: only the algebra after it does.
      * alpha is scattered back to the callers' order on the host (a permutation of 1 KiB rows) and each rank uploads and
        transforms ONLY its block.
      * Each rank makes ONE pass over its signers (fz_aggregate_target_partial_batch): exact int64 partial sums of the
        aggregate [l][d] and of the verification target [d].
      * ONE all-reduce of l*d + d int64 (sums of centred products need more than 32 bits) -- the path's only exchange.
      * Every rank then holds the complete sums: the aggregate is their centring (fz_reduce_i64), the verdict comes straight
        from the sums (fz_verify_partials_batch_async).  Integer sums are associative: bit-identical for any world size.
    `steps`: the device steps (default HipSteps(scheme): libfusion_hip).  The class itself holds the control flow only.
    """

    def __init__(self, scheme, rank, world, collective, alpha_mode="auto", steps=None):
        self.bs, self.rank, self.world, self.coll = scheme, int(rank), int(world), collective
        self.steps = steps if steps is not None else HipSteps(scheme)
        self.params, self.d, self.l = self.steps.params, self.steps.d, self.steps.l
        self.alpha_mode = resolve_alpha_mode(alpha_mode, self.world)

    def block(self, n):
        return shard_range(n, self.rank, self.world)

    def _local_operands(self, vk_all, messages_all):
        """-> (n, lo, hi, dC, c_row0, dAl, dL, dR): challenges on the device (dC, this rank's block starting at row c_row0),
        alpha_hat / vk rows of this rank's block"""
        st = self.steps
        vk, L, R = st.split_vk(vk_all)
        n = vk.shape[0]
        if n != len(messages_all):
            raise ValueError("Number of keys and messages must be equal.")
        lo, hi = self.block(n)
        m = hi - lo
        everything = self.alpha_mode == "replicated" or self.world == 1 or self.rank == 0
        c_hat = pre = None
        if everything:                                   # this rank runs the sponge: it needs every signer's challenge
            dC, c_hat, pre = st.challenges(vk, messages_all)
            c_row0 = lo
        elif m:                                          # "root" mode, not the root: its own block's challenges only
            dC, _, _ = st.challenges(vk[lo:hi], messages_all[lo:hi])
            c_row0 = 0
        else:
            dC, c_row0 = st.empty_rows((1, self.d)), 0
        try:
            alpha = sharded_alpha(self.rank, self.world, self.alpha_mode, self.coll, n, self.d,
                                  (lambda: st.alpha_rows(L, R, pre, c_hat)) if everything else None)
            dAl = st.rows(alpha[lo:hi])
            if m:
                st.transform_rows(dAl, m)
            dL = st.rows(L[lo:hi])
            dR = st.rows(R[lo:hi])
        except Exception:
            st.free(dC)
            raise
        return n, lo, hi, dC, c_row0, dAl, dL, dR

    def aggregate_verify_sharded(self, vk_all, messages_all, sig_local):
        """-> (aggregate [l][d] int32, (ok, reason)) on EVERY rank.  sig_local: this rank's signatures [hi - lo][l][d]
        (numpy or DeviceArray), rows in the callers' order.  One pass over the local signers, one all-reduce, verification
        from the int64 sums."""
        from .context import VERDICT_REASONS
        st, l, d = self.steps, self.l, self.d
        n, lo, hi, dC, c_row0, dAl, dL, dR = self._local_operands(vk_all, messages_all)
        m = hi - lo
        dS, own = st.take(sig_local, (m, l, d))
        part = self.coll.alloc_i64(l * d + d)                # zeros: a rank without signers contributes nothing
        try:
            if m:
                st.partial_sums(dS, dAl, dL, dR, dC, c_row0, self.coll, part, m)
            self.coll.allreduce(part)                         # the ONE exchange step
            agg = st.centred(self.coll, part, l * d).reshape(l, d)
            if n > self.params.capacity:                      # fusion.py:686-687 (checked before anything else there)
                return agg, (False, VERDICT_REASONS[1])
            code = st.verdict_from_sums(self.coll, part)
            return agg, (code == 0, VERDICT_REASONS[code])
        finally:
            st.synchronize()
            self.coll.free(part)
            st.free(dC, dAl, dL, dR)
            if own:
                st.free(dS)

    def aggregate_sharded(self, vk_all, messages_all, sig_local):
        """-> aggregate [l][d] int32 == aggregate(params, keys, messages, signatures).signature_hat, on every rank"""
        return self.aggregate_verify_sharded(vk_all, messages_all, sig_local)[0]

    def verify_sharded(self, vk_all, messages_all, aggregate):
        """-> (bool, reason) == verify(params, keys, messages, aggregate_signature) on every rank; the SIGNERS of the
        verification target sum_i (vkL_i c_i + vkR_i) alpha_i (fusion.py:706-714) are sharded, one all-reduce of d int64."""
        import numpy as np
        from .context import VERDICT_REASONS
        st, l, d = self.steps, self.l, self.d
        nk = vk_all.shape[0] if hasattr(vk_all, "ptr") else np.asarray(vk_all).reshape(-1, 2, d).shape[0]
        if nk > self.params.capacity:
            return False, VERDICT_REASONS[1]
        if nk != len(messages_all):
            return False, VERDICT_REASONS[2]
        n, lo, hi, dC, c_row0, dAl, dL, dR = self._local_operands(vk_all, messages_all)
        m = hi - lo
        dS, own = st.take(aggregate, (l, d))
        part = self.coll.alloc_i64(d)
        try:
            if m:
                st.target_partial(dL, dR, dC, c_row0, dAl, self.coll, part, m)
            self.coll.allreduce(part)
            code = st.verdict_with_target(dS, st.centred(self.coll, part, d))
            return code == 0, VERDICT_REASONS[code]
        finally:
            st.synchronize()
            self.coll.free(part)
            st.free(dC, dAl, dL, dR)
            if own:
                st.free(dS)
