"""One rank of tests/test_dist_cpu.py::test_sharded_host_logic_*: fusion_hip.dist.ShardedScheme ITSELF at up to 8 ranks over
gloo WITHOUT a GPU.  The class holds the control flow of the sharded aggregate() / verify() -- blocks, the two alpha modes
(sponge on every rank, or on rank 0 + broadcast), the offset of a rank's challenges inside the rows it computed, ranks that
own no signers, the one int64 all-reduce, the verdicts -- and reaches the device only through its `steps` object
(fusion_hip.dist.HipSteps in the product).  Here OracleSteps stands in: the same methods on numpy arrays with the C oracle
behind them, which is what the GPU tests check those kernels against (round 4 re-implemented the flow in this file instead,
so a regression in ShardedScheme._local_operands would have passed: ADVICE r04).  Compared with what the REFERENCE computed
over all signers (tests/golden/scheme_full_*.npz).
argv: rank world port tag n mode out_dir"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "fusion-cryptography_amd"), ROOT):
    sys.path.insert(0, p)
rank, world, port = (int(x) for x in sys.argv[1:4])
tag, n, mode, out_dir = sys.argv[4], int(sys.argv[5]), sys.argv[6], sys.argv[7]
import numpy as np
import torch
import torch.distributed as dist
import fusion.fusion as F
from fusion_hip import hostpipe
from fusion_hip.dist import ShardedScheme, TorchCollective, resolve_alpha_mode, shard_range
from oracle import oracle as O

torch.set_num_threads(1)
with open(os.path.join(ROOT, "tests", "golden", "scheme_full.json")) as fh:
    meta = json.load(fh)[tag]
secpar = meta["secpar"]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
dist.init_process_group("gloo", rank=rank, world_size=world)
params = F.fusion_setup(secpar, meta["setup_seed"])             # host only: the sampler is Python / C
P = hostpipe.scheme_params(params)
q, d, l = params.modulus, params.degree, params.num_rows_sk
orc = O.COracle()
A = np.array([z.values for row in params.public_challenge.matrix for z in row], dtype=np.int32)
seeds, msgs = meta["key_seeds"][:n], meta["messages"][:n]
lo, hi = shard_range(n, rank, world)
m = hi - lo
# this rank's block: keys and signatures (device: fz_sample_secret_polys_dev + keygen_core_bcast + challenge pipeline + sign_core)
polys = hostpipe.sample_secret_polys(np.array(seeds[lo:hi], dtype=np.uint64), q, d, params.beta_sk, params.omega_sk, 1) if m else np.zeros((0, 2, d), np.int32)
coef = np.repeat(polys[:, :, None, :], l, axis=2)                # the reference samples every row of a half with the same seed
sk, vk_blk = orc.keygen_core(A, coef, q, params.root) if m else (np.zeros((0, 2, l, d), np.int32), np.zeros((0, 2, d), np.int32))
parts = [None] * world                                           # verification keys are public: everyone gets all of them
dist.all_gather_object(parts, vk_blk)
vk = np.concatenate(parts)
L, R = np.ascontiguousarray(vk[:, 0]), np.ascontiguousarray(vk[:, 1])
amode = resolve_alpha_mode(mode, world)
coll = TorchCollective(None, None)                               # host tensors over gloo


class OracleSteps:
    """fusion_hip.dist.HipSteps' interface on host arrays: hostpipe for the hashing (the product's own host code), the C oracle
    for every kernel.  `calls` records what ShardedScheme asked for, so the test can assert the control flow it took."""

    def __init__(self):
        self.params, self.d, self.l, self.calls = params, d, l, []

    def split_vk(self, vk_all):
        v = np.ascontiguousarray(np.asarray(vk_all, dtype=np.int32).reshape(-1, 2, d))
        return v, np.ascontiguousarray(v[:, 0]), np.ascontiguousarray(v[:, 1])

    def challenges(self, vk_rows, messages):
        self.calls.append(("challenges", len(messages)))
        coefs, pre = hostpipe.challenge_coefficients(P, np.ascontiguousarray(vk_rows[:, 0]), np.ascontiguousarray(vk_rows[:, 1]), list(messages), 2)
        c_hat = orc.ntt_forward(coefs, q, params.root).reshape(-1, d)
        return c_hat, c_hat, pre

    def alpha_rows(self, L_, R_, pre, c_hat):
        self.calls.append(("sponge", len(L_)))
        order = hostpipe.sort_by_vk_string(P, L_, R_, 2)
        alpha_sorted = hostpipe.aggregation_coefficients(P, L_[order], R_[order], pre[order], c_hat[order], 2)
        alpha = np.empty_like(alpha_sorted)
        alpha[order] = alpha_sorted
        return alpha

    def rows(self, arr):
        return np.array(arr, dtype=np.int32, copy=True)

    def empty_rows(self, shape):
        return np.zeros(shape, np.int32)

    def take(self, a, shape):
        return np.asarray(a, dtype=np.int32).reshape(shape), False

    def free(self, *bufs):
        pass

    def synchronize(self):
        pass

    def transform_rows(self, buf, m_):
        buf[:m_] = orc.ntt_forward(buf[:m_], q, params.root).reshape(m_, d)

    @staticmethod
    def _cent64(v):
        r = v % q
        return np.where(r > q // 2, r - q, r)

    def partial_sums(self, sig, al, L_, R_, c, c_row0, coll_, part, m_):
        self.calls.append(("partial_sums", m_, c_row0))
        out = coll_.to_numpy(part)                               # (a host tensor's numpy view shares its memory)
        pa = out[:l * d].reshape(l, d)
        cb = c[c_row0:c_row0 + m_].astype(np.int64)
        for i in range(m_):                                      # fusion.py:670-676, one centring per reference cent()
            pa += self._cent64(sig[i].astype(np.int64) * al[i].astype(np.int64)[None, :])
        inner = self._cent64(self._cent64(L_[:m_].astype(np.int64) * cb) + R_[:m_])      # fusion.py:706-714
        out[l * d:] += self._cent64(inner * al[:m_].astype(np.int64)).sum(axis=0)

    def target_partial(self, L_, R_, c, c_row0, al, coll_, part, m_):
        self.calls.append(("target_partial", m_, c_row0))
        cb = c[c_row0:c_row0 + m_].astype(np.int64)
        inner = self._cent64(self._cent64(L_[:m_].astype(np.int64) * cb) + R_[:m_])
        coll_.to_numpy(part)[:d] += self._cent64(inner * al[:m_].astype(np.int64)).sum(axis=0)

    def centred(self, coll_, part, count):
        return self._cent64(coll_.to_numpy(part)[:count]).astype(np.int32)

    def _verdict(self, agg, target):
        observed = orc.matvec(A, agg[None], q).reshape(d)
        mx, wt = orc.norm_weight(orc.ntt_inverse(agg, q, params.inv_root).reshape(l, d), q)
        if not np.array_equal(observed, target):
            return 3
        if mx.max() > params.beta_vf:
            return 4
        return 5 if wt.max() > params.omega_vf else 0

    def verdict_from_sums(self, coll_, part):
        tot = coll_.to_numpy(part)
        return self._verdict(self._cent64(tot[:l * d]).astype(np.int32).reshape(l, d), self._cent64(tot[l * d:]).astype(np.int32))

    def verdict_with_target(self, agg, target):
        return self._verdict(np.asarray(agg, dtype=np.int32).reshape(l, d), np.asarray(target, dtype=np.int32))


# this rank's signatures (device: challenge pipeline + sign_core); the challenges of its own block only
c_blk = OracleSteps().challenges(vk[lo:hi], msgs[lo:hi])[0] if m else np.zeros((0, d), np.int32)
sig = orc.sign_core(sk, c_blk, q).reshape(m, l, d) if m else np.zeros((0, l, d), np.int32)
steps = OracleSteps()
sh = ShardedScheme(None, rank, world, coll, alpha_mode=mode, steps=steps)
assert sh.alpha_mode == amode
agg, verdict = sh.aggregate_verify_sharded(vk, msgs, sig)
everything = any(c_[0] == "sponge" for c_ in steps.calls)
# what the class must have asked of its device steps
ch = [c_ for c_ in steps.calls if c_[0] == "challenges"]
ps = [c_ for c_ in steps.calls if c_[0] == "partial_sums"]
if amode == "replicated" or rank == 0:
    assert everything and ch == [("challenges", n)] and (ps == [("partial_sums", m, lo)] if m else ps == [])
else:
    assert not everything and (ch == [("challenges", m)] and ps == [("partial_sums", m, 0)] if m else ch == [] and ps == [])
# verify_sharded of that aggregate: the target's signers sharded, d int64 exchanged
steps2 = OracleSteps()
v2 = ShardedScheme(None, rank, world, coll, alpha_mode=mode, steps=steps2).verify_sharded(vk, msgs, agg)
assert v2 == verdict, (v2, verdict)
bad = agg.copy()
bad[0, 0] += 1
assert ShardedScheme(None, rank, world, coll, alpha_mode=mode, steps=OracleSteps()).verify_sharded(vk, msgs, bad) == \
    (False, "Target doesn't match image of aggregate signature.")
verdict = list(verdict)
np.savez(os.path.join(out_dir, f"rank{rank}.npz"), agg=agg, lo=lo, hi=hi,
         vk_sha=hashlib.sha256(np.ascontiguousarray(vk, dtype="<i4").tobytes()).hexdigest())
with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as fh:
    json.dump(dict(verdict=verdict, mode=amode, ran_sponge=bool(everything)), fh)
dist.barrier()
dist.destroy_process_group()
