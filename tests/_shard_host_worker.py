"""One rank of tests/test_dist_cpu.py::test_sharded_host_logic_*: the HOST logic of the sharded aggregate() / verify()
(fusion_hip.dist: shard_range, sharded_alpha in both modes, the int64 all-reduce; hostpipe: sort by str(vk), the serial
SHAKE-256 of hash_ag, decoders, samplers) at up to 8 ranks over gloo WITHOUT a GPU -- the device steps of ShardedScheme
(transforms, keygen / sign cores, partial sums, verification) are stood in for by the C oracle, which is what the GPU tests
check those kernels against.  Compared with what the REFERENCE computed over all signers (tests/golden/scheme_full_*.npz).
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
from fusion_hip.dist import TorchCollective, allreduce_sum_i64, resolve_alpha_mode, shard_range, sharded_alpha
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
everything = amode == "replicated" or rank == 0
coll = TorchCollective(None, None)


def challenges(lo_, hi_):
    if hi_ == lo_:
        return np.zeros((0, d), np.int32), np.zeros((0, 32), np.uint8)
    coefs, pre = hostpipe.challenge_coefficients(P, L[lo_:hi_], R[lo_:hi_], msgs[lo_:hi_], 2)
    return orc.ntt_forward(coefs, q, params.root).reshape(-1, d), pre


if everything:
    c_hat, pre = challenges(0, n)
    c_blk = c_hat[lo:hi]
else:
    c_blk, _ = challenges(lo, hi)


def compute():
    order = hostpipe.sort_by_vk_string(P, L, R, 2)
    alpha_sorted = hostpipe.aggregation_coefficients(P, L[order], R[order], pre[order], c_hat[order], 2)
    alpha = np.empty_like(alpha_sorted)
    alpha[order] = alpha_sorted
    return alpha


alpha = sharded_alpha(rank, world, amode, coll, n, d, compute if everything else None)
al_blk = orc.ntt_forward(alpha[lo:hi], q, params.root).reshape(-1, d) if m else np.zeros((0, d), np.int32)
sig = orc.sign_core(sk, c_blk, q).reshape(m, l, d) if m else np.zeros((0, l, d), np.int32)
# exact int64 partial sums of the aggregate [l][d] and the verification target [d] over this rank's signers
part = torch.zeros(l * d + d, dtype=torch.int64)
if m:
    pa = part[:l * d].view(l, d).numpy()
    for i in range(m):
        prod = (sig[i].astype(np.int64) * al_blk[i].astype(np.int64)[None, :]) % q
        pa += np.where(prod > q // 2, prod - q, prod)
    inner = (L[lo:hi].astype(np.int64) * c_blk.astype(np.int64) + R[lo:hi]) % q
    tg = (inner * (al_blk.astype(np.int64) % q)) % q
    part[l * d:] += torch.from_numpy(np.where(tg > q // 2, tg - q, tg).sum(axis=0))
allreduce_sum_i64(part)                                          # the ONE exchange step
tot = part.numpy()
cent = lambda v: np.where(v % q > q // 2, v % q - q, v % q).astype(np.int32)
agg = cent(tot[:l * d]).reshape(l, d)
target = cent(tot[l * d:])
observed = orc.matvec(A, agg[None], q).reshape(d)
mx, wt = orc.norm_weight(orc.ntt_inverse(agg, q, params.inv_root).reshape(l, d), q)
if n > params.capacity:
    verdict = [False, "Too many keys."]
elif not np.array_equal(observed, target):
    verdict = [False, "Target doesn't match image of aggregate signature."]
elif mx.max() > params.beta_vf:
    verdict = [False, "Norm of aggregate signature too large."]
else:
    verdict = [True, ""]
np.savez(os.path.join(out_dir, f"rank{rank}.npz"), agg=agg, lo=lo, hi=hi,
         vk_sha=hashlib.sha256(np.ascontiguousarray(vk, dtype="<i4").tobytes()).hexdigest(),
         alpha_sha=hashlib.sha256(np.ascontiguousarray(alpha, dtype="<i4").tobytes()).hexdigest())
with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as fh:
    json.dump(dict(verdict=verdict, mode=amode, ran_sponge=bool(everything)), fh)
dist.barrier()
dist.destroy_process_group()
