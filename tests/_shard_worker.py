"""One rank of tests/test_gpu_sharded.py: the END-TO-END sharded aggregate() / verify() (fusion_hip.dist.ShardedScheme) on a
reference-generated signer set.  Every rank regenerates ONLY its block's keys and signatures on the device
(BatchScheme.keygen_batch / sign_batch from the golden seeds and messages), takes all verification keys from the golden file
(they are public), and runs aggregate_verify_sharded / verify_sharded with a real torch.distributed collective (gloo: the
ranks share the test box's one GPU; with a GPU per rank the same code runs over RCCL).
argv: rank world port secpar kind lo hi out_dir [alpha_mode]      kind = "many" (scheme_many_*.npz) | "small" (scheme_*.npz)
alpha_mode: "replicated" (the serial sponge of hash_ag on every rank) | "root" (rank 0 alone + broadcast) | "auto" """
import faulthandler
import hashlib
import json
import os
import sys

faulthandler.dump_traceback_later(240, exit=False)       # a rank that hangs says where (the test kills it at 300 s and prints this)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "fusion-cryptography_amd"), ROOT):
    sys.path.insert(0, p)
rank, world, port, secpar = (int(x) for x in sys.argv[1:5])
kind, lo_s, hi_s, out_dir = sys.argv[5], int(sys.argv[6]), int(sys.argv[7]), sys.argv[8]
alpha_mode = sys.argv[9] if len(sys.argv) > 9 else "auto"
import numpy as np
import torch
import torch.distributed as dist
import fusion.fusion as F
from fusion_hip.dist import ShardedScheme, TorchCollective, shard_range
from fusion_hip.scheme import BatchScheme

G = os.path.join(ROOT, "tests", "golden")
if kind == "many":
    S = np.load(os.path.join(G, f"scheme_many_{secpar}.npz"))
    with open(os.path.join(G, "scheme_many.json")) as fh:
        meta = json.load(fh)[str(secpar)]
elif kind.startswith("full"):                       # full-size goldens: the keys are regenerated (and digest-checked by the test)
    S = None
    with open(os.path.join(G, "scheme_full.json")) as fh:
        meta = json.load(fh)[kind.split(":")[1]]
    kind = "full"
else:
    S = np.load(os.path.join(G, f"scheme_{secpar}.npz"))
    with open(os.path.join(G, "scheme.json")) as fh:
        meta = json.load(fh)[str(secpar)]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
params = F.fusion_setup(secpar, meta["setup_seed"])
bs = BatchScheme(params, device=0, threads=4)
if kind == "full":                                # 2 MiB of public keys: every rank derives them from the public seeds
    _, vk_all = bs.keygen_batch(meta["key_seeds"][lo_s:hi_s])
    meta["agg"] = {f"{lo_s}_{hi_s}": {"tampered_at": meta["tampered_at"]}}
else:
    vk_all = S["vk"][lo_s:hi_s]                   # public: every rank has all verification keys (reference-generated)
msgs = meta["messages"][lo_s:hi_s]
seeds = meta["key_seeds"][lo_s:hi_s]
n = hi_s - lo_s
lo, hi = shard_range(n, rank, world)
sh = ShardedScheme(bs, rank, world, TorchCollective(bs.ctx, 0), alpha_mode=alpha_mode)
if hi > lo:                                       # this rank's block: keys and signatures made here, resident on the device
    sk, vk_loc, vk_dev = bs.keygen_batch(seeds[lo:hi], device=True, keep_vk=True)
    assert np.array_equal(vk_loc, vk_all[lo:hi]), "regenerated verification keys differ from the reference's"
    sig = bs.sign_batch(sk, vk_dev, msgs[lo:hi], device=True)
    sig_host = sig.numpy()
else:
    sig, sig_host = np.zeros((0, bs.l, bs.d), np.int32), np.zeros((0, bs.l, bs.d), np.int32)
agg, verdict = sh.aggregate_verify_sharded(vk_all, msgs, sig)
agg2 = sh.aggregate_sharded(vk_all, msgs, sig)
ver2 = sh.verify_sharded(vk_all, msgs, agg)
bad = agg.copy()
t_row, t_col = meta["agg"][f"{lo_s}_{hi_s}"]["tampered_at"] if kind in ("many", "full") else (0, 0)
bad[t_row, t_col] += 1
ver_bad = sh.verify_sharded(vk_all, msgs, bad)
swapped = list(msgs)
swapped[0], swapped[-1] = swapped[-1], swapped[0]
ver_swapped = sh.verify_sharded(vk_all, swapped, agg)
ver_short = sh.verify_sharded(vk_all, msgs[:-1], agg)
np.savez(os.path.join(out_dir, f"rank{rank}.npz"), agg=agg, agg2=agg2, lo=lo, hi=hi, vk_sha=hashlib.sha256(np.ascontiguousarray(vk_all, dtype="<i4").tobytes()).hexdigest(),
         sig_sha=np.array([hashlib.sha256(np.ascontiguousarray(r, dtype="<i4").tobytes()).hexdigest() for r in sig_host]))
with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as fh:
    json.dump(dict(verdict=list(verdict), verify=list(ver2), tampered=list(ver_bad), swapped=list(ver_swapped),
                   short=list(ver_short)), fh)
print(f"rank {rank}: results written", flush=True)
dist.barrier()
print(f"rank {rank}: barrier passed", flush=True)
dist.destroy_process_group()
bs.close()
print(f"rank {rank}: done", flush=True)
