"""One rank of tests/test_gpu_dist.py: the multi-GPU exchange step with the LIBRARY's partial sums and a REAL collective
in between (torch.distributed; gloo here, because two ranks share the one GPU of the test box -- with one GPU per rank the
same code runs over RCCL, and bench.py issues the collective through fz_allreduce_i64).
argv: rank world port secpar n_signers out_dir"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "fusion-cryptography_amd"), ROOT):
    sys.path.insert(0, p)
rank, world, port, secpar, n = (int(x) for x in sys.argv[1:6])
out_dir = sys.argv[6]
import numpy as np
import torch
import torch.distributed as dist
import fusion_hip
from fusion_hip.dist import allreduce_sum_i64, shard_range
from oracle import oracle as O          # parameters and the synthetic generator only (the checking happens in the test)

os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
P = O.PARAMS[secpar]
q, d, l = P["q"], P["d"], P["rank"]
ctx = fusion_hip.Context(q, d, P["root"], P["inv_root"], device=0)
stream = torch.cuda.Stream(dev)
torch.cuda.set_stream(stream)
ctx.set_stream(stream.cuda_stream)
# every rank derives the SAME signer set from the seed, then keeps only its contiguous block (as SURVEY 8e partitions it)
rng = np.random.default_rng(4242 + secpar)
A = torch.from_numpy(O.splitmix_centered(5, l * d).reshape(l, d)).to(dev)
coef_all = (rng.integers(1, 53, size=(n, 2, d)) * rng.choice(np.array([-1, 1]), size=(n, 2, d))).astype(np.int32)
c_all = np.zeros((n, d), np.int32)
a_all = np.zeros((n, d), np.int32)
for i in range(n):
    c_all[i, rng.choice(d, P["omega_ch"], replace=False)] = rng.choice(np.array([-1, 1], np.int32), P["omega_ch"])
    a_all[i, rng.choice(d, P["omega_ag"], replace=False)] = rng.choice(np.array([-1, 1], np.int32), P["omega_ag"])
lo, hi = shard_range(n, rank, world)
m = hi - lo
coef = torch.from_numpy(coef_all[lo:hi]).to(dev)
sk = torch.empty((m, 2, l, d), dtype=torch.int32, device=dev)
vk = torch.empty((m, 2, d), dtype=torch.int32, device=dev)
ctx.keygen_core_bcast_dev(A.data_ptr(), coef.data_ptr(), sk.data_ptr(), vk.data_ptr(), m, l)
cc, aa = torch.from_numpy(c_all[lo:hi]).to(dev), torch.from_numpy(a_all[lo:hi]).to(dev)
c_hat, al_hat = torch.empty_like(cc), torch.empty_like(aa)
ctx.ntt_forward_dev(cc.data_ptr(), c_hat.data_ptr(), m)
ctx.ntt_forward_dev(aa.data_ptr(), al_hat.data_ptr(), m)
sig = torch.empty((m, l, d), dtype=torch.int32, device=dev)
ctx.sign_core_dev(sk.data_ptr(), c_hat.data_ptr(), sig.data_ptr(), m, l)
vkL, vkR = vk[:, 0].contiguous(), vk[:, 1].contiguous()
part = torch.zeros(l * d + d, dtype=torch.int64, device=dev)
ctx.aggregate_target_partial_batch_dev(sig.data_ptr(), al_hat.data_ptr(), vkL.data_ptr(), vkR.data_ptr(), c_hat.data_ptr(),
                                       part.data_ptr(), l * d, part[l * d:].data_ptr(), d, 1, m, l)
torch.cuda.synchronize(dev)
local = part.cpu().numpy().copy()
allreduce_sum_i64(part)                                   # the ONE exchange step
torch.cuda.synchronize(dev)
verd = torch.full((1,), -1, dtype=torch.int32, device=dev)
ctx.verify_partials_batch_async_dev(A.data_ptr(), part.data_ptr(), l * d, part[l * d:].data_ptr(), d, 1, l, P["beta_vf"], d, verd.data_ptr())
bad = part.clone()
bad[7] += 1
verd_bad = torch.full((1,), -1, dtype=torch.int32, device=dev)
ctx.verify_partials_batch_async_dev(A.data_ptr(), bad.data_ptr(), l * d, bad[l * d:].data_ptr(), d, 1, l, P["beta_vf"], d, verd_bad.data_ptr())
agg = torch.empty((l, d), dtype=torch.int32, device=dev)
ctx.reduce_i64_dev(part.data_ptr(), agg.data_ptr(), l * d)
torch.cuda.synchronize(dev)
np.savez(os.path.join(out_dir, f"rank{rank}.npz"), local=local, total=part.cpu().numpy(), agg=agg.cpu().numpy(), verdict=verd.cpu().numpy(),
         verdict_bad=verd_bad.cpu().numpy(), sig=sig.cpu().numpy(), vk=vk.cpu().numpy(), c_hat=c_hat.cpu().numpy(),
         al_hat=al_hat.cpu().numpy(), A=A.cpu().numpy(), lo=lo, hi=hi)
dist.barrier()
dist.destroy_process_group()
