"""where the end-to-end aggregate / verify time goes (host pipeline pieces timed one by one)"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "fusion-cryptography_amd")); sys.path.insert(0, R)
import numpy as np
import fusion.fusion as F
from fusion_hip.scheme import BatchScheme
from fusion_hip import hostpipe
params = F.fusion_setup(256, 2026)
bs = BatchScheme(params)
n = 1024
seeds = [10_000 + 2 * i for i in range(n)]
msgs = [f"synthetic message {i:06d}" for i in range(n)]
sk, vk, vkd = bs.keygen_batch(seeds, device=True, keep_vk=True)
sig = bs.sign_batch(sk, vkd, msgs, device=True)
def T(label, fn, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); best = min(best, time.perf_counter() - t0)
    print(f"  {label:42s} {best * 1e3:8.2f} ms")
    return r
print("threads", bs.threads)
agg = T("aggregate (whole)", lambda: bs.aggregate(vk, msgs, sig))
T("verify (whole)", lambda: bs.verify(vk, msgs, agg))
vk3 = np.ascontiguousarray(vk, dtype=np.int32).reshape(-1, 2, bs.d)
L, Rr = np.ascontiguousarray(vk3[:, 0]), np.ascontiguousarray(vk3[:, 1])
order = T("sort_by_vk_string", lambda: hostpipe.sort_by_vk_string(bs.P, L, Rr, bs.threads))
Ls, Rs = T("permute keys (numpy)", lambda: (L[order], Rr[order]))
ms = [msgs[i] for i in order]
c_hat, pre = T("challenges (device pipeline + copies)", lambda: bs.challenges(np.stack([Ls, Rs], axis=1), ms))
for thr in (1, 2, 3, 4, 8, bs.threads, 1, 4):
    al = T(f"aggregation_coefficients, {thr} threads", lambda: hostpipe.aggregation_coefficients(bs.P, Ls, Rs, pre, c_hat, thr))
T("ntt_forward(alpha) host face", lambda: bs.ctx.ntt_forward(al))
