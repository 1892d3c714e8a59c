import sys, os, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
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
bs.keygen_batch(seeds[:4])
for rep in range(3):
    t0 = time.perf_counter(); sk, vk, vkd = bs.keygen_batch(seeds, device=True, keep_vk=True); t1 = time.perf_counter()
    print("keygen", (t1 - t0) * 1e3, "ms")
    t0 = time.perf_counter(); pre = hostpipe.hash_messages(bs.P, msgs); t1 = time.perf_counter(); print(" prehash", (t1 - t0) * 1e3)
    t0 = time.perf_counter(); polys = hostpipe.sample_secret_polys(seeds, params.modulus, params.degree, params.beta_sk, params.omega_sk, bs.threads); t1 = time.perf_counter(); print(" sample", (t1 - t0) * 1e3)
    for k in range(2):
        t0 = time.perf_counter(); s = bs.sign_batch(sk, vkd, msgs, device=True); t1 = time.perf_counter(); print(" sign(dev vk)", (t1 - t0) * 1e3); s.free()
    t0 = time.perf_counter(); s = bs.sign_batch(sk, vk, msgs, device=True); t1 = time.perf_counter(); print(" sign(host vk)", (t1 - t0) * 1e3); s.free()
    t0 = time.perf_counter(); c, _ = bs.challenges_dev(vkd, msgs); bs.ctx.synchronize(); t1 = time.perf_counter(); print(" challenges_dev", (t1 - t0) * 1e3); c.free()
    sk.free(); vkd.free()
