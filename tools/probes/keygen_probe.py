import sys, os, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(R, "fusion-cryptography_amd")); sys.path.insert(0, R)
import numpy as np
import fusion.fusion as F
from fusion_hip.scheme import BatchScheme
params = F.fusion_setup(256, 2026)
bs = BatchScheme(params)
for n in (1024, 16384):
    seeds = [10_000 + 2 * i for i in range(n)]
    bs.keygen_batch(seeds[:4])
    for dev in (True, False):
        bs.device_sampler = dev
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); sk, vk, vkd = bs.keygen_batch(seeds, device=True, keep_vk=True); best = min(best, time.perf_counter() - t0)
            sk.free(); vkd.free()
        print(f"n={n} device_sampler={dev}: keygen_batch {best*1e3:.2f} ms  {n/best/1e6:.2f} M keys/s")
    sd = np.array(seeds, dtype=np.uint64)
    from fusion_hip.context import DeviceArray
    c = DeviceArray(bs.ctx, (n, 2, 256))
    t0 = time.perf_counter(); bs.ctx.sample_secret_polys_dev(sd, params.modulus, 256, params.beta_sk, params.omega_sk, c.ptr); print(f"   sampler alone {(time.perf_counter()-t0)*1e3:.3f} ms")
    c.free()
