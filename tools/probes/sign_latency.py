"""Where the time of BatchScheme.sign_batch(1024 signers, BASELINE configs[2]) goes: message packing, the challenge call,
the signing launch, the final synchronisation -- each as the minimum and median of 30 calls.  Run on a GPU box."""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "fusion-cryptography_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np
import fusion.fusion as F
from fusion_hip.numa import pin_to_gpu_node
pin_to_gpu_node(0)
from fusion_hip import hostpipe
from fusion_hip.scheme import BatchScheme, DeviceArray


def stat(name, fn, reps=30):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    print(f"  {name:<58s} min {min(ts) * 1e6:8.1f} us   median {statistics.median(ts) * 1e6:8.1f} us", flush=True)
    return min(ts)


def main():
    secpar = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    params = F.fusion_setup(secpar, 2026)
    bs = BatchScheme(params)
    ctx = bs.ctx
    print(f"secpar {secpar}, FZ_SHAKE_FORM={os.environ.get('FZ_SHAKE_FORM', '(by batch size)')}")
    for n in (1024, 2048):
        seeds = [10_000 + 2 * i for i in range(n)]
        msgs = [f"synthetic message {i:06d}" for i in range(n)]
        sk, vk, vkd = bs.keygen_batch(seeds, device=True, keep_vk=True)
        print(f"N = {n}")
        bs.sign_batch(sk, vkd, msgs, device=True).free()
        t = stat("sign_batch(device keys, signatures stay on the device)", lambda: bs.sign_batch(sk, vkd, msgs, device=True).free())
        print(f"  -> {n / t / 1e6:.2f} M signatures/s")
        stat("  hostpipe._pack_messages", lambda: hostpipe._pack_messages(msgs))
        blob, off = hostpipe._pack_messages(msgs)
        dC = DeviceArray(ctx, (n, bs.d))
        dS = DeviceArray(ctx, (n, bs.l, bs.d))

        def chal():
            ctx.challenge_msgs_dev(bs.P, vkd.ptr, blob, off, n, dC.ptr, False)
        stat("  challenge_msgs_dev, call only (asynchronous)", chal)
        ctx.synchronize()

        def chal_sync():
            ctx.challenge_msgs_dev(bs.P, vkd.ptr, blob, off, n, dC.ptr, False)
            ctx.synchronize()
        stat("  challenge_msgs_dev + synchronize", chal_sync)

        def sign_sync():
            ctx.sign_core_dev(sk.ptr, dC.ptr, dS.ptr, n, bs.l)
            ctx.synchronize()
        stat("  sign_core_dev + synchronize", sign_sync)

        def both():
            ctx.challenge_msgs_dev(bs.P, vkd.ptr, blob, off, n, dC.ptr, False)
            ctx.sign_core_dev(sk.ptr, dC.ptr, dS.ptr, n, bs.l)
            ctx.synchronize()
        stat("  challenge + sign + synchronize (packed messages given)", both)
        for b in (dC, dS, sk, vkd):
            b.free()


if __name__ == "__main__":
    main()
