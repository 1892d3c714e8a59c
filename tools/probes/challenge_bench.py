"""Timing of the device challenge pipeline (fz_challenge_hat_dev) against the host pipeline (fz_challenge_coefficients on
the box's host threads + upload + forward NTT), secpar 256 unless given.  Scratch tool for DESIGN.md / profiles."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "fusion-cryptography_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np
import fusion.fusion as F
import fusion_hip
from fusion_hip.numa import pin_to_gpu_node
pin_to_gpu_node(0)          # host threads on the GPU's NUMA node (before the first HIP call)
from fusion_hip import hostpipe


def main():
    secpar = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    params = F.fusion_setup(secpar, 7)
    P = hostpipe.scheme_params(params)
    ctx = fusion_hip.get_context(params.modulus, params.degree, params.root, params.inv_root)
    print(f"secpar {secpar}, FZ_SHAKE_FORM={os.environ.get('FZ_SHAKE_FORM', '(by batch size)')}", flush=True)
    d, q = params.degree, params.modulus
    rng = np.random.default_rng(1)
    sizes = [int(a) for a in sys.argv[2:]] or [256, 1024, 2048, 3072, 4096, 16384, 65536]
    for n in sizes:
        vk = rng.integers(-(q // 2), q // 2 + 1, size=(n, 2, d)).astype(np.int32)
        msgs = [f"synthetic message {i:06d}" for i in range(n)]
        t0 = time.perf_counter()
        pre = hostpipe.hash_messages(P, msgs)
        t_pre = time.perf_counter() - t0
        dvk = fusion_hip.DeviceBuffer.from_numpy(ctx, vk)
        dout = fusion_hip.DeviceBuffer(ctx, n * d * 4)
        ctx.challenge_dev(P, dvk.ptr, pre, n, dout.ptr)          # warm-up (scratch growth)
        ctx.synchronize()
        reps = 5 if n <= 4096 else 2
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.challenge_dev(P, dvk.ptr, pre, n, dout.ptr)
        ctx.synchronize()
        t_dev = (time.perf_counter() - t0) / reps
        blob, off = hostpipe._pack_messages(msgs)
        ctx.challenge_msgs_dev(P, dvk.ptr, blob, off, n, dout.ptr)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.challenge_msgs_dev(P, dvk.ptr, blob, off, n, dout.ptr)
        ctx.synchronize()
        t_msg = (time.perf_counter() - t0) / reps
        line = (f"N={n:6d}  device hash_ch {t_dev * 1e3:8.3f} ms ({n / t_dev / 1e6:6.2f} M/s) from digests, {t_msg * 1e3:8.3f} ms "
                f"({n / t_msg / 1e6:6.2f} M/s) from messages;  prehash on host {t_pre * 1e3:6.2f} ms")
        if n <= 4096:
            t0 = time.perf_counter()
            coefs, _ = hostpipe.challenge_coefficients(P, vk[:, 0], vk[:, 1], msgs)
            hat = ctx.ntt_forward(coefs)
            t_host = time.perf_counter() - t0
            assert np.array_equal(hat, dout.to_numpy(np.int32, (n, d)))
            line += f"  host pipeline ({hostpipe.default_threads()} threads) {t_host * 1e3:8.2f} ms ({n / t_host / 1e6:6.3f} M/s)"
        print(line, flush=True)
        dvk.free()
        dout.free()


if __name__ == "__main__":
    main()
