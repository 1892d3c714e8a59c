#!/usr/bin/env python3
"""Batches of the BASELINE size (1024 keys / signatures) in flight on several streams at once: W worker threads, each with a
BatchScheme(private_context=True) -- its own context and HIP stream --, each running keygen_batch + sign_batch on its own
1024 seeds and messages, repeatedly.  One such call alone is a latency chain on a few dozen waves (the device SHAKE-256: 108
permutations per signer; the MT19937 seeding); on separate streams the calls overlap.  Reports keys + signatures per second
for W = 1, 2, 4, 8, 16.  Needs an MI355X.      usage: concurrent_batches.py [--secpar 128|256] [--n 1024] [--seconds 1.5]"""
import os
import sys
import threading
import time

# the HIP runtime maps a process's streams onto 4 hardware queues unless told otherwise: more workers than queues take turns
# (read when the runtime starts: before the first HIP call).  Measured: 8 workers 2.6 M pairs/s with 4 queues, 3.9 M/s with 16.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "fusion-cryptography_amd"))

import fusion.fusion as F  # noqa: E402
from fusion_hip.numa import pin_to_gpu_node  # noqa: E402
from fusion_hip.scheme import BatchScheme  # noqa: E402


def arg(name, default, cast):
    return cast(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


def main():
    pin_to_gpu_node(0)
    secpar, n, seconds = arg("--secpar", 256, int), arg("--n", 1024, int), arg("--seconds", 1.5, float)
    params = F.fusion_setup(secpar, 2026)
    print(f"# secpar {secpar}: {n} keys + {n} signatures per call, W worker threads with a private context and stream each, {seconds} s per line; "
          f"GPU_MAX_HW_QUEUES={os.environ['GPU_MAX_HW_QUEUES']}")
    print(f"{'W':>3} {'calls':>7} {'keygen+sign pairs/s':>22} {'ms per call (one worker)':>26}")
    for W in (1, 2, 4, 8, 16):
        workers = [BatchScheme(params, threads=2, private_context=True) for _ in range(W)]
        counts, lat = [0] * W, [0.0] * W
        stop = threading.Event()
        start = threading.Barrier(W + 1)

        def run(i):
            bs = workers[i]
            seeds = [1_000_000 * (i + 1) + 2 * k for k in range(n)]
            msgs = [f"worker {i} message {k:06d}" for k in range(n)]
            sk, vk, vkd = bs.keygen_batch(seeds, device=True, keep_vk=True)          # warm-up: allocations, scratch
            bs.sign_batch(sk, vkd, msgs, device=True).free()
            sk.free()
            vkd.free()
            start.wait()
            while not stop.is_set():
                t0 = time.perf_counter()
                sk, vk, vkd = bs.keygen_batch(seeds, device=True, keep_vk=True)
                sig = bs.sign_batch(sk, vkd, msgs, device=True)
                bs.ctx.synchronize()
                lat[i] += time.perf_counter() - t0
                counts[i] += 1
                for b in (sig, sk, vkd):
                    b.free()
        threads = [threading.Thread(target=run, args=(i,)) for i in range(W)]
        for t in threads:
            t.start()
        start.wait()
        t0 = time.perf_counter()
        time.sleep(seconds)
        stop.set()
        for t in threads:
            t.join()
        dt = time.perf_counter() - t0
        calls = sum(counts)
        print(f"{W:3d} {calls:7d} {calls * n / dt:22,.0f} {1e3 * sum(lat) / max(1, calls):26.3f}")
        for bs in workers:
            bs.close()


if __name__ == "__main__":
    main()
