#!/usr/bin/env python3
"""ONE Python thread submitting BASELINE-sized calls (1024 keys + 1024 signatures each) to the asynchronous batch queue of the
C ABI (fz_queue_*, fusion_hip.queue.BatchQueue), against the same calls through BatchScheme one after the other, and against
what W Python threads with private contexts reach (tools/probes/concurrent_batches.py; profiles/r03_concurrent_batches.txt: 0.93 M
pairs/s alone, 3.8-4.3 M/s with 8-16 threads and GPU_MAX_HW_QUEUES=16).  Reference call pattern: fusion.py:338-373, :534-557.
usage: queue_probe.py [--secpar 128|256] [--n 1024] [--calls 96]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "fusion-cryptography_amd"))

import numpy as np  # noqa: E402
import fusion.fusion as F  # noqa: E402
from fusion_hip.numa import pin_to_gpu_node  # noqa: E402
from fusion_hip.queue import BatchQueue, PackedMessages  # noqa: E402
from fusion_hip.scheme import BatchScheme  # noqa: E402


def arg(name, default, cast):
    return cast(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


def main():
    pin_to_gpu_node(0)
    secpar, n, calls = arg("--secpar", 256, int), arg("--n", 1024, int), arg("--calls", 96, int)
    params = F.fusion_setup(secpar, 2026)
    seeds = [np.arange(n, dtype=np.uint64) * 2 + np.uint64(70_000 + 4096 * c) for c in range(calls)]
    msg_list = [f"synthetic message {i:06d}" for i in range(n)]
    msgs = PackedMessages(msg_list)
    print(f"# secpar {secpar}: {n} keys + {n} signatures per call, {calls} calls per pass, ONE submitting Python thread; "
          f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', 'default')}")
    bs = BatchScheme(params, threads=2)
    sl = [int(x) for x in seeds[0]]
    for _ in range(2):
        sk, vk, vkd = bs.keygen_batch(sl, device=True, keep_vk=True)
        bs.sign_batch(sk, vkd, msg_list, device=True).free()
        sk.free()
        vkd.free()
    t0 = time.perf_counter()
    reps = 24
    for _ in range(reps):
        sk, vk, vkd = bs.keygen_batch(sl, device=True, keep_vk=True)
        sig = bs.sign_batch(sk, vkd, msg_list, device=True)
        for b in (sig, sk, vkd):
            b.free()
    dt = time.perf_counter() - t0
    print(f"BatchScheme, one call after the other      {n * reps / dt:12,.0f} pairs/s   {dt / reps * 1e3:7.3f} ms per call")
    for max_rows in (16384, 4096):
        for workers in (1, 2, 3, 4):
            with BatchQueue(params, workers=workers, max_rows=max_rows) as bq:
                for c in range(min(calls, 16)):
                    bq.submit_keygen_sign(seeds[c], msgs, discard=True)
                bq.drain()
                bq.collect_discarded()
                c0, b0, _ = bq.stats()
                best, sub = 1e30, 0.0
                for _ in range(3):
                    t0 = time.perf_counter()
                    for c in range(calls):
                        bq.submit_keygen_sign(seeds[c], msgs, discard=True)
                    t1 = time.perf_counter()
                    bq.drain()
                    t2 = time.perf_counter()
                    bq.collect_discarded()
                    if t2 - t0 < best:
                        best, sub = t2 - t0, t1 - t0
                c1, b1, _ = bq.stats()
                # and with every result kept and read, as a service does: at most WINDOW calls outstanding, wait() for the oldest,
                # take its rows, release it, submit the next -- so released blocks come back from the pool (round 4's probe submitted
                # all 96 calls before the first release: 8.4 GB of fresh hipMalloc, which is what it then measured); second pass timed
                WINDOW = 16
                for timed in (False, True):
                    t0 = time.perf_counter()
                    tickets = [bq.submit_keygen_sign(seeds[c], msgs) for c in range(min(WINDOW, calls))]
                    for c in range(calls):
                        r = bq.wait(tickets[c])
                        assert r.n == n and r.sig_ptr
                        r.release()
                        if c + WINDOW < calls:
                            tickets.append(bq.submit_keygen_sign(seeds[c + WINDOW], msgs))
                    kept = time.perf_counter() - t0
                print(f"queue workers={workers} max_rows={max_rows:5d}   {n * calls / best:12,.0f} pairs/s discarded   "
                      f"{n * calls / kept:12,.0f} pairs/s kept+released   {(c1 - c0) / max(1, b1 - b0):5.1f} calls per batch   "
                      f"submit {sub / calls * 1e6:6.1f} us per call", flush=True)


if __name__ == "__main__":
    main()
