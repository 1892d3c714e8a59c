#!/usr/bin/env python3
"""Few-signer aggregation and verification through the batch queue (reference call pattern fusion.py:655-677,
:680-728: one call per aggregate).
  (1) the kernel: ONE fz_aggregate_target_partial_ragged launch over G aggregates of N signers each, operands rotated through
      sets larger than the Infinity Cache (cold), against the same aggregates as G separate fz_aggregate_target_partial_batch
      launches: duration, bytes MOVED ((l + 4) rows of 4*d bytes per signer: sigma, alpha, c, vkL, vkR) and the fraction of 8 TB/s;
  (2) end to end: G calls of N signers submitted to the queue from ONE Python thread (hashing included: hash_ch on the device,
      hash_ag's serial sponge of every aggregate on a host thread), against BatchScheme.aggregate + verify called G times and
      against aggregate_many + verify_many.
usage: python tools/probes/queue_aggregates.py [--secpar 256]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "fusion-cryptography_amd"))
import numpy as np  # noqa: E402

import fusion.fusion as F  # noqa: E402
import fusion_hip  # noqa: E402
from fusion_hip.queue import BatchQueue, PackedMessages  # noqa: E402
from fusion_hip.scheme import BatchScheme  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--secpar", type=int, default=256)
ap.add_argument("--kernel-only", action="store_true")
args = ap.parse_args()
params = F.fusion_setup(args.secpar, 2026)
bs = BatchScheme(params)
ctx, l, d = bs.ctx, params.num_rows_sk, params.degree
s_ = ctx.stream_create()
ctx.set_stream(s_)
DB = fusion_hip.DeviceBuffer
print(f"secpar {args.secpar}: l = {l}, d = {d}; bytes moved per signer = (l + 4) rows x {4 * d} B = {(l + 4) * 4 * d}")
print("\n(1) one ragged launch for G aggregates of N signers (cold operand sets) against G single launches")
print(f"{'G x N':>10} {'ragged us':>10} {'frac':>6} {'GB/s':>8} | {'G launches us':>13} {'frac':>6}")
for G, N in ((64, 64), (64, 16), (16, 64), (128, 32), (8, 128), (4, 256), (256, 16)):
    T = G * N
    row = T * l * d * 4
    nsets = max(2, int(600e6 // row) + 1)
    sets = []
    for k in range(nsets):
        sig = DB(ctx, row)
        ctx.fill_synthetic_dev(sig.ptr, T * l * d, 100 + k)
        small = [DB(ctx, T * d * 4) for _ in range(4)]
        for j, b in enumerate(small):
            ctx.fill_synthetic_dev(b.ptr, T * d, 1000 + 10 * k + j)
        sets.append((sig, small))
    part = DB(ctx, G * (l * d + d) * 8)
    off = np.arange(G + 1, dtype=np.uintp) * N

    def ragged(k):
        sig, (al, vl, vr, c) = sets[k % nsets]
        ctx.aggregate_target_partial_ragged_dev(sig.ptr, al.ptr, vl.ptr, vr.ptr, c.ptr, off, l, part.ptr, l * d + d, part.ptr + l * d * 8, l * d + d)

    def singles(k):
        sig, (al, vl, vr, c) = sets[k % nsets]
        for g in range(G):
            o = g * N
            ctx.aggregate_target_partial_batch_dev(sig.ptr + o * l * d * 4, al.ptr + o * d * 4, vl.ptr + o * d * 4, vr.ptr + o * d * 4, c.ptr + o * d * 4,
                                                   part.ptr + g * (l * d + d) * 8, l * d + d, part.ptr + (g * (l * d + d) + l * d) * 8, l * d + d, 1, N, l)
    res = []
    for fn, reps in ((ragged, 60), (singles, 6)):
        for k in range(3):
            fn(k)
        ctx.synchronize()
        ctx.timer_start()
        for k in range(reps):
            fn(k)
        res.append(ctx.timer_stop_ms() * 1e3 / reps)
    moved = T * (l + 4) * 4 * d
    print(f"{G:>4} x {N:<4} {res[0]:>10.1f} {moved / res[0] / 1e3 / 8000:>6.3f} {moved / res[0] / 1e3:>8.0f} | {res[1]:>13.1f} {moved / res[1] / 1e3 / 8000:>6.3f}")
    for sig, small in sets:
        sig.free()
        for b in small:
            b.free()
    part.free()

if args.kernel_only:
    bs.close()
    sys.exit(0)
print("\n(2) end to end (hashing included): G aggregate()+verify() calls of N signers each")
print(f"{'G x N':>10} {'queue ms':>9} {'sig/s':>10} | {'G BatchScheme calls ms':>22} {'sig/s':>10} | {'aggregate_many+verify_many ms':>30} {'sig/s':>10}")
for G, N in ((64, 64), (64, 16), (16, 64), (256, 16)):
    T = G * N
    seeds = [9000 + 3 * i for i in range(T)]
    msgs = [f"synthetic message {i:06d}" for i in range(T)]
    sk, vk = bs.keygen_batch(seeds)
    sig = bs.sign_batch(sk, vk, msgs)
    packed = [PackedMessages(msgs[g * N:(g + 1) * N]) for g in range(G)]
    with BatchQueue(params, workers=2, max_rows=max(4096, T), host_threads=16) as bq:
        best_q = 1e9
        for rep in range(4):
            t0 = time.perf_counter()
            ts = [bq.submit_aggregate_verify(vk[g * N:(g + 1) * N], packed[g], sig[g * N:(g + 1) * N]) for g in range(G)]
            outs = [bq.wait_aggregate(t) for t in ts]
            dt = time.perf_counter() - t0
            if rep:
                best_q = min(best_q, dt)
        assert all(v == (True, "") for _, v in outs)
    t0 = time.perf_counter()
    singles = [bs.aggregate_verify(vk[g * N:(g + 1) * N], msgs[g * N:(g + 1) * N], sig[g * N:(g + 1) * N]) for g in range(G)]
    t_s = time.perf_counter() - t0
    assert all(np.array_equal(a, o[0]) for (a, _), o in zip(singles, outs))
    t0 = time.perf_counter()
    am = bs.aggregate_many(vk, msgs, sig, [N] * G)
    vm = bs.verify_many(vk, msgs, am, [N] * G)
    t_m = time.perf_counter() - t0
    assert all(v == (True, "") for v in vm) and all(np.array_equal(am[g], outs[g][0]) for g in range(G))
    print(f"{G:>4} x {N:<4} {best_q * 1e3:>9.2f} {T / best_q:>10.0f} | {t_s * 1e3:>22.2f} {T / t_s:>10.0f} | {t_m * 1e3:>30.2f} {T / t_m:>10.0f}")
bs.close()
