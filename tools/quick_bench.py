"""Timing of both NTT schedules (FZ_NTT_KERNEL = 16 | 4) over batch sizes, device-resident buffers, HIP events
on the context's stream, oracle parity on a ragged batch first.
usage: quick_bench.py [tpb...]   (FZ_NTT_TPB values to sweep for the radix-4 kernels; default 0 = persistent)"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "fusion-cryptography_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import fusion_hip
from oracle import oracle as O

q = O.PRIME
orc = O.COracle()
variants = [16, 4]
secpars = tuple(int(v) for v in os.environ.get("QB_SECPAR", "256").split(","))
mults = [int(v) for v in sys.argv[1:]] or [0]
for secpar in secpars:
    P = O.PARAMS[secpar]; d = P["d"]
    for var, mult in [(v, m) for v in variants for m in mults]:
        os.environ["FZ_NTT_KERNEL"] = str(var)
        os.environ["FZ_NTT_TPB"] = str(mult)
        ctx = fusion_hip.Context(q, d, P["root"], P["inv_root"])
        xs = O.splitmix_centered(9, 1003 * d).reshape(1003, d)
        ok = np.array_equal(ctx.ntt_forward(xs), orc.ntt_forward(xs, q, P["root"])) and \
            np.array_equal(ctx.ntt_inverse(xs), orc.ntt_inverse(xs, q, P["inv_root"]))
        line = f"secpar={secpar} var={var} tpb={mult} parity={'OK' if ok else 'FAIL'}"
        for logB in (12, 13, 14, 15, 16, 18, 20):
            B = (1 << logB) * (256 // d)             # the same bytes per launch for every degree
            nbytes = B * d * 4
            # QB_ROTATE=1: cycle through enough buffer pairs (>= 2 GiB together) that no launch finds its input in the
            # 256 MB Infinity Cache or the L2s from an earlier repetition ("cold" numbers)
            pairs = max(1, -(-(2 << 30) // (2 * nbytes))) if os.environ.get("QB_ROTATE") else 1
            pairs = min(pairs, 64)
            bufs = [(fusion_hip.DeviceBuffer(ctx, nbytes), fusion_hip.DeviceBuffer(ctx, nbytes)) for _ in range(pairs)]
            for a, _ in bufs:
                ctx.fill_synthetic_dev(a.ptr, B * d, 5)
            for name, fn in (("f", ctx.ntt_forward_dev), ("i", ctx.ntt_inverse_dev)):
                t_end = time.perf_counter() + 0.04      # 40 ms of the same launches first (clock ramp after idle)
                k = 0
                while time.perf_counter() < t_end:
                    for _ in range(3):
                        fn(bufs[k % pairs][0].ptr, bufs[k % pairs][1].ptr, B); k += 1
                    ctx.synchronize()
                reps = 400 if logB <= 12 else (100 if logB <= 16 else 20)
                ctx.timer_start()
                for _ in range(reps):
                    fn(bufs[k % pairs][0].ptr, bufs[k % pairs][1].ptr, B); k += 1
                ms = ctx.timer_stop_ms() / reps
                gbs = 8 * d * B / (ms * 1e-3) / 1e9
                line += f" | {B}{name} {ms*1e3:8.2f}us {gbs/80:5.1f}%"
            for a, b in bufs:
                a.free(); b.free()
        print(line, flush=True)
        ctx.close()
