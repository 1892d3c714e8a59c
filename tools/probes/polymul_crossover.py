"""fz_poly_mul (algebra/ntt.py:380-484) in its two fused forms -- the radix-4 kernel (polymul_fused, FZ_POLYMUL_FORM=1) and the one on
the 16-per-lane transforms (polymul16, FZ_POLYMUL_FORM=2) -- over batch sizes, on cold operands (sets rotate through 2.25 GiB pools),
degrees 256 and 64: where the second overtakes the first is kPolymul16MinCoefs in csrc/fz_ntt.hip.  HIP events on the kernels' stream.
Output: profiles/r06_polymul_crossover.txt"""
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "fusion-cryptography_amd"))
sys.path.insert(0, ROOT)
import fusion_hip
from fusion_hip.numa import pin_to_gpu_node
pin_to_gpu_node(0)
from oracle import oracle as O      # parameters only (tools/ is not product code)

POOL = 9 << 28


def timed(ctx, calls):
    n, k = len(calls), 0
    te = time.perf_counter() + 0.03
    while time.perf_counter() < te:
        for _ in range(3):
            calls[k % n]()
            k += 1
        ctx.synchronize()
    best = 1e30
    for _ in range(3):
        ctx.timer_start()
        for _ in range(200):
            calls[k % n]()
            k += 1
        best = min(best, ctx.timer_stop_ms() / 200 * 1e3)
    return best


for secpar in (256, 128):
    P = O.PARAMS[secpar]
    d = P["d"]
    ctxs = {}
    for form in (1, 2):
        os.environ["FZ_POLYMUL_FORM"] = str(form)
        ctxs[form] = fusion_hip.Context(P["q"], d, P["root"], P["inv_root"])
    del os.environ["FZ_POLYMUL_FORM"]
    pin, pout = fusion_hip.DeviceBuffer(ctxs[1], POOL), fusion_hip.DeviceBuffer(ctxs[1], POOL)
    ctxs[1].fill_synthetic_dev(pin.ptr, POOL // 4, 5)
    ctxs[1].synchronize()
    print(f"degree {d}: microseconds per launch (fraction of 8 TB/s at 12 d bytes per product)", flush=True)
    print(f"  {'products':>9s} {'radix-4':>18s} {'16 per lane':>18s}")
    for lb in range(8, 19):
        n = 1 << lb
        step = 2 * n * d * 4
        sets = max(1, min(POOL // step, 2048))
        osets = max(1, min(POOL // (n * d * 4), 2048))
        row = []
        for form in (1, 2):
            c = ctxs[form]
            calls = [lambda k=k, c=c: c.poly_mul_dev(pin.ptr + (k % sets) * step, pin.ptr + (k % sets) * step + n * d * 4,
                                                   pout.ptr + (k % osets) * n * d * 4, n) for k in range(max(sets, osets) if max(sets, osets) < 4096 else 4096)]
            us = timed(c, calls)
            row.append(f"{us:9.2f} ({n * 12 * d / us / 8e6 * 100:5.1f} %)")
        print(f"  {n:9d} {row[0]:>18s} {row[1]:>18s}", flush=True)
    pin.free(); pout.free()
    for c in ctxs.values():
        c.close()
