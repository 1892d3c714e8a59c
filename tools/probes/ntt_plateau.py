import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(ROOT, "fusion-cryptography_amd")); sys.path.insert(0, ROOT)
import numpy as np, fusion_hip
ctx = fusion_hip.Context(2147465729, 256, 3337519, pow(3337519, -1, 2147465729))
s = ctx.stream_create(); ctx.set_stream(s)
DB = fusion_hip.DeviceBuffer
for logb in (16, 18, 20):
    B = 1 << logb
    n = max(2, (1 << 21) // B * 1)   # rotate buffers: >= 2 GiB? keep modest
    n = min(max(2, (3 << 30) // (B * 1024 * 2)), 16)
    xs = [DB(ctx, B * 1024) for _ in range(n)]; ys = [DB(ctx, B * 1024) for _ in range(n)]
    for k, x in enumerate(xs): ctx.fill_synthetic_dev(x.ptr, B * 256, 7 + k)
    for inv in (False, True):
        fn = ctx.ntt_inverse_dev if inv else ctx.ntt_forward_dev
        for k in range(n): fn(xs[k].ptr, ys[k].ptr, B)
        ctx.synchronize()
        reps = max(8, 64 >> (logb - 16))
        ctx.timer_start()
        for k in range(reps): fn(xs[k % n].ptr, ys[k % n].ptr, B)
        us = ctx.timer_stop_ms() * 1e3 / reps
        print(f"2^{logb} {'inv' if inv else 'fwd'} {us:9.2f} us  {B * 2048 / us / 1e3 / 8000:.3f}", flush=True)
    for b in xs + ys: b.free()
