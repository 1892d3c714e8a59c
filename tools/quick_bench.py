"""Scratch timing of the NTT kernels (device-resident buffers, HIP events on the ctx stream)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "fusion-cryptography_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import fusion_hip
from oracle import oracle as O

q = O.PRIME
for secpar in (256, 128):
    P = O.PARAMS[secpar]; d = P["d"]
    ctx = fusion_hip.Context(q, d, P["root"], P["inv_root"])
    for logB in (12, 14, 16, 18, 20):
        B = 1 << logB
        x = O.splitmix_centered(5, B * d).reshape(B, d)
        din = fusion_hip.DeviceBuffer.from_numpy(ctx, x)
        dout = fusion_hip.DeviceBuffer(ctx, x.nbytes)
        for name, fn in (("fwd", ctx.ntt_forward_dev), ("inv", ctx.ntt_inverse_dev)):
            for _ in range(3): fn(din.ptr, dout.ptr, B)
            ctx.synchronize()
            reps = 20
            ctx.timer_start()
            for _ in range(reps): fn(din.ptr, dout.ptr, B)
            ms = ctx.timer_stop_ms() / reps
            gbs = 8 * d * B / (ms * 1e-3) / 1e9
            print(f"secpar={secpar} d={d} B=2^{logB} {name}: {ms*1e3:9.2f} us  {B/(ms*1e-3)/1e9:7.3f} G NTT/s  {gbs:8.1f} GB/s  ({gbs/8000*100:5.1f}% of 8 TB/s)", flush=True)
        din.free(); dout.free()
