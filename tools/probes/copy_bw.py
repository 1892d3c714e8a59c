"""practical HBM ceiling next to the transforms: fz_diag_copy (16 B per lane, non-temporal stores) over cold buffers"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "fusion-cryptography_amd")); sys.path.insert(0, R)
import fusion_hip
from fusion_hip.numa import pin_to_gpu_node
pin_to_gpu_node(0)          # host threads on the GPU's NUMA node (before the first HIP call)
from oracle import oracle as O
P = O.PARAMS[256]
ctx = fusion_hip.Context(P["q"], P["d"], P["root"], P["inv_root"])
s = ctx.stream_create(); ctx.set_stream(s)
POOL = 9 << 28
pin, pout = fusion_hip.DeviceBuffer(ctx, POOL), fusion_hip.DeviceBuffer(ctx, POOL)
ctx.fill_synthetic_dev(pin.ptr, POOL // 4, 5); ctx.synchronize()
for mb in (4, 16, 64, 256, 1024):
    n = mb << 20
    ns = POOL // n; k = [0]
    def one():
        o = (k[0] % ns) * n; k[0] += 1
        ctx.diag_copy_dev(pin.ptr + o, pout.ptr + o, n)
    te = time.perf_counter() + 0.05
    while time.perf_counter() < te:
        one(); ctx.synchronize()
    reps = max(5, min(300, int(2e9 // n)))
    ctx.timer_start()
    for _ in range(reps): one()
    us = ctx.timer_stop_ms() / reps * 1e3
    print(f"copy {mb:5d} MiB in + {mb:5d} MiB out: {us:9.2f} us  {2 * n / us / 1e3:8.1f} GB/s ({2 * n / us / 1e3 / 80:5.1f} % of 8 TB/s)")
