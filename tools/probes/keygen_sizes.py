"""keygen_fused cold timing over key counts (workgroup-round quantisation check)"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "fusion-cryptography_amd")); sys.path.insert(0, R)
import fusion_hip
from fusion_hip.numa import pin_to_gpu_node
pin_to_gpu_node(0)          # host threads on the GPU's NUMA node (before the first HIP call)
from oracle import oracle as O
P = O.PARAMS[256]
q, d, l = P["q"], P["d"], P["rank"]
ctx = fusion_hip.Context(q, d, P["root"], P["inv_root"])
s = ctx.stream_create(); ctx.set_stream(s)
POOL = 9 << 28
DB = fusion_hip.DeviceBuffer
pin, pout, A = DB(ctx, POOL), DB(ctx, POOL), DB(ctx, l * d * 4)
ctx.fill_synthetic_dev(pin.ptr, POOL // 4, 5); ctx.fill_synthetic_dev(A.ptr, l * d, 6); ctx.synchronize()
row = d * 4
for S in [int(a) for a in sys.argv[1:]] or (512, 640, 768, 1024, 1280, 1536, 2048, 2560, 4096):
    kb = S * 2 * l * row
    step_in = (kb + 4095) & ~4095; step_out = (kb + S * 2 * row + 4095) & ~4095
    ns, no = POOL // step_in, POOL // step_out; k = [0]
    def one():
        i = pin.ptr + (k[0] % ns) * step_in; o = pout.ptr + (k[0] % no) * step_out; k[0] += 1
        ctx.keygen_core_dev(A.ptr, i, o, o + kb, S, l)
    te = time.perf_counter() + 0.04
    while time.perf_counter() < te:
        one(); ctx.synchronize()
    reps = max(5, min(200, int(20000 / (S * 0.08))))
    ctx.timer_start()
    for _ in range(reps): one()
    us = ctx.timer_stop_ms() / reps * 1e3
    print(f"S={S:5d} keys: {us:8.2f} us  {us / S * 1e3:7.2f} ns/key  {S * (4 * l + 2) * row / us / 8e3:5.1f} % of 8 TB/s")
