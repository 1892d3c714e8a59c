"""distribution of per-dispatch durations (begin/end events) of the B=4096 transforms, several passes in one process"""
import os, sys, time, ctypes
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "fusion-cryptography_amd")); sys.path.insert(0, R)
import numpy as np
import fusion_hip
from fusion_hip.numa import pin_to_gpu_node
pin_to_gpu_node(0)          # host threads on the GPU's NUMA node (before the first HIP call)
from oracle import oracle as O
P = O.PARAMS[256]
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
ctx = fusion_hip.Context(P["q"], P["d"], P["root"], P["inv_root"])
if mode in ("torch_stream", "torch_import"):
    import torch
    torch.cuda.set_device(0)
    if mode == "torch_stream":
        ts = torch.cuda.Stream(torch.device("cuda", 0)); torch.cuda.set_stream(ts); ctx.set_stream(ts.cuda_stream)
    else:
        torch.zeros(8, device="cuda"); s = ctx.stream_create(); ctx.set_stream(s)
else:
    s = ctx.stream_create(); ctx.set_stream(s)
print("mode", mode)
B, d = 4096, 256
DB = fusion_hip.DeviceBuffer
x, y, z = DB(ctx, B * d * 4), DB(ctx, B * d * 4), DB(ctx, B * d * 4)
ctx.fill_synthetic_dev(x.ptr, B * d, 3); ctx.synchronize()
lib, h = ctx._lib, ctx._h
xp, yp, zp, nB = ctypes.c_void_p(x.ptr), ctypes.c_void_p(y.ptr), ctypes.c_void_p(z.ptr), ctypes.c_size_t(B)
def step():
    lib.fz_ntt_forward(h, xp, yp, nB); lib.fz_ntt_inverse(h, yp, zp, nB)
def busy(ms):
    te = time.perf_counter() + ms * 1e-3
    while time.perf_counter() < te:
        for _ in range(50): step()
        ctx.synchronize()
for p in range(4):
    busy(150 if p == 0 else 20)
    n = 400
    ctx.profile_begin(2 * n, 1)
    t0 = time.perf_counter()
    for _ in range(n): step()
    us, kind = ctx.profile_end_samples(2 * n)
    wall = (time.perf_counter() - t0) / (2 * n) * 1e6
    f = np.sort(us[kind == 0])
    print(f"pass {p}: fwd mean {f.mean():6.3f} median {np.median(f):6.3f} min {f[0]:6.3f} p10 {f[len(f)//10]:6.3f} p90 {f[9*len(f)//10]:6.3f} max {f[-1]:7.3f}  host {wall:5.2f} us/launch")
