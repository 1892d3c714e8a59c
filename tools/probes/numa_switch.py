"""does moving the LAUNCHING thread between NUMA nodes after the GPU is initialised change the per-dispatch duration?"""
import os, sys, time, ctypes
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "fusion-cryptography_amd")); sys.path.insert(0, R)
import numpy as np
import fusion_hip
from fusion_hip import numa
from oracle import oracle as O
start = sys.argv[1] if len(sys.argv) > 1 else "none"
allowed = os.sched_getaffinity(0)
sets = {}
for d in sorted(os.listdir("/sys/devices/system/node")):
    if d.startswith("node") and d[4:].isdigit():
        c = numa._cpulist(open(f"/sys/devices/system/node/{d}/cpulist").read()) & allowed
        if c: sets[int(d[4:])] = c
if start != "none":
    os.sched_setaffinity(0, sets[int(start)])
print("gpu node", numa.gpu_numa_nodes(), "initialised on", start)
P = O.PARAMS[256]
ctx = fusion_hip.Context(P["q"], P["d"], P["root"], P["inv_root"])
s = ctx.stream_create(); ctx.set_stream(s)
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
busy(150)
for rep in range(3):
    for n, cpus in sets.items():
        os.sched_setaffinity(0, cpus)
        busy(20)
        ctx.profile_begin(800, 1)
        t0 = time.perf_counter()
        for _ in range(400): step()
        us, kind = ctx.profile_end_samples(800)
        wall = (time.perf_counter() - t0) / 800 * 1e6
        f = us[kind == 0]
        print(f"  launching thread on node {n}: fwd mean {f.mean():6.3f} median {np.median(f):6.3f}  host {wall:5.2f} us/launch")
