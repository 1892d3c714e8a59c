import sys, os, time, ctypes
sys.path.insert(0, "fusion-cryptography_amd"); sys.path.insert(0, ".")
import numpy as np, torch, fusion_hip
from oracle import oracle as O
P = O.PARAMS[256]; q, d = P["q"], P["d"]; B = 4096
dev = torch.device("cuda", 0)
x = torch.from_numpy(O.splitmix_centered(1, B * d).reshape(B, d)).to(dev)
for nstreams in (1, 2, 3, 4):
    streams = [torch.cuda.Stream(dev) for _ in range(nstreams)]
    ctxs, bufs = [], []
    for s in streams:
        c = fusion_hip.Context(q, d, P["root"], P["inv_root"])
        c.set_stream(s.cuda_stream); ctxs.append(c)
        bufs.append((torch.empty_like(x), torch.empty_like(x)))
    lib = ctxs[0]._lib
    args = [(c._h, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), ctypes.c_void_p(z.data_ptr())) for c, (y, z) in zip(ctxs, bufs)]
    nB = ctypes.c_size_t(B)
    def run(K):
        for i in range(K):
            h, xp, yp, zp = args[i % nstreams]
            lib.fz_ntt_forward(h, xp, yp, nB); lib.fz_ntt_inverse(h, yp, zp, nB)
    run(40); torch.cuda.synchronize()
    K = 400
    t0 = time.perf_counter(); run(K); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ok = all(torch.equal(z, x) for _, z in bufs)
    print(f"streams={nstreams}: {dt/K*1e6:.2f} us/step  {2*B*K/dt/1e9:.3f} G NTT/s  roundtrip_ok={ok}", flush=True)
