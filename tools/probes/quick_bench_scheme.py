"""Scratch timing of the fused scheme cores on device-resident synthetic data (secpar 256):
algorithmic bytes per unit from SURVEY.md 8d / DESIGN.md 5, HIP events on the context's stream."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "fusion-cryptography_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import fusion_hip
from oracle import oracle as O

P = O.PARAMS[256]
q, d, l = P["q"], P["d"], P["rank"]
ctx = fusion_hip.Context(q, d, P["root"], P["inv_root"])
DB = fusion_hip.DeviceBuffer
rng = np.random.default_rng(1)


def timeit(fn, reps=50):
    import time
    t_end = time.perf_counter() + 0.04      # 40 ms of the same launches first: the GPU ramps its clocks for ~15 ms after idle
    while time.perf_counter() < t_end:
        for _ in range(3):
            fn()
        ctx.synchronize()
    ctx.timer_start()
    for _ in range(reps):
        fn()
    return ctx.timer_stop_ms() / reps * 1e-3


S = 1024
A = DB.from_numpy(ctx, O.splitmix_centered(1, l * d).reshape(l, d))
coef = DB.from_numpy(ctx, rng.integers(-52, 53, size=(S, 2, l, d)).astype(np.int32))
sk = DB(ctx, S * 2 * l * d * 4)
vk = DB(ctx, S * 2 * d * 4)
t = timeit(lambda: ctx.keygen_core_dev(A.ptr, coef.ptr, sk.ptr, vk.ptr, S, l))
b = (4 * l + 2) * 4 * d
print(f"keygen_core   {S} keys: {t*1e6:9.1f} us  {S/t/1e6:7.3f} M keys/s  {b*S/t/1e9:7.1f} GB/s algorithmic ({b*S/t/8e12*100:.1f}% of 8 TB/s)")
c_hat = DB.from_numpy(ctx, O.splitmix_centered(2, S * d).reshape(S, d))
al_hat = DB.from_numpy(ctx, O.splitmix_centered(3, S * d).reshape(S, d))
sig = DB(ctx, S * l * d * 4)
t = timeit(lambda: ctx.sign_core_dev(sk.ptr, c_hat.ptr, sig.ptr, S, l))
b = (3 * l + 1) * 4 * d
print(f"sign_core     {S} sigs: {t*1e6:9.1f} us  {S/t/1e6:7.3f} M sigs/s  {b*S/t/1e9:7.1f} GB/s algorithmic ({b*S/t/8e12*100:.1f}% of 8 TB/s)")
out = DB(ctx, l * d * 4)
for N in (256, 1024):
    t = timeit(lambda: ctx.aggregate_core_dev(sig.ptr, al_hat.ptr, out.ptr, N, l))
    b = (l + 1) * 4 * d
    print(f"aggregate_core N={N}: {t*1e6:9.1f} us  {N/t/1e6:7.3f} M sigs/s  {b*N/t/1e9:7.1f} GB/s algorithmic ({b*N/t/8e12*100:.1f}% of 8 TB/s)")
vkL = DB.from_numpy(ctx, O.splitmix_centered(4, S * d).reshape(S, d))
vkR = DB.from_numpy(ctx, O.splitmix_centered(5, S * d).reshape(S, d))
for N in (256, 1024):
    t = timeit(lambda: ctx.verify_core_dev(A.ptr, out.ptr, vkL.ptr, vkR.ptr, c_hat.ptr, al_hat.ptr, N, l, P["beta_vf"], d), reps=20)
    print(f"verify_core   N={N}: {t*1e6:9.1f} us per call (incl. the verdict D2H)")
mv = DB(ctx, S * 2 * d * 4)
t = timeit(lambda: ctx.matvec_dev(A.ptr, sk.ptr, mv.ptr, 2 * S, l))
b = (l + 1) * 4 * d
print(f"matvec        {2*S} products: {t*1e6:9.1f} us  {b*2*S/t/1e9:7.1f} GB/s algorithmic ({b*2*S/t/8e12*100:.1f}% of 8 TB/s)")
n = S * l * d
t = timeit(lambda: ctx.pw_dev(fusion_hip.OP_MUL, sig.ptr, sig.ptr, sk.ptr, n))
print(f"pw_mul        {n} coefficients: {t*1e6:9.1f} us  {12*n/t/1e9:7.1f} GB/s algorithmic ({12*n/t/8e12*100:.1f}% of 8 TB/s)")
for logn in (12, 16, 18):
    n = 1 << logn
    f = DB.from_numpy(ctx, O.splitmix_centered(6, n * d).reshape(n, d))
    g = DB.from_numpy(ctx, O.splitmix_centered(7, n * d).reshape(n, d))
    o = DB(ctx, n * d * 4)
    t = timeit(lambda: ctx.poly_mul_dev(f.ptr, g.ptr, o.ptr, n), reps=20)
    os.environ["FZ_UNFUSED"] = "1"
    tu = timeit(lambda: ctx.poly_mul_dev(f.ptr, g.ptr, o.ptr, n), reps=20)
    del os.environ["FZ_UNFUSED"]
    print(f"poly_mul      {n} products: {t*1e6:9.1f} us  {12*d*n/t/1e9:7.1f} GB/s algorithmic ({12*d*n/t/8e12*100:.1f}% of 8 TB/s); "
          f"four launches instead: {tu*1e6:9.1f} us")
    for b in (f, g, o):
        b.free()
