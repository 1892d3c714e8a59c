import sys, os, time
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "fusion-cryptography_amd")); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import fusion_hip
from fusion_hip.numa import pin_to_gpu_node
pin_to_gpu_node(0)          # host threads on the GPU's NUMA node (before the first HIP call)
from oracle import oracle as O
import numpy as np
P = O.PARAMS[256]
ctx = fusion_hip.Context(P["q"], P["d"], P["root"], P["inv_root"])
s = ctx.stream_create(); ctx.set_stream(s)
POOL = 9 << 28
pin, pout = fusion_hip.DeviceBuffer(ctx, POOL), fusion_hip.DeviceBuffer(ctx, POOL)
ctx.fill_synthetic_dev(pin.ptr, POOL // 4, 5); ctx.synchronize()
x = O.splitmix_centered(1, 300 * 256).reshape(300, 256)
assert np.array_equal(ctx.ntt_inverse(ctx.ntt_forward(x)), x)
def t(fn, nb, jobs=1):
    step = nb * 1024 * jobs; ns = POOL // step; k = [0]
    def one():
        i = pin.ptr + (k[0] % ns) * step; o = pout.ptr + (k[0] % ns) * step; k[0] += 1; fn(i, o)
    te = time.perf_counter() + 0.03
    while time.perf_counter() < te:
        for _ in range(3): one()
        ctx.synchronize()
    ctx.timer_start()
    for _ in range(300): one()
    return ctx.timer_stop_ms() / 300 * 1e3
for lb in (12, 13, 14, 15):
    nb = 1 << lb
    f = t(lambda i, o: ctx.ntt_forward_dev(i, o, nb), nb); v = t(lambda i, o: ctx.ntt_inverse_dev(i, o, nb), nb)
    print(f"B=2^{lb}: fwd {f:7.2f} us ({nb*2048/f/8e6*100:5.1f}%)  inv {v:7.2f} us ({nb*2048/v/8e6*100:5.1f}%)")
for jobs in (1, 2, 3, 4, 6, 8):
    nb = 4096
    f = t(lambda i, o: ctx.ntt_multi_dev([(i + j * nb * 1024, o + j * nb * 1024, nb, False) for j in range(jobs)]), nb, jobs)
    v = t(lambda i, o: ctx.ntt_multi_dev([(i + j * nb * 1024, o + j * nb * 1024, nb, True) for j in range(jobs)]), nb, jobs)
    print(f"multi {jobs}x4096 (cold): fwd {f:7.2f} us ({jobs*nb*2048/f/8e6*100:5.1f}%)  inv {v:7.2f} us ({jobs*nb*2048/v/8e6*100:5.1f}%)")
