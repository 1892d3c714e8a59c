"""Sweep of the one-pass aggregation's launch shape (FZ_AGG_WAVES x FZ_AGG_SLICES) on cold operands; knobs are read at
context creation, so every setting gets its own context.  Scratch tool for DESIGN.md section 5's numbers."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "fusion-cryptography_amd"), ROOT):
    sys.path.insert(0, p)
import fusion_hip
from fusion_hip.numa import pin_to_gpu_node
pin_to_gpu_node(0)          # host threads on the GPU's NUMA node (before the first HIP call)
from oracle import oracle as O

P = O.PARAMS[256]
q, d, l = P["q"], P["d"], P["rank"]
row = d * 4
POOL = 9 << 28


def bench(ctx, pool, out, N, groups=1):
    sb = groups * N * l * row
    step = (sb + groups * N * row + 4095) & ~4095
    nsets = POOL // step
    k = 0

    def one():
        nonlocal k
        i = pool.ptr + (k % nsets) * step
        if groups == 1:
            ctx.aggregate_core_dev(i, i + sb, out.ptr, N, l)
        else:
            ctx.aggregate_partial_batch_dev(i, i + sb, out.ptr, l * d, groups, N, l)
        k += 1
    t_end = time.perf_counter() + 0.03
    while time.perf_counter() < t_end:
        for _ in range(3):
            one()
        ctx.synchronize()
    reps = 200 if N * groups <= 1024 else 100
    ctx.timer_start()
    for _ in range(reps):
        one()
    return ctx.timer_stop_ms() / reps * 1e3


def main():
    base = fusion_hip.Context(q, d, P["root"], P["inv_root"])
    pool = fusion_hip.DeviceBuffer(base, POOL)
    out = fusion_hip.DeviceBuffer(base, 8 * l * d * 8)
    base.fill_synthetic_dev(pool.ptr, POOL // 4, 3)
    base.synchronize()
    configs = [("auto", {})] + [(f"w{w} s{s}", {"FZ_AGG_WAVES": str(w), "FZ_AGG_SLICES": str(s)})
                                for w in (4, 8) for s in (1, 2, 4, 6, 8, 12, 16, 24, 32, 48)]
    if "--twopass" in sys.argv:
        configs.append(("twopass", {"FZ_AGG_TWOPASS": "1"}))
    shapes = [(256, 1), (1024, 1), (2048, 1), (256, 4)]
    print("config        " + "".join(f"  N={n}x{g:<3d} us (frac)" for n, g in shapes))
    for name, env in configs:
        for k_, v in env.items():
            os.environ[k_] = v
        ctx = fusion_hip.Context(q, d, P["root"], P["inv_root"])
        for k_ in env:
            del os.environ[k_]
        s_ = ctx.stream_create()
        ctx.set_stream(s_)
        cells = []
        for n, g in shapes:
            if name.startswith("w") and int(name.split("s")[1]) * int(name[1]) > n:
                cells.append("        -        ")
                continue
            us = bench(ctx, pool, out, n, g)
            cells.append(f"  {us:7.2f} ({(l + 1) * row * n * g / (us * 1e-6) / 8e12 * 100:4.1f}%)")
        print(f"{name:14s}" + "".join(cells), flush=True)
        ctx.set_stream(0)
        ctx.stream_destroy(s_)
        ctx.close()


if __name__ == "__main__":
    main()
