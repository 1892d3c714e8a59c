"""A/B of the two aggregation kernels on cold operands: aggregate_onepass (signer slices + shared accumulator words) against
aggregate_direct (no slices, 16-column tiles of R rows, direct write; R = 1 and 64-column tiles were measured with this tool in
round 4 and removed from the library), per (signers, aggregates) shape, secpar 256 (rank 83, degree 256;
--secpar 128: rank 195, degree 64).  FZ_AGG_DIRECT is read at context creation, so every setting gets its own context.
Reference arithmetic: fusion/fusion.py:670-676.  Output: profiles/r04_aggregate_direct_ab.txt"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "fusion-cryptography_amd"), ROOT):
    sys.path.insert(0, p)
import fusion_hip
from fusion_hip.numa import pin_to_gpu_node
pin_to_gpu_node(0)          # host threads on the GPU's NUMA node (before the first HIP call)
from oracle import oracle as O      # parameters only (tools/ is not product code)

P = O.PARAMS[int(sys.argv[sys.argv.index("--secpar") + 1]) if "--secpar" in sys.argv else 256]
q, d, l = P["q"], P["d"], P["rank"]
row = d * 4
POOL = 9 << 28


def bench(ctx, pool, out, N, groups, target):
    sb = groups * N * l * row
    vb = groups * N * row
    step = (sb + (4 if target else 1) * vb + 4095) & ~4095
    nsets = max(1, POOL // step)
    k = 0

    def one():
        nonlocal k
        i = pool.ptr + (k % nsets) * step
        if target:
            ctx.aggregate_target_partial_batch_dev(i, i + sb, i + sb + vb, i + sb + 2 * vb, i + sb + 3 * vb, out.ptr, l * d,
                                                   out.ptr + groups * l * d * 8, d, groups, N, l)
        elif groups == 1:
            ctx.aggregate_core_dev(i, i + sb, out.ptr, N, l)
        else:
            ctx.aggregate_partial_batch_dev(i, i + sb, out.ptr, l * d, groups, N, l)
        k += 1
    t_end = time.perf_counter() + 0.03
    while time.perf_counter() < t_end:
        for _ in range(3):
            one()
        ctx.synchronize()
    reps = 200 if N * groups <= 2048 else 60
    best = 1e30
    for _ in range(3):
        ctx.timer_start()
        for _ in range(reps):
            one()
        best = min(best, ctx.timer_stop_ms() / reps * 1e3)
    return best


def main():
    base = fusion_hip.Context(q, d, P["root"], P["inv_root"])
    pool = fusion_hip.DeviceBuffer(base, POOL)
    out = fusion_hip.DeviceBuffer(base, 64 * (l + 1) * d * 8)
    base.fill_synthetic_dev(pool.ptr, POOL // 4, 3)
    base.synchronize()
    configs = [("auto", {}), ("sliced (onepass)", {"FZ_AGG_DIRECT": "-1"})] + \
        [(f"direct R={r}", {"FZ_AGG_DIRECT": str(r)}) for r in (4, 2)]
    shapes = [(64, 1, False), (128, 1, False), (256, 1, False), (512, 1, False), (1024, 1, False), (2048, 1, False),
              (256, 4, True), (128, 8, True), (64, 16, True), (16, 64, True), (1024, 1, True), (1024, 4, True)]
    if "--small" in sys.argv:
        shapes = [(2, 1, False), (8, 1, False), (32, 1, False), (96, 1, False), (192, 1, False), (320, 1, False), (64, 2, False),
                  (128, 2, False), (192, 2, False), (64, 3, True), (128, 3, True), (32, 4, True), (64, 4, True), (128, 4, True), (32, 8, True)]
    print(f"# secpar {256 if d == 256 else 128}: rank {l}, degree {d}; cold operands (sets rotate through a {POOL >> 20} MiB pool); best of 3 passes")
    print(f"# algorithmic bytes per signer: (l + 1) rows ((l + 5) with the verification target in the same launch)")
    print("config            " + "".join(f"  {n}x{g}{'+t' if t else '  '}".ljust(19) for n, g, t in shapes))
    for name, env in configs:
        for k_, v in env.items():
            os.environ[k_] = v
        ctx = fusion_hip.Context(q, d, P["root"], P["inv_root"])
        for k_ in env:
            del os.environ[k_]
        s_ = ctx.stream_create()
        ctx.set_stream(s_)
        cells = []
        for n, g, t in shapes:
            us = bench(ctx, pool, out, n, g, t)
            cells.append(f"  {us:7.2f} ({(l + (5 if t else 1)) * row * n * g / (us * 1e-6) / 8e12 * 100:4.1f}%) ".ljust(19))
        print(f"{name:18s}" + "".join(cells), flush=True)
        ctx.set_stream(0)
        ctx.stream_destroy(s_)
        ctx.close()


if __name__ == "__main__":
    main()
