#!/usr/bin/env python3
"""The shader clock the chip holds under each of the library's kernels (fz_diag_shader_clock: a one-wave probe on a private
stream beside the kernel's launches).  The fp64-dense kernels run power-limited below the nominal 2.4 GHz: a vector-issue
roofline has to be priced at the clock measured here.  Needs an MI355X.   usage: clock_under_load.py [--secpar 128|256]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "fusion-cryptography_amd"))
sys.path.insert(0, ROOT)

import fusion_hip  # noqa: E402
from fusion_hip.numa import pin_to_gpu_node  # noqa: E402
from oracle import oracle as O  # noqa: E402   (parameters only: tools/ is not product code)


def main():
    pin_to_gpu_node(0)
    P = O.PARAMS[int(sys.argv[sys.argv.index("--secpar") + 1]) if "--secpar" in sys.argv else 256]
    q, d, l = P["q"], P["d"], P["rank"]
    ctx = fusion_hip.Context(q, d, P["root"], P["inv_root"])
    s = ctx.stream_create()
    ctx.set_stream(s)
    DB = fusion_hip.DeviceBuffer
    row = d * 4
    G, K, NB = 8192, 1024, 1 << 18
    big = max(G * l * row, 2 * K * l * row, NB * row)
    src, dst = DB(ctx, big), DB(ctx, big)
    A, T, V = DB(ctx, l * row), DB(ctx, G * row), DB(ctx, G * 4)
    ctx.fill_synthetic_dev(src.ptr, big // 4, 3)
    ctx.fill_synthetic_dev(A.ptr, l * d, 4)
    ctx.fill_synthetic_dev(T.ptr, G * d, 5)
    ctx.synchronize()
    cases = [
        ("idle (nothing queued)", None),
        ("plain copy 256 MiB", lambda: ctx.diag_copy_dev(src.ptr, dst.ptr, 1 << 28)),
        (f"ntt_forward {NB} rows", lambda: ctx.ntt_forward_dev(src.ptr, dst.ptr, NB)),
        (f"ntt_inverse {NB} rows", lambda: ctx.ntt_inverse_dev(src.ptr, dst.ptr, NB)),
        (f"keygen_core {K} keys", lambda: ctx.keygen_core_dev(A.ptr, src.ptr, dst.ptr, T.ptr, K, l)),
        (f"sign_core {K} signatures", lambda: ctx.sign_core_dev(src.ptr, T.ptr, dst.ptr, K, l)),
        (f"aggregate_core {2048} signers", lambda: ctx.aggregate_core_dev(src.ptr, T.ptr, dst.ptr, 2048, l)),
        (f"verify (fused) {G} aggregates", lambda: ctx.verify_with_target_batch_async_dev(A.ptr, src.ptr, T.ptr, G, l, P["beta_vf"], d, V.ptr)),
        (f"matvec {2048} products", lambda: ctx.matvec_dev(A.ptr, src.ptr, dst.ptr, 2048, l)),
        (f"poly_mul {1 << 16} products", lambda: ctx.poly_mul_dev(src.ptr, src.ptr + (1 << 16) * row, dst.ptr, 1 << 16)),
    ]
    print(f"# shader clock while each kernel runs back to back (degree {d}, rank {l}); nominal 2400 MHz")
    for name, fn in cases:
        if fn is None:
            print(f"{name:36s} {ctx.diag_shader_clock(300):7.0f} MHz")
            continue
        t_end = time.perf_counter() + 0.05                      # reach the steady clock first
        while time.perf_counter() < t_end:
            for _ in range(4):
                fn()
            ctx.synchronize()
        t0 = time.perf_counter()
        fn()
        ctx.synchronize()
        one = max(time.perf_counter() - t0, 5e-6)
        n = max(4, int(4e-3 / one))                            # ~4 ms of launches queued, the probe watches 1 ms inside them
        for _ in range(n):
            fn()
        mhz = ctx.diag_shader_clock(1000)
        ctx.synchronize()
        print(f"{name:36s} {mhz:7.0f} MHz")
    ctx.set_stream(0)
    ctx.stream_destroy(s)


if __name__ == "__main__":
    main()
