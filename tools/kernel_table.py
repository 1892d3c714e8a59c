"""Cold, per-kernel timing of the scheme kernels at secpar 256 (--secpar 128: degree 64, rank 195) (SURVEY.md 8d byte counts).

Every launch reads an operand set that no launch has touched for >= 2 GiB of other traffic: the sets are carved out
of one 2.25 GiB input pool (and the outputs rotate through a second pool), so nothing is served by the 256 MB Infinity
Cache or an L2 from an earlier repetition -- a printed fraction is HBM evidence, never cache bandwidth (round 1 printed
120 % for pw_mul because a 261 MB working set was re-read every repetition).

Used by bench.py (`roofline.kernels`) and runnable on its own -- `rocprofv3 --kernel-trace --stats -- python3
tools/kernel_table.py` gives the profiler's per-kernel durations of the same launches (profiles/r02_*)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "fusion-cryptography_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0
POOL_BYTES = 9 << 28          # 2.25 GiB


def measure(ctx, P, stream_ptr=None, reps_target_ms=30.0, quick=False, only=None):
    """-> {kernel: {units, bytes_per_unit, avg_us, GB/s, frac, sets_cycled}} ; ctx: fusion_hip.Context on the device.
    Device memory comes from fz_malloc (no torch needed); timing from events on the context's stream."""
    import fusion_hip
    DB = fusion_hip.DeviceBuffer
    q, d, l = P["q"], P["d"], P["rank"]
    row = d * 4
    pool_in, pool_out = DB(ctx, POOL_BYTES), DB(ctx, POOL_BYTES)
    # small centred values for everything: contents do not change any kernel's work (no data-dependent branch or exit)
    ctx.fill_synthetic_dev(pool_in.ptr, POOL_BYTES // 4, 11)
    ctx.fill_synthetic_dev(pool_out.ptr, POOL_BYTES // 4, 12)
    A = DB(ctx, l * row)
    ctx.fill_synthetic_dev(A.ptr, l * d, 13)
    ctx.synchronize()
    out = {}

    def run(name, units, bytes_per_unit, in_bytes, out_bytes, launch, note=None):
        """launch(in_ptr, out_ptr): one launch on operand set (in_ptr .. in_ptr + in_bytes) -> (out_ptr ..)"""
        if only and only not in name:
            return
        in_step = (in_bytes + 4095) & ~4095
        out_step = (max(out_bytes, 16) + 4095) & ~4095
        nsets = max(1, min(POOL_BYTES // in_step, 4096))
        nout = max(1, min(POOL_BYTES // out_step, 4096))
        k = 0

        def one():
            nonlocal k
            launch(pool_in.ptr + (k % nsets) * in_step, pool_out.ptr + (k % nout) * out_step)
            k += 1
        t_end = time.perf_counter() + (0.01 if quick else 0.04)    # clock ramp: ~15-20 ms of load after idle
        while time.perf_counter() < t_end:
            for _ in range(3):
                one()
            ctx.synchronize()
        ctx.timer_start()
        one()
        est = max(ctx.timer_stop_ms(), 1e-3)
        reps = int(max(5, min(400, reps_target_ms / est)))
        if quick:
            reps = min(reps, 20)
        us = 1e30
        for _ in range(3):                     # best of three passes: a pass now and then is disturbed from outside
            ctx.timer_start()
            for _ in range(reps):
                one()
            us = min(us, ctx.timer_stop_ms() / reps * 1e3)
        gbs = units * bytes_per_unit / (us * 1e-6) / 1e9
        out[name] = {"units_per_launch": units, "bytes_per_unit": bytes_per_unit, "avg_us": round(us, 2),
                     "GB/s": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4), "sets_cycled": int(nsets),
                     "launches_timed": reps, "passes": 3}
        # the shader clock the chip holds under this kernel (a one-wave probe beside ~5 ms of queued launches, read over their second 1.5 ms): the
        # fp64-dense kernels run power-limited below the nominal 2.4 GHz, the streaming ones do not
        if us >= 8.0 and not quick:
            for _ in range(int(max(4, min(1000, 5000.0 / us)))):        # the power limiter settles over milliseconds
                one()
            ctx.diag_shader_clock(1500)
            out[name]["shader_mhz"] = int(round(ctx.diag_shader_clock(1500)))
            ctx.synchronize()
        if note:
            out[name]["note"] = note

    S = 1024
    # keygen (fusion.py:363-370): coef [S][2][l][d] -> sk_hat same shape + vk [S][2][d]
    kb = S * 2 * l * row
    run("keygen_fused", S, (4 * l + 2) * row, kb, kb + S * 2 * row,
        lambda i, o: ctx.keygen_core_dev(A.ptr, i, o, o + kb, S, l))
    # the reference's SEEDED keygen (fusion.py:156-173, :338-362): one polynomial per key half -> the same sk_hat and vk arrays
    run("keygen_bcast_fused", S, (2 * l + 4) * row, S * 2 * row, kb + S * 2 * row,
        lambda i, o: ctx.keygen_core_bcast_dev(A.ptr, i, o, o + kb, S, l),
        note="one transform per key half, l row stores: 2 polynomials in, 2 l rows + 2 vk rows out")
    # sign (fusion.py:557): sk_hat [S][2][l][d], c_hat [S][d] -> sig [S][l][d]
    run("sign_kernel", S, (3 * l + 1) * row, kb + S * row, S * l * row,
        lambda i, o: ctx.sign_core_dev(i, i + kb, o, S, l))
    # aggregate (fusion.py:670-676): sig [N][l][d], alpha [N][d] -> [l][d]
    for N in (128, 256, 1024, 2048):
        sb = N * l * row
        run(f"aggregate N={N}" + (" (direct)" if N <= 128 else " (onepass)"), N, (l + 1) * row, sb + N * row, l * row,
            lambda i, o, N=N, sb=sb: ctx.aggregate_core_dev(i, i + sb, o, N, l),
            note="launches of <= 128 signers take aggregate_direct (no signer slices), larger ones aggregate_onepass")
    # aggregate + target partials in one pass, 4 aggregates of 256 signers (the bench's sign_verify step)
    G, per = 4, 256
    sb = G * per * l * row
    vb = G * per * row
    run("aggregate_onepass+target 4x256", G * per, (l + 5) * row, sb + 4 * vb, G * (l + 1) * d * 8,
        lambda i, o: ctx.aggregate_target_partial_batch_dev(i, i + sb, i + sb + vb, i + sb + 2 * vb, i + sb + 3 * vb, o, l * d,
                                                            o + G * l * d * 8, d, G, per, l),
        note="reads sigma + alpha + vkL + vkR + c_hat: (l + 5) rows per signer")
    # the same at 8 ranks (BASELINE configs[3]: 1024 signers over 8 GPUs): 8 aggregates of 128 local signers
    G, per = 8, 128
    sb = G * per * l * row
    vb = G * per * row
    run("aggregate_onepass+target 8x128", G * per, (l + 5) * row, sb + 4 * vb, G * (l + 1) * d * 8,
        lambda i, o: ctx.aggregate_target_partial_batch_dev(i, i + sb, i + sb + vb, i + sb + 2 * vb, i + sb + 3 * vb, o, l * d,
                                                            o + G * l * d * 8, d, G, per, l),
        note="one rank's launch of the sign_verify step at world 8")
    # sign + aggregate + target partials in ONE pass (fz_sign_aggregate_target_partial_batch, round 4): the signatures are written
    # and never read back -- moved bytes per signature: 2l (key halves) + c + alpha + vkL + vkR read, l written
    for G, per in ((4, 256), (8, 128)):
        kb2 = G * per * 2 * l * row
        vb = G * per * row
        ob = G * per * l * row
        run(f"sign+aggregate_onepass+target {G}x{per}", G * per, (3 * l + 4) * row, kb2 + 4 * vb, ob + G * (l + 1) * d * 8,
            lambda i, o, G=G, per=per, kb2=kb2, vb=vb, ob=ob: ctx.sign_aggregate_target_partial_batch_dev(
                i, i + kb2, i + kb2 + vb, i + kb2 + 2 * vb, i + kb2 + 3 * vb, o, o + ob, l * d, o + ob + G * l * d * 8, d, G, per, l),
            note="bytes MOVED per signature ((3l + 4) rows); the two launches it replaces move (3l + 1) + (l + 5)")
    # verify from int32 aggregates (fusion.py:690-727): sig [G][l][d] + target [G][d] -> verdict codes
    for G in (1, 64, 1024, 8192):
        vb = G * l * row
        run(f"verify_fused G={G}", G, (2 * l + 2) * row if G == 1 else (l + 2) * row, vb + G * row, G * 4,
            lambda i, o, G=G, vb=vb: ctx.verify_with_target_batch_async_dev(A.ptr, i, i + vb, G, l, P["beta_vf"], d, o),
            note="A (l rows) is shared by the aggregates of a launch: counted for G = 1 only")
    # matvec (matrices.py:115-131): S [batch][l][d] -> [batch][d]
    mb = 2 * S
    sizes = ([int(x) for x in sys.argv[sys.argv.index("--matvec-batches") + 1].split(",")] if "--matvec-batches" in sys.argv
             else [mb] if quick else [mb, 4 * mb])
    for nb in sizes:
        run("matvec" if nb == mb else f"matvec {nb} products", nb, (l + 1) * row, nb * l * row, nb * row,
            lambda i, o, nb=nb: ctx.matvec_dev(A.ptr, i, o, nb, l))
    # fused negacyclic product (ntt.py:380-484): f, g [n][d] -> [n][d]
    # (2^13 products: the radix-4 kernel at either degree; 2^17: the 16-per-lane kernel at degree 256 -- fz_launch_polymul_fused chooses)
    for n in ((1 << 17,) if quick else (1 << 13, 1 << 17)):
        run(f"polymul_fused n=2^{n.bit_length() - 1}", n, 3 * row, 2 * n * row, n * row,
            lambda i, o, n=n: ctx.poly_mul_dev(i, i + n * row, o, n))
    # pointwise product (polynomials.py:341-385)
    cnt = S * l * d
    run("pw_kernel<mul>", cnt, 12, 2 * cnt * 4, cnt * 4,
        lambda i, o: ctx.pw_dev(fusion_hip.OP_MUL, i, i + cnt * 4, o, cnt))
    # the practical ceiling: a plain copy (16 B per lane, non-temporal stores) of as many bytes, same cold cycling
    for mb in (64, 256):
        nbytes = mb << 20
        run(f"plain copy {mb} MiB -> {mb} MiB", nbytes // row, 2 * row, nbytes, nbytes,
            lambda i, o, nbytes=nbytes: ctx.diag_copy_dev(i, o, nbytes), note="not a scheme kernel: what HBM gives a read+write stream")
    if not quick:
        for logb in (12, 14, 16, 18):
            nb = 1 << logb
            run(f"ntt_forward B=2^{logb}", nb, 2 * row, nb * row, nb * row, lambda i, o, nb=nb: ctx.ntt_forward_dev(i, o, nb))
            run(f"ntt_inverse B=2^{logb}", nb, 2 * row, nb * row, nb * row, lambda i, o, nb=nb: ctx.ntt_inverse_dev(i, o, nb))
    if not quick:
        # bench.py's software-pipelined step: a forward job and an inverse job of 4096 rows in ONE fz_ntt_multi launch
        nb = 4096
        run("ntt_multi pair 2x4096 (fwd + inv)", 2 * nb, 2 * row, 2 * nb * row, 2 * nb * row,
            lambda i, o: ctx.ntt_multi_dev([(i, o, nb, False), (i + nb * row, o + nb * row, nb, True)]),
            note="HOST-PACED here (the job table is built per launch in Python): see tools/probes/ntt_ab.py / bench.py for the dense figure")
    for b in (pool_in, pool_out, A):
        b.free()
    return out


def main():
    import fusion_hip
    from fusion_hip.numa import pin_to_gpu_node
    pin_to_gpu_node(0)                      # host threads on the GPU's NUMA node (before the first HIP call)
    from oracle import oracle as O      # parameters only (tools/ is not product code)
    P = O.PARAMS[int(sys.argv[sys.argv.index("--secpar") + 1]) if "--secpar" in sys.argv else 256]
    ctx = fusion_hip.Context(P["q"], P["d"], P["root"], P["inv_root"])
    s = ctx.stream_create()
    ctx.set_stream(s)
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
    table = measure(ctx, P, quick="--quick" in sys.argv, only=only)
    for name, r in table.items():
        clk = f"  {r['shader_mhz']} MHz" if "shader_mhz" in r else ""
        print(f"{name:34s} {r['units_per_launch']:9d} units  {r['avg_us']:9.2f} us  {r['GB/s']:8.1f} GB/s algorithmic "
              f"({r['frac'] * 100:5.1f} % of 8 TB/s)  cold: {r['sets_cycled']} operand sets cycled{clk}")
    if "--json" in sys.argv:
        print(json.dumps(table))
    ctx.set_stream(0)
    ctx.stream_destroy(s)


if __name__ == "__main__":
    main()
