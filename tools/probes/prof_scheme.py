"""Minimal launcher for PMC / kernel-trace passes over the scheme kernels on COLD operands (one launch per operand set,
sets carved out of a 2.25 GiB pool).  usage: prof_scheme.py [reps] [manifest.json]     (secpar 256; N = 1024 signers / keys /
signatures; a third argument keeps only the launches whose name contains it).  The manifest lists what was launched, in order, with the algorithmic bytes of each launch: tools/pmc_summary.py
attributes counters to launches by that order, never by kernel name or grid."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "fusion-cryptography_amd"), ROOT):
    sys.path.insert(0, p)
import fusion_hip
from fusion_hip.numa import pin_to_gpu_node
pin_to_gpu_node(0)          # host threads on the GPU's NUMA node (before the first HIP call)
from oracle import oracle as O

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
manifest_path = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else None
only = sys.argv[3] if len(sys.argv) > 3 else None
manifest = []
P = O.PARAMS[256]
q, d, l = P["q"], P["d"], P["rank"]
row = d * 4
ctx = fusion_hip.Context(q, d, P["root"], P["inv_root"])
POOL = 9 << 28
pin, pout = fusion_hip.DeviceBuffer(ctx, POOL), fusion_hip.DeviceBuffer(ctx, POOL)
A = fusion_hip.DeviceBuffer(ctx, l * row)
ctx.fill_synthetic_dev(pin.ptr, POOL // 4, 21)
ctx.fill_synthetic_dev(pout.ptr, POOL // 4, 22)
ctx.fill_synthetic_dev(A.ptr, l * d, 23)
S = 1024
kb = S * 2 * l * row


def sets(step):
    step = (step + 4095) & ~4095
    n = POOL // step
    return [pin.ptr + k * step for k in range(n)], [pout.ptr + (k % max(1, POOL // step)) * step for k in range(n)]


# (name, kernel family, algorithmic bytes per launch, extras, input bytes, launch)
for name, family, alg, extra, in_bytes, fn in (
        ("keygen", "keygen_fused", S * (4 * l + 2) * row, {}, kb, lambda i, o: ctx.keygen_core_dev(A.ptr, i, o, o + kb, S, l)),
        ("sign", "sign_kernel", S * (3 * l + 1) * row, {}, kb + S * row, lambda i, o: ctx.sign_core_dev(i, i + kb, o, S, l)),
        ("aggregate1024", "aggregate_", S * (l + 1) * row + l * row, {"signers": 1024}, S * (l + 1) * row,
         lambda i, o: ctx.aggregate_core_dev(i, i + S * l * row, o, S, l)),
        ("aggregate256", "aggregate_", 256 * (l + 1) * row + l * row, {"signers": 256}, 256 * (l + 1) * row,
         lambda i, o: ctx.aggregate_core_dev(i, i + 256 * l * row, o, 256, l)),
        # signing + aggregation + target sums in ONE pass, 4 aggregates of 256 (fz_sign_aggregate_target_partial_batch: the bench's step)
        ("sign+aggregate 4x256", "aggregate_", S * (3 * l + 4) * row,
         {"signers": 1024, "what": "sign + aggregate + target sums in one pass, 4 x 256 signers: (3l + 4) rows moved per signature"}, kb + 4 * S * row,
         lambda i, o: ctx.sign_aggregate_target_partial_batch_dev(i, i + kb, i + kb + S * row, i + kb + 2 * S * row, i + kb + 3 * S * row, o,
                                                                  o + S * l * row, l * d, o + S * l * row + 4 * l * d * 8, d, 4, 256, l)),
        ("matvec", "matvec_", 2 * S * (l + 1) * row, {}, 2 * S * l * row, lambda i, o: ctx.matvec_dev(A.ptr, i, o, 2 * S, l)),
        ("pw_mul", "pw_kernel", S * l * d * 12, {}, 2 * S * l * row, lambda i, o: ctx.pw_dev(fusion_hip.OP_MUL, i, i + S * l * row, o, S * l * d)),
        ("verify64", "verify_fused", 64 * (l + 2) * row + l * row, {"aggregates": 64}, 64 * (l + 1) * row,
         lambda i, o: ctx.verify_with_target_batch_async_dev(A.ptr, i, i + 64 * l * row, 64, l, P["beta_vf"], d, o)),
        ("verify1024", "verify_fused", 1024 * (l + 2) * row + l * row, {"aggregates": 1024}, 1024 * (l + 1) * row,
         lambda i, o: ctx.verify_with_target_batch_async_dev(A.ptr, i, i + 1024 * l * row, 1024, l, P["beta_vf"], d, o)),
        ("verify8192", "verify_fused", 8192 * (l + 2) * row + l * row, {"aggregates": 8192}, 8192 * (l + 1) * row,
         lambda i, o: ctx.verify_with_target_batch_async_dev(A.ptr, i, i + 8192 * l * row, 8192, l, P["beta_vf"], d, o)),
        # the coefficient-domain product (ntt.py:380-484) in both fused forms: 2^13 products take the radix-4 kernel, 2^17 the 16-per-lane one
        ("polymul 2^13", "polymul_fused", (1 << 13) * 3 * row, {"products": 1 << 13}, 2 * (1 << 13) * row,
         lambda i, o: ctx.poly_mul_dev(i, i + (1 << 13) * row, o, 1 << 13)),
        ("polymul 2^17", "polymul16", (1 << 17) * 3 * row, {"products": 1 << 17}, 2 * (1 << 17) * row,
         lambda i, o: ctx.poly_mul_dev(i, i + (1 << 17) * row, o, 1 << 17))):
    if only and only not in name:
        continue
    ins, outs = sets(max(in_bytes, kb + S * 2 * row))
    for k in range(reps):
        fn(ins[k % len(ins)], outs[k % len(outs)])
    ctx.synchronize()
    manifest.append(dict({"name": name, "kernel": family, "bytes": alg, "launches": reps}, **extra))
if manifest_path:
    with open(manifest_path, "w") as fh:
        json.dump(manifest, fh, indent=1)
print("done", reps)
