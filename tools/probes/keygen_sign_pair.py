"""keygen_core and sign_core, each alone and chained (the bench's keygen_sign leg), on 1024 distinct keys, operand sets rotating
(8 sets of 174 MB of sk_hat each): microseconds per launch / per pair from events on the kernels' stream, three passes.
For same-box A/B runs of the fused keygen kernel (FUSION_HIP_LIB selects the library).  usage: keygen_sign_pair.py [secpar] [pair]
(`pair`: the chained launches only -- under rocprofv3 --kernel-trace --stats, the per-kernel durations INSIDE the chain)"""
import os
import sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "fusion-cryptography_amd"))
sys.path.insert(0, ROOT)
import fusion_hip
from fusion_hip.numa import pin_to_gpu_node
pin_to_gpu_node(0)
from oracle import oracle as O      # parameters only (tools/ is not product code)

secpar = int(sys.argv[1]) if len(sys.argv) > 1 else 256
P = O.PARAMS[secpar]
q, d, l = P["q"], P["d"], P["rank"]
ctx = fusion_hip.Context(q, d, P["root"], P["inv_root"])
s = ctx.stream_create()
ctx.set_stream(s)
DB = fusion_hip.DeviceBuffer
S, NSETS, row = 1024, 8, d * 4
A = DB(ctx, l * row)
ctx.fill_synthetic_dev(A.ptr, l * d, 13)
sets = []
for i in range(NSETS):
    st = {"coef": DB(ctx, S * 2 * l * row), "sk": DB(ctx, S * 2 * l * row), "vk": DB(ctx, S * 2 * row), "c": DB(ctx, S * row), "sig": DB(ctx, S * l * row)}
    ctx.fill_synthetic_dev(st["coef"].ptr, S * 2 * l * d, 100 + i)
    ctx.fill_synthetic_dev(st["c"].ptr, S * d, 200 + i)
    sets.append(st)
ctx.synchronize()
kg = lambda i: ctx.keygen_core_dev(A.ptr, sets[i % NSETS]["coef"].ptr, sets[i % NSETS]["sk"].ptr, sets[i % NSETS]["vk"].ptr, S, l)
sg = lambda i: ctx.sign_core_dev(sets[i % NSETS]["sk"].ptr, sets[i % NSETS]["c"].ptr, sets[i % NSETS]["sig"].ptr, S, l)


def timed(fn, n=160):
    for i in range(40):
        fn(i)
    ctx.synchronize()
    best = 1e30
    for _ in range(3):
        ctx.timer_start()
        for i in range(n):
            fn(i)
        best = min(best, ctx.timer_stop_ms() / n * 1e3)
    return best


def pair(i):
    kg(i)
    sg(i)


alpha = DB(ctx, S * row)
ctx.fill_synthetic_dev(alpha.ptr, S * d, 300)
agg_out = DB(ctx, l * row)
ag = lambda i: ctx.aggregate_core_dev(sets[i % NSETS]["sig"].ptr, alpha.ptr, agg_out.ptr, S, l)


def sign_then_aggregate(i):
    sg(i)
    ag(i)


if len(sys.argv) > 2 and sys.argv[2] == "pair":
    print(f"secpar {secpar}: keygen + sign chained {timed(pair):7.2f} us per pair")
    sys.exit(0)
print(f"secpar {secpar}: aggregate alone {timed(ag):7.2f} us   sign + aggregate chained {timed(sign_then_aggregate):7.2f} us per pair")
print(f"secpar {secpar}: keygen alone {timed(kg):7.2f} us   sign alone {timed(sg):7.2f} us   keygen + sign chained {timed(pair):7.2f} us per pair "
      f"({fusion_hip.runtime_report().get('library', '')})")
