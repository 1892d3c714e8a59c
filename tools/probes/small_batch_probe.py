"""Per-launch time of the forward and inverse transforms at the bench's batch (default 4096 x degree 256),
launches issued back to back on one stream, HIP events on that stream.
usage: small_batch_probe.py [rows ...]   (library chosen by FUSION_HIP_LIB, schedule by FZ_NTT_KERNEL)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "fusion-cryptography_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import fusion_hip
from oracle import oracle as O

P = O.PARAMS[256]; d = P["d"]
ctx = fusion_hip.Context(O.PRIME, d, P["root"], P["inv_root"])
line = os.environ.get("PROBE_TAG", "")
for B in [int(v) for v in sys.argv[1:]] or [4096]:
    x = O.splitmix_centered(5, B * d).reshape(B, d)
    din = fusion_hip.DeviceBuffer.from_numpy(ctx, x)
    dout = fusion_hip.DeviceBuffer(ctx, x.nbytes)
    for name, fn in (("fwd", ctx.ntt_forward_dev), ("inv", ctx.ntt_inverse_dev)):
        best = 1e9
        for _ in range(5):
            for _ in range(20): fn(din.ptr, dout.ptr, B)
            ctx.synchronize()
            reps = 400
            ctx.timer_start()
            for _ in range(reps): fn(din.ptr, dout.ptr, B)
            best = min(best, ctx.timer_stop_ms() / reps)
        line += f" | B={B} {name} {best*1e3:6.2f}us"
    din.free(); dout.free()
print(line, flush=True)
ctx.close()
