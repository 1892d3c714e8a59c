#!/usr/bin/env python3
"""Where the drop-in OBJECT API (fusion.fusion: keygen / sign / aggregate / verify on the reference's own types, lists of Python
ints) spends its time per call: cProfile of each function at secpar 256, N = 16 signers.  Needs an MI355X."""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "fusion-cryptography_amd"))

import fusion.fusion as F  # noqa: E402


def prof(label, fn):
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    r = fn()
    pr.disable()
    dt = time.perf_counter() - t0
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18)
    print(f"==== {label}: {dt * 1e3:.2f} ms")
    print("\n".join(ln for ln in s.getvalue().splitlines()[4:] if ln.strip())[:3500])
    return r


def main():
    secpar = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    params = F.fusion_setup(secpar, 12345)
    seeds = [100 + 7 * i for i in range(n)]
    msgs = [f"message {i}" for i in range(n)]
    [F.keygen(params, 1)]                                  # warm-up: library load, context, A upload
    keys = prof(f"keygen x{n}", lambda: [F.keygen(params, s) for s in seeds])
    sigs = prof(f"sign x{n}", lambda: [F.sign(params, k, m) for k, m in zip(keys, msgs)])
    vks = [k[1] for k in keys]
    agg = prof(f"aggregate N={n}", lambda: F.aggregate(params, vks, msgs, sigs))
    ok = prof(f"verify N={n}", lambda: F.verify(params, vks, msgs, agg))
    assert ok == (True, ""), ok


if __name__ == "__main__":
    main()
