#!/usr/bin/env python3
"""The reference's demonstration flow (misc/demo.py: setup -> N one-time keys -> N signatures -> one aggregate -> verify)
on the drop-in package: the same five calls on the same types, the algebra on the GPU.  Needs an MI355X and the built library
(`python -c "import __graft_entry__ as g; g.build()"`).

    python examples/demo.py [--secpar 128|256] [--signatures N] [--seed S] [--distinct-seeds]

Like the reference demo, every key is generated from the SAME seed unless --distinct-seeds is given (BASELINE config 1 is
this flow at secpar 128 with two signatures)."""
import argparse
import os
import random
import string
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fusion-cryptography_amd"))

from fusion.fusion import aggregate, fusion_setup, keygen, sign, verify  # noqa: E402


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--secpar", type=int, default=256, choices=(128, 256))
    ap.add_argument("--signatures", type=int, default=2)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--distinct-seeds", action="store_true", help="seed + 2 i for key i instead of one seed for all")
    args = ap.parse_args()
    n = args.signatures

    def step(label, fn):
        t0 = time.perf_counter()
        out = fn()
        print(f"{label} ({(time.perf_counter() - t0) * 1e3:.1f} ms)")
        return out

    params = step(f"Setup completed with security parameter {args.secpar} and seed {args.seed}.",
                  lambda: fusion_setup(args.secpar, args.seed))
    keys = step(f"Generated {n} key pairs.",
                lambda: [keygen(params, args.seed + (2 * i if args.distinct_seeds else 0)) for i in range(n)])
    alphabet = string.ascii_letters + string.digits
    messages = ["".join(random.choices(alphabet, k=20)) for _ in range(n)]
    signatures = step(f"Signed {n} messages.", lambda: [sign(params, k, m) for k, m in zip(keys, messages)])
    vks = [vk for _, vk in keys]
    agg = step("Aggregated the signatures.", lambda: aggregate(params, vks, messages, signatures))
    ok, why = step("Verified the aggregate signature.", lambda: verify(params, vks, messages, agg))
    print("Verification successful!" if ok else f"Verification failed! Reason: {why}")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
