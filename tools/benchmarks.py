#!/usr/bin/env python3
"""Counterpart of the reference's benchmarks/benchmarks.py: wall time of the five
API functions for N in {2, 4, 8, 16, 32} signers, through (a) the drop-in object API and (b) the
array-backed BatchScheme; prints one JSON document.  Needs an MI355X.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fusion-cryptography_amd"))

import fusion.fusion as F  # noqa: E402
from fusion_hip.scheme import BatchScheme  # noqa: E402


def timed(fn, *a):
    t0 = time.perf_counter()
    r = fn(*a)
    return r, time.perf_counter() - t0


def main():
    secpars = [int(x) for x in sys.argv[1:]] or [128, 256]
    out = {}
    for secpar in secpars:
        res = {"object_api": {}, "batch_api": {}}
        params, t_setup = timed(F.fusion_setup, secpar, 12345)
        res["fusion_setup_s"] = t_setup
        bs = BatchScheme(params)
        for n in (2, 4, 8, 16, 32):
            seeds = [100 + 7 * i for i in range(n)]
            msgs = [f"message {i}" for i in range(n)]
            keys, t_kg = timed(lambda: [F.keygen(params, s) for s in seeds])
            sigs, t_sg = timed(lambda: [F.sign(params, k, m) for k, m in zip(keys, msgs)])
            vks = [k[1] for k in keys]
            agg, t_ag = timed(F.aggregate, params, vks, msgs, sigs)
            ok, t_vf = timed(F.verify, params, vks, msgs, agg)
            assert ok == (True, "")
            res["object_api"][n] = dict(keygen_s=t_kg, sign_s=t_sg, aggregate_s=t_ag, verify_s=t_vf)
            (sk, vk), t_kg = timed(bs.keygen_batch, seeds)
            sg, t_sg = timed(bs.sign_batch, sk, vk, msgs)
            ag, t_ag = timed(bs.aggregate, vk, msgs, sg)
            ok, t_vf = timed(bs.verify, vk, msgs, ag)
            assert ok == (True, "")
            res["batch_api"][n] = dict(keygen_s=t_kg, sign_s=t_sg, aggregate_s=t_ag, verify_s=t_vf)
        out[secpar] = res
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
