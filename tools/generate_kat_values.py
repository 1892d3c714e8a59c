#!/usr/bin/env python3
"""Counterpart of the reference's KATs/generate_KAT_values.py: writes the same
`str(inputs), str(outputs)` CSV rows, for the same functions and file names, using the drop-in package
running on the GPU.  Because every str() format is byte-identical to the reference's, the files can be
diffed against KAT files produced by a reference checkout with the same seeds -- including the 12 files
the reference's checkout is missing (.MISSING_LARGE_BLOBS).  tests/test_gpu_kat_tools.py compares every row this script
writes for one fixed (seed, sigs) with SHA-256 digests of the rows the REFERENCE's functions produce in the same flow
(tests/golden/kat_flow.json, made by tests/golden/gen_golden.py).

usage: generate_kat_values.py [out_dir] [--seed S] [--sigs N]     (needs an MI355X)
"""
import argparse
import csv
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fusion-cryptography_amd"))

from fusion.fusion import (_challenge_bytes_needed, aggregate, fusion_setup, hash_ag, hash_ch,  # noqa: E402
                           hash_message_to_int, hash_vk_and_int_to_bytes,
                           hash_vks_and_ints_and_challs_to_bytes, keygen, sign, verify)


def row(path, inputs, outputs):
    with open(path, "a", newline="") as fh:
        csv.writer(fh).writerow([str(inputs), str(outputs)])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out_dir", nargs="?", default="KAT_values")
    ap.add_argument("--seed", type=int, default=None, help="seed of the seed generator (default: entropy, like the reference)")
    ap.add_argument("--sigs", type=int, default=10)
    ap.add_argument("--batch", action="store_true",
                    help="keys, challenges, signatures and the aggregate come from the ARRAY API (BatchScheme: the MT19937 "
                         "sampler, the message pre-hash and the challenge pipeline run on the device) and are only printed "
                         "through the object classes; the hash rows still come from the object API")
    args = ap.parse_args()
    os.makedirs(args.out_dir, exist_ok=True)
    rng = random.Random(args.seed)

    def f(name, secpar):
        return os.path.join(args.out_dir, f"{name}_KAT_{secpar}.csv")
    for secpar in (128, 256):
        seed_a = rng.randint(0, 2**32 - 1)
        params = fusion_setup(secpar, seed_a)
        row(f("fusion_setup", secpar), (secpar, seed_a), params)
        seeds, msgs, keys, pre, challs, sigs = [], [], [], [], [], []
        n = _challenge_bytes_needed(params)
        if args.batch:
            import fusion.fusion as F
            from fusion_hip.scheme import BatchScheme, signature_to_object, sk_to_object, vk_to_object
            bs = BatchScheme(params)
            b_seeds = [rng.randint(0, 2**32 - 1) for _ in range(args.sigs)]
            b_msgs = [str(i) for i in range(args.sigs)]
            b_sk, b_vk = bs.keygen_batch(b_seeds)                     # device sampler + keygen_core_bcast
            assert bs.device_sampler and bs.device_hash
            b_chat, _ = bs.challenges(b_vk, b_msgs)                    # device pre-hash + challenge pipeline
            b_sig = bs.sign_batch(b_sk, b_vk, b_msgs)
            assert bs.device_hash, "the device challenge pipeline fell back to the host"
        for i in range(args.sigs):
            seeds.append(b_seeds[i] if args.batch else rng.randint(0, 2**32 - 1))
            msgs.append(str(i))
            keys.append((sk_to_object(params, seeds[i], b_sk[i]), vk_to_object(params, b_vk[i])) if args.batch
                        else keygen(params, seeds[i]))
            row(f("fusion_keygen", secpar), (params, seeds[i]), keys[-1])
            vk = keys[-1][1]
            pre.append(hash_message_to_int(params, msgs[i]))
            row(f("intermediate_hash_message_to_int", secpar), (params, msgs[i]), pre[-1])
            row(f("intermediate_hash_vk_and_int_to_bytes_to_int", secpar), (params, vk, pre[i], n),
                hash_vk_and_int_to_bytes(params, vk, pre[i], n))
            challs.append(F.SignatureChallenge(c_hat=F._ntt_poly(params, b_chat[i].tolist())) if args.batch
                          else hash_ch(params, vk, msgs[i]))
            row(f("intermediate_hash_ch", secpar), (params, vk, msgs[i]), challs[-1])
            sigs.append(signature_to_object(params, b_sig[i]) if args.batch else sign(params, keys[i], msgs[i]))
            row(f("fusion_sign", secpar), (params, keys[i], pre[i]), sigs[-1])
        vks = [k[1] for k in keys]
        # like the reference's script, these two rows hash the (signing key, verification key) TUPLES
        # (KATs/generate_KAT_values.py:115, :127 pass `otks`); aggregate and verify get the verification keys
        row(f("intermediate_hash_vks_and_ints_and_challs_to_bytes", secpar), (params, keys, pre, challs),
            hash_vks_and_ints_and_challs_to_bytes(params, keys, pre, challs))
        row(f("intermediate_hash_ag", secpar), (params, keys, msgs), hash_ag(params, keys, msgs))
        if args.batch:
            b_agg = bs.aggregate(b_vk, b_msgs, b_sig)
            agg = signature_to_object(params, b_agg)
            ok_b, why_b = bs.verify(b_vk, b_msgs, b_agg)
            assert ok_b, why_b
        else:
            agg = aggregate(params, vks, msgs, sigs)
        row(f("fusion_aggregate", secpar), (params, vks, msgs, sigs), agg)
        ok, why = verify(params, vks, msgs, agg)
        assert ok, why                      # (the reference asserts on the tuple, which is always truthy)
        print(f"secpar {secpar}: {args.sigs} signatures, aggregate verifies")


if __name__ == "__main__":
    main()
