#!/usr/bin/env python3
"""Dev-container check (needs the reference checkout, NOT part of the test suite): replays KAT files
written by tools/generate_kat_values.py on the GPU through the REFERENCE implementation and compares
every output string.  usage: PYTHONPATH=/root/reference python3 tools/check_kat_against_reference.py <kat_dir>"""
import csv
import re
import sys

csv.field_size_limit(1 << 30)
import fusion.fusion as F  # the reference (PYTHONPATH)

kat_dir = sys.argv[1]


def rows(name, secpar):
    with open(f"{kat_dir}/{name}_KAT_{secpar}.csv", newline="") as fh:
        return list(csv.reader(fh))


total = 0
for secpar in (128, 256):
    (inp, exp), = rows("fusion_setup", secpar)
    sp, seed_a = eval(inp)
    params = F.fusion_setup(sp, seed_a)
    assert str(params) == exp
    keys, msgs = [], []
    for inp, exp in rows("fusion_keygen", secpar):
        seed = int(re.search(r", (\d+)\)$", inp).group(1))
        k = F.keygen(params, seed)
        assert inp == str((params, seed)) and str(k) == exp
        keys.append(k)
    for (inp, exp), k in zip(rows("intermediate_hash_message_to_int", secpar), keys):
        msg = re.search(r", '([^']*)'\)$", inp).group(1)
        msgs.append(msg)
        assert str(F.hash_message_to_int(params, msg)) == exp
    pre = [F.hash_message_to_int(params, m) for m in msgs]
    for (inp, exp), k, p in zip(rows("intermediate_hash_vk_and_int_to_bytes_to_int", secpar), keys, pre):
        n = int(re.search(r", (\d+)\)$", inp).group(1))
        assert inp == str((params, k[1], p, n)) and str(F.hash_vk_and_int_to_bytes(params, k[1], p, n)) == exp
    challs = []
    for (inp, exp), k, m in zip(rows("intermediate_hash_ch", secpar), keys, msgs):
        c = F.hash_ch(params, k[1], m)
        assert inp == str((params, k[1], m)) and str(c) == exp
        challs.append(c)
    sigs = []
    for (inp, exp), k, m, p in zip(rows("fusion_sign", secpar), keys, msgs, pre):
        s = F.sign(params, k, m)
        assert inp == str((params, k, p)) and str(s) == exp
        sigs.append(s)
    vks = [k[1] for k in keys]
    (inp, exp), = rows("intermediate_hash_vks_and_ints_and_challs_to_bytes", secpar)
    assert inp == str((params, keys, pre, challs)) and str(F.hash_vks_and_ints_and_challs_to_bytes(params, keys, pre, challs)) == exp
    (inp, exp), = rows("intermediate_hash_ag", secpar)
    assert inp == str((params, keys, msgs)) and str(F.hash_ag(params, keys, msgs)) == exp
    (inp, exp), = rows("fusion_aggregate", secpar)
    agg = F.aggregate(params, vks, msgs, sigs)
    assert inp == str((params, vks, msgs, sigs)) and str(agg) == exp
    assert F.verify(params, vks, msgs, agg) == (True, "")
    total += 3 + 6 * len(keys)
    print(f"secpar {secpar}: every row of the 9 KAT files reproduces from the reference ({len(keys)} signatures)")
print("OK", total, "rows")
