#!/usr/bin/env python3
"""Host Keccak-f[1600] variants (csrc/fz_host.cpp: scalar, BMI2, AVX-512) on this machine's cores: absorb rate of ONE sponge
(what bounds hash_ag, fusion/fusion.py:632-652) per variant, one child process each (FZ_KECCAK is read at load time), and
which one the measured dispatch picks.  No GPU needed."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, time
sys.path.insert(0, os.path.join(%r, "fusion-cryptography_amd"))
from fusion_hip import hostpipe
data = os.urandom(1 << 24)
best = 1e9
for _ in range(9):
    t0 = time.perf_counter(); hostpipe.shake256(data, 32); best = min(best, time.perf_counter() - t0)
print("%%-8s -> runs %%-7s absorb %%.3f GB/s  (%%.0f ns per permutation, 136 bytes each)" %% (os.environ.get("FZ_KECCAK", "auto"), hostpipe.keccak_variant(),
      len(data) / best / 1e9, best / (len(data) / 136) * 1e9))
''' % ROOT
for v in ("scalar", "bmi2", "x64", "x64v", None):
    env = dict(os.environ)
    env.pop("FZ_KECCAK", None)
    if v:
        env["FZ_KECCAK"] = v
    sys.stdout.write(subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True).stdout)
