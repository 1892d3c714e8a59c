"""tools/microbench/keccak_wave (one Keccak state per wave) against hashlib.shake_256, and the time of the challenge
pipeline's chain (47 absorbed + 61 squeezed blocks per signer).  Run on a GPU box from the repository root:
    python tools/probes/keccak_wave_check.py [N ...]
"""
import hashlib
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BIN = os.path.join(ROOT, "tools", "microbench", "build", "keccak_wave")
RATE = 136


def pad(msg: bytes, stride: int):
    nb = len(msg) // RATE + 1
    row = bytearray(stride)
    row[:len(msg)] = msg
    row[len(msg)] ^= 0x1F
    row[nb * RATE - 1] ^= 0x80
    return bytes(row), nb


def run(n: int, out_blocks: int = 61, reps: int = 10, lengths=None):
    rng = np.random.default_rng(n)
    max_len = 47 * RATE - 1
    stride = 48 * RATE
    if lengths is None:
        lengths = [max_len - int(rng.integers(0, 200)) for _ in range(n)]
    msgs = [rng.integers(0, 256, size=L, dtype=np.uint8).tobytes() for L in lengths]
    rows, nbs = zip(*(pad(m, stride) for m in msgs))
    with tempfile.TemporaryDirectory() as tmp:
        fin, fout = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
        with open(fin, "wb") as fh:
            fh.write(b"".join(rows))
            fh.write(struct.pack("<%di" % n, *nbs))
        out = subprocess.run([BIN, fin, fout, str(n), str(stride), str(out_blocks), str(reps)], capture_output=True, text=True, timeout=120)
        sys.stdout.write(out.stdout)
        if out.returncode:
            print("FAILED rc", out.returncode, out.stderr)
            return False
        got = open(fout, "rb").read()
    bad = 0
    for i, m in enumerate(msgs):
        want = hashlib.shake_256(m).digest(out_blocks * RATE)
        if got[i * out_blocks * RATE:(i + 1) * out_blocks * RATE] != want:
            bad += 1
            if bad <= 3:
                g = got[i * out_blocks * RATE:(i + 1) * out_blocks * RATE]
                first = next(k for k in range(len(want)) if g[k] != want[k])
                print(f"  row {i} (len {len(m)}): first difference at byte {first}: got {g[first:first + 8].hex()} want {want[first:first + 8].hex()}")
    print(f"N={n}: {n - bad} of {n} rows equal hashlib.shake_256 ({out_blocks} blocks squeezed)")
    return bad == 0


if __name__ == "__main__":
    ok = run(7, out_blocks=3, reps=1, lengths=[0, 1, 135, 136, 137, 271, 1000])
    for n in [int(a) for a in sys.argv[1:]] or [256, 1024, 2048, 4096]:
        ok = run(n) and ok
    sys.exit(0 if ok else 1)
