"""fz_arith.h (the exact fp64 modular arithmetic used by every kernel) compiled for the HOST with
g++ and hammered against Python integers at the operand bounds the kernels rely on."""
import ctypes
import os
import random
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
#include "fz_arith.h"
extern "C" {
double t_mulmod(double a, double b, double q) { FzMod m{q, 1.0 / q}; return fz_mulmod(a, b, m); }
double t_cent(double x, double q) { FzMod m{q, 1.0 / q}; return fz_cent(x, m); }
double t_cent_wide(double x, double q) { FzMod m{q, 1.0 / q}; return fz_cent_wide(x, m); }
double t_mulmod_cent(double a, double b, double q) { FzMod m{q, 1.0 / q}; return fz_mulmod_cent(a, b, m); }
}
'''


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    d = tmp_path_factory.mktemp("arith")
    src = d / "t.cpp"
    src.write_text(SRC)
    so = d / "libt.so"
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-I",
                           os.path.join(ROOT, "fusion-cryptography_amd", "csrc"), "-o", str(so), str(src)])
    L = ctypes.CDLL(str(so))
    for n, k in (("t_mulmod", 3), ("t_cent", 2), ("t_cent_wide", 2), ("t_mulmod_cent", 3)):
        getattr(L, n).restype = ctypes.c_double
        getattr(L, n).argtypes = [ctypes.c_double] * k
    return L


def cent(v, q):
    y = v % q
    return y - q if y > q // 2 else y


@pytest.mark.parametrize("q", [2147465729, 5, 17, 65537, 12289, 2147483629, 3])
def test_cent_and_mulmod(q, lib):
    rng = random.Random(q)
    half = (q - 1) // 2
    edge = [0, 1, -1, half, -half, half + 1, -half - 1, q, -q, q - 1, 2**31 - 1, -2**31, 2 * q, 5 * q + half, -5 * q - half]
    for x in edge + [rng.randrange(-2**34, 2**34) for _ in range(4000)]:
        assert int(lib.t_cent(float(x), float(q))) == cent(x, q)
    for x in [rng.randrange(-2**52, 2**52) for _ in range(4000)] + [2**52, -2**52 + 1, 2818 * half]:
        assert int(lib.t_cent_wide(float(x), float(q))) == cent(x, q)
    ops = edge + [rng.randrange(-2**31, 2**31) for _ in range(200)]
    for a in ops:
        for b in ops[::7]:
            if abs(a) >= 2**31 + 2**30 or abs(b) > 2**31:
                continue
            assert int(lib.t_mulmod_cent(float(a), float(b), float(q))) == cent(a * b, q)
    # lazy operands: |a| up to 2^39 (what eight un-reduced GS stages can build), twiddle in [0, q)
    for _ in range(4000):
        a, b = rng.randrange(-2**39, 2**39), rng.randrange(0, q)
        r = lib.t_mulmod(float(a), float(b), float(q))
        assert r == int(r) and (int(r) - a * b) % q == 0
        assert abs(r) <= q / 2 + q * 2.0**-10 + 1
