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


SRC4 = r'''
#include "fz_arith.h"
extern "C" {
double t_mulmod4(double a, double w, unsigned q) { FzMod m = fz_make_mod(q); return fz_mulmod4(a, w, w * m.kq, m); }
int t_fast(unsigned q) { return fz_make_mod(q).fast; }
double t_kappa_times_K(unsigned q) { FzMod m = fz_make_mod(q); return m.kappa * m.K; }
}
'''


def test_mulmod4_exact_at_its_bounds(tmp_path):
    src = tmp_path / "t4.cpp"
    src.write_text(SRC4)
    so = tmp_path / "libt4.so"
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-I",
                           os.path.join(ROOT, "fusion-cryptography_amd", "csrc"), "-o", str(so), str(src)])
    L = ctypes.CDLL(str(so))
    L.t_mulmod4.restype = ctypes.c_double
    L.t_mulmod4.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_uint]
    L.t_fast.argtypes = [ctypes.c_uint]
    L.t_kappa_times_K.restype = ctypes.c_double
    L.t_kappa_times_K.argtypes = [ctypes.c_uint]
    assert L.t_fast(2147465729) == 1 and L.t_fast(12289) == 1 and L.t_fast(65537) == 0 and L.t_fast(40961) == 1
    assert L.t_kappa_times_K(2147465729) == 17919.0
    for q in (2147465729, 12289, 7681, 257, 97, 17, 5, 3, 40961, 2147483647 - 32766 * 0 - 18):
        if not L.t_fast(q):
            continue
        rng = random.Random(q + 1)
        A = [0, 1, -1, 2**38, -2**38, 2**38 - 1, 2**31, -2**31, q, q // 2, -(q // 2), 2**37 + 12345]
        A += [rng.randrange(-2**38, 2**38 + 1) for _ in range(6000)]
        W = [0, 1, q - 1, q // 2, q // 2 + 1] + [rng.randrange(q) for _ in range(12)]
        for a in A:
            for w in W:
                r = L.t_mulmod4(float(a), float(w), q)
                assert r == int(r), (q, a, w)
                assert (int(r) - a * w) % q == 0, (q, a, w)
                assert abs(r) <= q / 2 + q * 2.0**-12 + 1, (q, a, w, r)
