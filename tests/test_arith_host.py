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


@pytest.mark.parametrize("q", [2147465729, 5, 17, 65537, 12289, 2147483629, 3, 2147565569, 4294828033, 4294967291])
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
    assert L.t_fast(4294962689) == 0 and L.t_fast(4294967291) == 0       # 2^32 - 4607, 2^32 - 5: the 4-op multiply stays below 2^31
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


SRC5 = r'''
#include "fz_arith.h"
extern "C" {
double t_cent_i64(long long v, unsigned q) { return fz_cent_i64(v, fz_make_mod(q)); }
double t_fold(double x, unsigned q) { return fz_fold(x, fz_make_mod(q)); }
// the one-pass aggregation's inner step: x * (alpha >> 16) and x * (alpha & 0xffff) accumulated exactly over n terms,
// folded, returned as one residue
double t_split_accumulate(const int *x, const int *alpha, int n, unsigned q) {
    const FzMod m = fz_make_mod(q);
    double hi = 0.0, lo = 0.0;
    for (int i = 0; i < n; ++i) {
        hi = __builtin_fma((double)x[i], (double)(alpha[i] >> 16), hi);
        lo = __builtin_fma((double)x[i], (double)(alpha[i] & 0xffff), lo);
    }
    return fz_fold(lo, m) + fz_fold(hi * 65536.0, m);
}
}
'''


def test_exact_int64_centring_and_the_split_accumulation(tmp_path):
    """fz_cent_i64 (sums that crossed an all-reduce: exact for ANY int64), fz_fold and the hi/lo split of the one-pass
    aggregation at their operand bounds (16 terms of raw int32 operands: |sum| < 2^51)"""
    src = tmp_path / "t5.cpp"
    src.write_text(SRC5)
    so = tmp_path / "libt5.so"
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-I",
                           os.path.join(ROOT, "fusion-cryptography_amd", "csrc"), "-o", str(so), str(src)])
    L = ctypes.CDLL(str(so))
    L.t_cent_i64.restype = ctypes.c_double
    L.t_cent_i64.argtypes = [ctypes.c_longlong, ctypes.c_uint]
    L.t_fold.restype = ctypes.c_double
    L.t_fold.argtypes = [ctypes.c_double, ctypes.c_uint]
    L.t_split_accumulate.restype = ctypes.c_double
    I16 = ctypes.c_int * 16
    L.t_split_accumulate.argtypes = [I16, I16, ctypes.c_int, ctypes.c_uint]
    for q in (2147465729, 12289, 65537, 3, 5, 2147483629, 2147565569, 4294828033, 4294967291):        # (the last three: 2^31 <= q < 2^32, round 5)
        rng = random.Random(q + 5)
        vals = [0, 1, -1, 2**63 - 1, -2**63, 2**53, 2**53 + 1, -(2**53) - 1, 2**62 + 12345, q, -q, q * 2**31 + 7, (q // 2) * 2818 * 8]
        vals += [rng.randrange(-2**63, 2**63) for _ in range(5000)]
        for v in vals:
            assert int(L.t_cent_i64(v, q)) == cent(v, q), (q, v)
        for x in [rng.randrange(-2**67, 2**67) for _ in range(3000)] + [2**66, -2**66, 2**51 * 65536, 2**79, -2**79 + 2**30]:
            x = float(x)                       # an integer-valued double (at most 53 significant bits)
            r = L.t_fold(x, q)
            slack = abs(x) * 2.0**-51          # the quotient estimate is off by at most |x| / q * 2^-51
            assert r == int(r) and (int(r) - int(x)) % q == 0 and abs(r) <= q / 2 + slack + 1, (q, x, r)
        for _ in range(2000):
            ext = rng.random() < 0.2
            xs = [rng.choice([2**31 - 1, -2**31]) if ext else rng.randrange(-2**31, 2**31) for _ in range(16)]
            al = [rng.choice([2**31 - 1, -2**31]) if ext else rng.randrange(-2**31, 2**31) for _ in range(16)]
            r = L.t_split_accumulate(I16(*xs), I16(*al), 16, q)
            assert r == int(r) and (int(r) - sum(a * b for a, b in zip(xs, al))) % q == 0, (q, xs, al)
