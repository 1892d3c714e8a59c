"""The generic-parameter path of the drop-in packages: any odd modulus 3 <= q < 2^63 and any power-of-two length, on int64 rows.

The int32 kernels (Context) cover q < 2^32 and transform lengths up to 4096 -- every parameter set the scheme defines.  The
reference itself takes any odd modulus and any power-of-two length (algebra/ntt.py:239-290, :345-377; algebra/polynomials.py),
so everything beyond goes through csrc/fz_wide.hip: exact 64-bit integer arithmetic on the device, written for correctness,
not speed.  WideContext has the host face of Context (numpy in, numpy out, the same method names), with int64 arrays.
There is no CPU route: without the library or a GPU every call raises FusionHipError."""
from ctypes import POINTER, c_int32, c_int64, c_uint64

import collections
import ctypes

import numpy as np

from ._lib import FZ_E_BADARG, FZ_E_UNSUPPORTED, FusionHipError, check, load_library

OP_MUL, OP_ADD, OP_SUB, OP_NEG = 0, 1, 2, 3
MAX_WIDE_MODULUS = 2 ** 63          # exclusive
_I64P = POINTER(c_int64)


def _as_i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _p(a):
    return a.ctypes.data_as(_I64P)


def bit_reversed_powers(root, q, n):
    """bit_reverse_copy([root^i mod q for i < n]) -- the table the reference's constructors build (polynomials.py:396-397)"""
    bits = n.bit_length() - 1
    powers, acc = [], 1
    for _ in range(n):
        powers.append(acc)
        acc = (acc * root) % q
    return [powers[int(format(i, f"0{bits}b")[::-1], 2)] if bits else powers[i] for i in range(n)]


class WideContext:
    def __init__(self, modulus, degree, fwd_table=None, inv_table=None, device=0):
        if not (isinstance(modulus, int) and 3 <= modulus < MAX_WIDE_MODULUS and modulus % 2 == 1):
            raise FusionHipError(FZ_E_UNSUPPORTED, f"modulus {modulus} is outside what the HIP kernels implement (odd, 3 <= q < 2^63: the "
                                                   "generic path's storage type is int64); there is no CPU fallback")
        self._lib = load_library()
        self.modulus, self.degree, self.device = modulus, int(degree), device
        self._tables = {}
        for inverse, tab in ((0, fwd_table), (1, inv_table)):
            if tab is not None:
                if len(tab) < self.degree:
                    raise IndexError("list index out of range")
                self._tables[inverse] = (c_uint64 * self.degree)(*[int(v) % modulus for v in tab[:self.degree]])
        self._n_inv = pow(self.degree, -1, modulus) if self._tables and np.gcd(self.degree, modulus) == 1 else 0

    # -- transforms ------------------------------------------------------------------------------------------------------
    def _rows(self, a):
        a = _as_i64(a)
        if a.size % self.degree:
            raise FusionHipError(FZ_E_BADARG, f"array of {a.size} values is not a whole number of degree-{self.degree} rows")
        return a, a.size // self.degree

    def _transform(self, x, inverse):
        if inverse not in self._tables:
            raise FusionHipError(FZ_E_UNSUPPORTED, "ring-only context (created without tables) has no transforms")
        a, rows = self._rows(x)
        out = np.empty_like(a)
        check(self._lib, self._lib.fz_wide_ntt_host(self.device, self.modulus, self.degree, self._tables[inverse], self._n_inv, inverse,
                                                    _p(a), _p(out), rows))
        return out

    def ntt_forward(self, x):
        """cooley_tukey_ntt of every row (ntt.py:216-291); returns a new array."""
        return self._transform(x, 0)

    def ntt_inverse(self, x):
        """gentleman_sande_intt of every row (ntt.py:294-377); returns a new array."""
        return self._transform(x, 1)

    def poly_mul(self, f, g):
        """INTT(NTT(f) * NTT(g)) for host rows [batch][degree] (or one row): ntt_poly_mult, ntt.py:380-484"""
        f, g = _as_i64(f), _as_i64(g)
        if f.shape != g.shape or f.shape[-1] != self.degree:
            raise FusionHipError(FZ_E_BADARG, f"shape mismatch {f.shape} vs {g.shape} (degree {self.degree})")
        return self.ntt_inverse(self.pw_mul(self.ntt_forward(f), self.ntt_forward(g))).reshape(f.shape)

    # -- pointwise -------------------------------------------------------------------------------------------------------
    def _pw(self, op, a, b):
        a = _as_i64(a)
        out = np.empty_like(a)
        if b is not None:
            b = _as_i64(b)
            if b.shape != a.shape:
                raise FusionHipError(FZ_E_BADARG, f"shape mismatch {a.shape} vs {b.shape}")
        check(self._lib, self._lib.fz_wide_pw_host(self.device, self.modulus, op, _p(a), _p(b) if b is not None else None, _p(out), a.size))
        return out

    def pw_mul(self, a, b):
        return self._pw(OP_MUL, a, b)

    def pw_add(self, a, b):
        return self._pw(OP_ADD, a, b)

    def pw_sub(self, a, b):
        return self._pw(OP_SUB, a, b)

    def pw_neg(self, a):
        """-(x mod q) in [-(q-1), 0], the reference's __neg__ (polynomials.py:155-163, :325-333)"""
        return self._pw(OP_NEG, a, None)

    def matvec(self, A, S):
        """A: [l][d]; S: [batch][l][d] -> [batch][d]."""
        A, S = _as_i64(A), _as_i64(S)
        l = A.shape[0]
        S3 = S.reshape(-1, l, self.degree)
        out = np.empty((S3.shape[0], self.degree), dtype=np.int64)
        check(self._lib, self._lib.fz_wide_matvec_host(self.device, self.modulus, self.degree, _p(A), _p(S3), _p(out), S3.shape[0], l))
        return out

    def norm_weight(self, coef):
        a, rows = self._rows(coef)
        mx = np.empty(rows, dtype=np.uint64)                # |INT64_MIN| = 2^63 does not fit the signed type
        wt = np.empty(rows, dtype=np.int32)
        check(self._lib, self._lib.fz_wide_norm_weight_host(self.device, _p(a), rows, self.degree, mx.ctypes.data_as(POINTER(ctypes.c_uint64)),
                                                            wt.ctypes.data_as(POINTER(c_int32))))
        return mx, wt


_WIDE_MAX = 8
_WIDE_CACHE = collections.OrderedDict()


def get_wide_context(modulus, degree, fwd_table=None, inv_table=None, device=0):
    """memoised (tables as tuples or None): the `_WIDE_MAX` most recently used (a context holds its tables as arrays)"""
    key = (modulus, degree, fwd_table, inv_table, device)
    ctx = _WIDE_CACHE.get(key)
    if ctx is not None:
        _WIDE_CACHE.move_to_end(key)
        return ctx
    ctx = _WIDE_CACHE[key] = WideContext(modulus, degree, fwd_table, inv_table, device)
    while len(_WIDE_CACHE) > _WIDE_MAX:
        _WIDE_CACHE.popitem(last=False)
    return ctx
