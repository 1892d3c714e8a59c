"""Glue between the drop-in object API (lists of Python ints) and the HIP library.

Marshalling only: Python ints -> centred int32 (values are reduced mod q first, which never
changes a result because every library output depends on its inputs only through their
residues) and int32 -> Python ints.  All arithmetic happens in libfusion_hip.so; if the
library or a GPU is missing, the first compute call raises FusionHipError.

Two device paths, chosen by the parameters alone: q < 2^32 with transform lengths <= 4096 (every
parameter set the scheme defines) runs on the int32 kernels (fusion_hip.Context); anything else
the reference accepts below 2^63 -- moduli from 2^32, lengths beyond 4096 -- runs on the generic
int64 kernels (fusion_hip.wide.WideContext: exact, not fast).  Neither has a CPU route.
"""
import numpy as np

from fusion_hip import FusionHipError, get_context
from fusion_hip._lib import FZ_E_UNSUPPORTED

INT32_MIN, INT32_MAX = -(2 ** 31), 2 ** 31 - 1


NARROW_MODULUS = 2 ** 32       # exclusive: centred residues are int32 for every odd q below it (round 5; 2^31 before)
MAX_MODULUS = 2 ** 63          # exclusive: the generic path's rows are int64
MAX_DEGREE = 4096              # transform lengths of the int32 kernels (fz_internal.h kFzMaxDegree); longer ones: the generic path


def check_modulus(q):
    if not (isinstance(q, int) and 3 <= q < MAX_MODULUS and q % 2 == 1):
        raise FusionHipError(FZ_E_UNSUPPORTED,
                             f"modulus {q} is outside what the HIP kernels implement (odd, 3 <= q < 2^63: the generic path's "
                             "storage type is int64); there is no CPU fallback")


def is_wide(q):
    return q >= NARROW_MODULUS


def to_i32(rows, q):
    """list (or list of lists) of Python ints -> array of the ring's storage type (int32; int64 from q = 2^32 on); values
    outside the centred range are reduced mod q."""
    if is_wide(q):
        half = q // 2
        try:
            a = np.array(rows, dtype=np.int64)
            if a.size and (int(a.min()) < -half or int(a.max()) > half):
                raise OverflowError
            return a
        except OverflowError:
            def redw(v):
                y = v % q
                return y - q if y > half else y
            if rows and isinstance(rows[0], (list, tuple)):
                return np.array([[redw(v) for v in r] for r in rows], dtype=np.int64)
            return np.array([redw(v) for v in rows], dtype=np.int64)
    try:
        a = np.array(rows, dtype=np.int64)
        if a.size and (a.min() < INT32_MIN or a.max() > INT32_MAX):
            raise OverflowError
    except OverflowError:
        half = q // 2

        def red(v):
            y = v % q
            return y - q if y > half else y
        if rows and isinstance(rows[0], (list, tuple)):
            a = np.array([[red(v) for v in r] for r in rows], dtype=np.int64)
        else:
            a = np.array([red(v) for v in rows], dtype=np.int64)
    return a.astype(np.int32)


def stack_polys(polys, q):
    """polynomial objects of one length -> [len(polys)][d] int32.  Objects still backed by the row the library returned
    contribute it as it is; lists of Python ints are converted (all of them in one call when nothing is array-backed)."""
    if any(z._arr is not None for z in polys):
        return np.stack([z._i32() for z in polys]) if polys else np.empty((0, 0), np.int64 if is_wide(q) else np.int32)
    return to_i32([z._list for z in polys], q)


def _wide_tables(q, degree, root, inv_root):
    from fusion_hip.wide import bit_reversed_powers
    return tuple(bit_reversed_powers(root % q, q, degree)), tuple(bit_reversed_powers(inv_root % q, q, degree))


def ntt_ctx(q, degree, root, inv_root):
    check_modulus(q)
    if is_wide(q) or degree > MAX_DEGREE:
        from fusion_hip.wide import get_wide_context
        return get_wide_context(q, degree, *_wide_tables(q, degree, root, inv_root))
    return get_context(q, degree, root % q, inv_root % q)


def table_ctx(q, degree, fwd_table, inv_table):
    """a context whose twiddle tables are the caller's lists (cooley_tukey_ntt / gentleman_sande_intt use whatever table they
    are handed: ntt.py:274-290, :354-372)"""
    check_modulus(q)
    fwd, inv = tuple(int(v) % q for v in fwd_table), tuple(int(v) % q for v in inv_table)
    if is_wide(q) or degree > MAX_DEGREE:
        from fusion_hip.wide import get_wide_context
        return get_wide_context(q, degree, fwd, inv)
    from fusion_hip.context import get_table_context
    return get_table_context(q, degree, fwd, inv)


def neg_values(ctx, arr, q):
    """the reference's __neg__, -(x mod q) in [-(q-1), 0] (polynomials.py:155-163, :325-333): int32 rows from the device below
    2^31; from 2^31 on the values no longer fit the device's type, so the device computes the CENTRED negation (fz_pw_sub from
    zero) and the representative is shifted here -- c <= 0 stays, c > 0 becomes c - q: a change of representative of a value
    the device computed, not arithmetic of the path -- and the result is a list of Python ints."""
    if q < 2 ** 31 or is_wide(q):            # (the generic path's int64 holds -(q - 1) for every q it takes)
        return ctx.pw_neg(arr), None
    c = ctx.pw_sub(np.zeros_like(arr), arr)
    return None, [int(v) if v <= 0 else int(v) - q for v in c.tolist()]


def ring_ctx(q, degree):
    """pointwise-only context (no transform tables)"""
    check_modulus(q)
    if is_wide(q):
        from fusion_hip.wide import get_wide_context
        return get_wide_context(q, degree)
    return get_context(q, degree, 0, 0)
