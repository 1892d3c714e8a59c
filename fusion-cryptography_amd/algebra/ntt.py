"""Drop-in for the reference's ``algebra.ntt`` (same names, arguments and error behaviour),
with the transforms executed by the MI355X kernels behind ``libfusion_hip.so``.

Reference surface mirrored (file:line in the reference checkout):
  is_odd_prime :17 · has_primitive_root_of_unity :36 · is_pow_two_geq_two :59 ·
  bit_reverse_copy :74 · cent :93 · is_root_of_unity :126 · is_primitive_root :153 ·
  find_primitive_root :184 · cooley_tukey_ntt :216 · gentleman_sande_intt :294 ·
  ntt_poly_mult :380 · ntt_poly_mult_half :487
The predicates and the scalar ``cent`` are host-side parameter plumbing (setup time, single
integers) and stay in Python; every operation on coefficient vectors goes to the GPU.
"""
from copy import deepcopy
from typing import Dict, List, Optional, Tuple

from . import _backend

CACHED_PRIMITIVE_ROOTS: Dict[Tuple[int, int], int] = {}
CACHED_IS_ODD_PRIME: Dict[int, bool] = {}
CACHED_HAS_PRIMITIVE_ROOT_OF_UNITY: Dict[Tuple[int, int], bool] = {}
CACHED_IS_POW_TWO_GEQ_TWO: Dict[int, bool] = {}
CACHED_IS_ROOT_OF_UNITY: Dict[Tuple[int, int, int], bool] = {}
CACHED_IS_PRIMITIVE_ROOT_OF_UNITY: Dict[Tuple[int, int, int], bool] = {}
CACHED_FIND_PRIMITIVE_ROOT: Dict[Tuple[int, int], Optional[int]] = {}


def is_odd_prime(val: int) -> bool:
    """True for ints >= 3 with no ODD divisor in [3, floor(sqrt(val))] -- exactly the reference's
    predicate (ntt.py:25-33), including its quirk that even numbers are never divided by 2."""
    hit = CACHED_IS_ODD_PRIME.get(val)
    if hit is None:
        hit = False
        if isinstance(val, int) and val >= 3:
            hit = True
            top = int(val ** 0.5)
            cand = 3
            while cand <= top:
                if val % cand == 0:
                    hit = False
                    break
                cand += 2
        CACHED_IS_ODD_PRIME[val] = hit
    return hit


def has_primitive_root_of_unity(modulus: int, root_order: int) -> bool:
    key = (modulus, root_order)
    if key not in CACHED_HAS_PRIMITIVE_ROOT_OF_UNITY:
        ok = (isinstance(modulus, int) and isinstance(root_order, int) and modulus >= 3 and root_order >= 2
              and (modulus - 1) % root_order == 0)
        CACHED_HAS_PRIMITIVE_ROOT_OF_UNITY[key] = ok
    return CACHED_HAS_PRIMITIVE_ROOT_OF_UNITY[key]


def is_pow_two_geq_two(val: int) -> bool:
    if val not in CACHED_IS_POW_TWO_GEQ_TWO:
        CACHED_IS_POW_TWO_GEQ_TWO[val] = isinstance(val, int) and val >= 2 and (val & (val - 1)) == 0
    return CACHED_IS_POW_TWO_GEQ_TWO[val]


def bit_reverse_copy(val: list):
    """New list with element i taken from the bit-reversed index (ntt.py:74-90)."""
    if not isinstance(val, list):
        raise ValueError("Input must be a list")
    n = len(val)
    bits = n.bit_length() - 1
    out = []
    for i in range(n):
        r = 0
        for b in range(bits):
            r |= ((i >> b) & 1) << (bits - 1 - b)
        out.append(deepcopy(val[r]))
    return out


def cent(val: int, modulus: int, halfmod: int, logmod: int) -> int:
    """Centred residue of one integer: val mod q moved into [-(q//2), q//2] (ntt.py:93-123).
    Scalar helper (host side); the vector form lives in the kernels."""
    if not (isinstance(val, int) and isinstance(modulus, int) and isinstance(halfmod, int)
            and isinstance(logmod, int)):
        raise TypeError("Input must be integers")
    if modulus < 2:
        raise ValueError("Modulus must be at least 2")
    if halfmod < 1:
        raise ValueError("Halfmod must be at least 1")
    if logmod < 1:
        raise ValueError("Logmod must be at least 1")
    y = val % modulus
    # the reference's branch-free select: subtract q exactly when y - halfmod - 1 >= 0
    return y - (1 + ((y - halfmod - 1) >> logmod)) * modulus


def is_root_of_unity(val: int, modulus: int, root_order: int) -> bool:
    key = (val, modulus, root_order)
    if key not in CACHED_IS_ROOT_OF_UNITY:
        ok = False
        if (isinstance(val, int) and isinstance(modulus, int) and isinstance(root_order, int)
                and modulus >= 2 and root_order >= 1):
            ok = pow(val, root_order, modulus) == 1
        CACHED_IS_ROOT_OF_UNITY[key] = ok
    return CACHED_IS_ROOT_OF_UNITY[key]


def is_primitive_root(val: int, modulus: int, root_order: int) -> bool:
    key = (val, modulus, root_order)
    if key not in CACHED_IS_PRIMITIVE_ROOT_OF_UNITY:
        ok = False
        if (isinstance(val, int) and isinstance(modulus, int) and isinstance(root_order, int)
                and modulus >= 2 and root_order >= 1):
            if is_root_of_unity(val=val, modulus=modulus, root_order=root_order):
                # no smaller positive exponent gives 1 (ntt.py:177-179); walk the powers once
                ok = True
                acc = val % modulus
                for _ in range(1, root_order):
                    if acc == 1 % modulus:
                        ok = False
                        break
                    acc = (acc * val) % modulus
        CACHED_IS_PRIMITIVE_ROOT_OF_UNITY[key] = ok
    return CACHED_IS_PRIMITIVE_ROOT_OF_UNITY[key]


def find_primitive_root(modulus: int, root_order: int) -> int:
    """Smallest r >= 2 that is a primitive root_order-th root of unity (linear search, ntt.py:203-207)."""
    key = (modulus, root_order)
    if key not in CACHED_FIND_PRIMITIVE_ROOT:
        CACHED_FIND_PRIMITIVE_ROOT[key] = None
        if (isinstance(modulus, int) and isinstance(root_order, int) and modulus >= 2 and root_order >= 1
                and has_primitive_root_of_unity(modulus=modulus, root_order=root_order)):
            r = 2
            while r < modulus and not is_primitive_root(val=r, modulus=modulus, root_order=root_order):
                r += 1
            if not is_primitive_root(val=r, modulus=modulus, root_order=root_order):
                raise RuntimeError(
                    f"No primitive root found with modulus={modulus}, root_order={root_order}.")
            CACHED_FIND_PRIMITIVE_ROOT[key] = r
    return CACHED_FIND_PRIMITIVE_ROOT[key]


# ---------------------------------------------------------------------------------------------
# transforms
# ---------------------------------------------------------------------------------------------
def _validate_transform_args(val, modulus, root_order, table, table_name):
    """Same checks in the same order as ntt.py:239-270 / :318-349."""
    if not isinstance(val, list):
        raise TypeError(f"val must be a list, but got {type(val)}")
    if not isinstance(modulus, int):
        raise TypeError(f"modulus must be an int, but got {type(modulus)}")
    if not isinstance(table, list):
        raise TypeError(f"{table_name} must be a list, but got {type(table)}")
    if not all(isinstance(v, int) for v in table):
        raise TypeError(f"{table_name} must be a list of ints, but got {type(table)}")
    if not isinstance(root_order, int):
        raise TypeError(f"root_order must be an int, but got {type(root_order)}")
    if not all(isinstance(v, int) for v in val):
        raise TypeError(f"val must be a list of ints, but got {type(val)}")
    if not is_odd_prime(val=modulus):
        raise ValueError(f"modulus={modulus} must be an odd prime")
    if not has_primitive_root_of_unity(modulus=modulus, root_order=root_order):
        raise ValueError(f"modulus={modulus} does not have a primitive root of order root_order={root_order}")
    if not is_pow_two_geq_two(val=len(val)):
        raise ValueError(f"len(val)={len(val)} must be a power of 2 greater than 1")
    if root_order != 2 * len(val) and root_order != len(val):
        raise ValueError(f"root_order={root_order} must be degree or twice the degree, {len(val)}")
    if root_order == len(val):
        raise NotImplementedError(f"root_order={root_order}=degree={len(val)} is not implemented")


_TABLE_ROOTS: Dict[Tuple[int, int, Tuple[int, ...]], Optional[int]] = {}


def _root_of_table(table, modulus, n):
    """A bit-reversed power table [psi^brv(i)] stores psi itself at index n/2: recover it and check the whole table.
    -> the root when the table IS the power table of one primitive 2n-th root (the library then serves it from the context
    every polynomial of that ring shares), None for any other table -- which the reference uses as it stands
    (ntt.py:274-290 `s = bit_rev_root_powers[m + i]`) and so do we, through a context built from the list itself."""
    key = (modulus, n, tuple(table))
    if key in _TABLE_ROOTS:
        return _TABLE_ROOTS[key]
    root = None
    if len(table) == n:
        cand = table[n // 2] % modulus
        if pow(cand, n, modulus) == modulus - 1:
            bits = n.bit_length() - 1
            acc, powers = 1, []
            for _ in range(n):
                powers.append(acc)
                acc = (acc * cand) % modulus
            if all(table[i] % modulus == powers[int(format(i, f"0{bits}b")[::-1], 2)] for i in range(n)):
                root = cand
    _TABLE_ROOTS[key] = root
    return root


def _transform_ctx(table, modulus, n, inverse):
    if len(table) < n:
        raise IndexError("list index out of range")            # what the reference's table lookup raises (ntt.py:277, :357)
    root = _root_of_table(table[:n], modulus, n)
    if root is not None:
        inv = pow(root, modulus - 2, modulus)
        return _backend.ntt_ctx(modulus, n, inv if inverse else root, root if inverse else inv)
    return _backend.table_ctx(modulus, n, table[:n], table[:n])


def cooley_tukey_ntt(val: List[int], modulus: int, root_order: int, bit_rev_root_powers: List[int]) -> List[int]:
    """In-place forward negacyclic NTT, natural order in, bit-reversed order out, centred
    outputs (ntt.py:216-291).  Mutates and returns ``val``."""
    _validate_transform_args(val, modulus, root_order, bit_rev_root_powers, "root_powers")
    n = len(val)
    ctx = _transform_ctx(bit_rev_root_powers, modulus, n, False)
    out = ctx.ntt_forward(_backend.to_i32(val, modulus))
    val[:] = out.tolist()
    return val


def gentleman_sande_intt(val: List[int], modulus: int, root_order: int,
                         bit_rev_inv_root_powers: List[int]) -> List[int]:
    """In-place inverse transform, bit-reversed in, natural out, scaled by n^-1 (ntt.py:294-377)."""
    _validate_transform_args(val, modulus, root_order, bit_rev_inv_root_powers, "inv_root_powers")
    n = len(val)
    ctx = _transform_ctx(bit_rev_inv_root_powers, modulus, n, True)
    out = ctx.ntt_inverse(_backend.to_i32(val, modulus))
    val[:] = out.tolist()
    return val


def _validate_mult_args(f, g, modulus, root, inv_root, root_order, expect_len):
    if not (isinstance(f, list) and isinstance(g, list) and isinstance(modulus, int) and isinstance(root, int)
            and isinstance(inv_root, int) and isinstance(root_order, int)):
        raise ValueError("f and g must be lists of integers; modulus, root, inv_root and root_order integers.")
    if not is_odd_prime(val=modulus):
        raise ValueError("Modulus must be an odd prime.")
    if not is_pow_two_geq_two(val=root_order):
        raise ValueError("Root order must be a power of two greater than or equal to 2.")
    if not len(f) == len(g) == expect_len:
        raise ValueError(f"f and g must both have length {expect_len}, but had len(f)={len(f)}, len(g)={len(g)}")
    if not has_primitive_root_of_unity(modulus=modulus, root_order=root_order):
        raise ValueError("Modulus does not have a primitive root of unity of order root_order.")
    if not is_primitive_root(val=root, modulus=modulus, root_order=root_order):
        raise ValueError("Input root must be a primitive root of unity.")
    if not (root * inv_root) % modulus == 1:
        raise ValueError("Input inv_root must be the inverse of the root of unity.")


def ntt_poly_mult(f: List[int], g: List[int], modulus: int, root: int, inv_root: int, root_order: int) -> List[int]:
    """Negacyclic product INTT(NTT(f) * NTT(g)) (ntt.py:380-484).  Like the reference, f and g
    are transformed and transformed back in place, so they come back as centred equivalents."""
    _validate_mult_args(f, g, modulus, root, inv_root, root_order, root_order // 2)
    n = len(f)
    ctx = _backend.ntt_ctx(modulus, n, root, inv_root)
    fi, gi = _backend.to_i32(f, modulus), _backend.to_i32(g, modulus)
    prod = ctx.poly_mul(fi, gi)                      # one launch: both transforms, product, inverse
    # INTT(NTT(x)) is the centred residue of x: the round trip the reference performs on f and g
    halfmod = modulus // 2
    f[:] = [((int(v) + halfmod) % modulus) - halfmod for v in fi]
    g[:] = [((int(v) + halfmod) % modulus) - halfmod for v in gi]
    return prod.tolist()


def ntt_poly_mult_half(f: List[int], g: List[int], modulus: int, root: int, inv_root: int,
                       root_order: int) -> List[int]:
    """The reference's even/odd variant (ntt.py:487-596) is dead code that cannot run: after its
    argument checks it unpacks three names from a two-way zip (:573-576) and raises ValueError.
    Mirrored as exactly that behaviour."""
    _validate_mult_args(f, g, modulus, root, inv_root, root_order, root_order)
    raise ValueError("not enough values to unpack (expected 3, got 2)")
