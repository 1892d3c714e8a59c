"""Drop-in for the reference's ``algebra.polynomials``: the two ring-element value classes,
``transform`` and the seeded samplers, with every vector operation executed on the GPU.

Reference surface mirrored (file:line in the reference checkout):
  PolynomialRepresentation :16 · PolynomialCoefficientRepresentation :65 ·
  PolynomialNTTRepresentation :230 · transform :391 ·
  sample_polynomial_coefficient_representation :436 · sample_polynomial_ntt_representation :470
The module path and class names matter: ``str(GeneralMatrix)`` embeds
``<class 'algebra.polynomials.PolynomialNTTRepresentation'>`` and that text is hashed by the
scheme (fusion/fusion.py:416-418, :586-590).

Differences in mechanism (never in results):
  * the primitive-root check of the constructor (511 modular powers per object in the
    reference, polynomials.py:40) is done once per parameter tuple and memoised;
  * coefficient-domain ``*`` (schoolbook O(d^2) in the reference, :196-208) runs as
    NTT -> pointwise -> INTT on the device, which yields the same centred residues.
"""
from random import randrange, seed as random_seed
from typing import Dict, List, Optional, Tuple, Union

import numpy as np

from . import _backend

cached_halfmods: Dict[int, int] = {}
cached_logmods: Dict[int, int] = {}
_VALIDATED: Dict[Tuple[int, int, int, int], Optional[Exception]] = {}


def _check_ring_parameters(modulus, root, inv_root, root_order):
    """ValueError checks of polynomials.py:36-45, evaluated once per tuple."""
    key = (modulus, root, inv_root, root_order)
    if key not in _VALIDATED:
        err = None
        if (modulus - 1) % root_order != 0:
            err = ValueError("root_order must be a divisor of modulus - 1")
        elif pow(root, root_order, modulus) != 1:
            err = ValueError("root must be a root of unity of order root_order")
        else:
            acc = root % modulus
            for _ in range(1, root_order):
                if acc == 1:
                    err = ValueError("root must be a primitive root of unity of order root_order")
                    break
                acc = (acc * root) % modulus
            if err is None and (root * inv_root) % modulus != 1:
                err = ValueError("root and inv_root must be inverses of each other")
        _VALIDATED[key] = err
    if _VALIDATED[key] is not None:
        raise _VALIDATED[key]


class PolynomialRepresentation(object):
    modulus: int
    degree: int
    root: int
    inv_root: int
    root_order: int

    def __init__(self, modulus: int, degree: int, root: int, inv_root: int, root_order: int):
        for name, v in (("modulus", modulus), ("degree", degree), ("root", root), ("inv_root", inv_root),
                        ("root_order", root_order)):
            if not isinstance(v, int):
                raise TypeError(f"{name} must be an int")
        _check_ring_parameters(modulus, root, inv_root, root_order)
        self.modulus = modulus
        self.degree = degree
        self.root = root
        self.inv_root = inv_root
        self.root_order = root_order

    @property
    def halfmod(self) -> int:
        if self.modulus not in cached_halfmods:
            cached_halfmods[self.modulus] = self.modulus // 2
        return cached_halfmods[self.modulus]

    @property
    def logmod(self) -> int:
        if self.modulus not in cached_logmods:
            cached_logmods[self.modulus] = self.modulus.bit_length() - 1
        return cached_logmods[self.modulus]

    # -- storage ----------------------------------------------------------------------------
    # The data attribute (`coefficients` / `values`) is a list of Python ints as far as any caller can tell (the reference's
    # tests mutate its elements in place: tests/test_polynomials.py:280-283, tests/test_fusion.py:864-867).  An object the
    # library produced itself starts as the int32 row the device returned (`_arr`: never written to afterwards, possibly a
    # view shared with sibling objects) and builds the list on first access; from then on the list is the only copy.  A
    # key that goes keygen -> sign, or a signature that goes sign -> aggregate -> verify, never becomes 21 248 Python ints.
    def _get_data(self):
        lst = self._list
        if lst is None:
            lst = self._list = self._arr.tolist()
            self._arr = None
        return lst

    def _set_data(self, data):
        self._list = data
        self._arr = None

    def _data(self):
        return self._get_data()

    def _len(self):
        return len(self._list) if self._arr is None else int(self._arr.shape[0])

    def _like(self, data):
        """same ring, new data the library produced itself (a fresh list of Python ints): the per-element isinstance scan
        of __init__ -- 70 % of the object API's time -- has nothing to find"""
        z = object.__new__(type(self))
        z.modulus, z.degree, z.root, z.inv_root, z.root_order = self.modulus, self.degree, self.root, self.inv_root, self.root_order
        z._list, z._arr = data, None
        return z

    def _like_arr(self, arr):
        """same ring, data = a one-dimensional int32 array the library produced (not copied, never written to)"""
        z = object.__new__(type(self))
        z.modulus, z.degree, z.root, z.inv_root, z.root_order = self.modulus, self.degree, self.root, self.inv_root, self.root_order
        z._list, z._arr = None, arr
        return z

    def _like_list(self, values):
        """same ring, data = a list of Python ints (values the device's int32 cannot hold: -(x mod q) for q >= 2^31)"""
        z = self._like_arr(None)
        z._list = values
        return z

    # -- helpers shared by both representations ---------------------------------------------
    def _ring(self):
        return _backend.ring_ctx(self.modulus, max(1, self._len()))

    def _ntt(self):
        return _backend.ntt_ctx(self.modulus, self.degree, self.root, self.inv_root)

    def _i32(self):
        return self._arr if self._arr is not None else _backend.to_i32(self._list, self.modulus)

    def _same_ring(self, other, verb):
        """NotImplementedError on any mismatch, in the reference's order (:121-132, :289-302)."""
        if self.modulus != other.modulus:
            raise NotImplementedError(f"Cannot {verb} polynomials with different moduli")
        if self.degree != other.degree:
            raise NotImplementedError(f"Cannot {verb} polynomials with different degrees")
        if self.root != other.root:
            raise NotImplementedError(f"Cannot {verb} polynomials with different roots of unity")
        if self.root_order != other.root_order:
            raise NotImplementedError(f"Cannot {verb} polynomials with different root orders")


class PolynomialCoefficientRepresentation(PolynomialRepresentation):
    coefficients = property(PolynomialRepresentation._get_data, PolynomialRepresentation._set_data,
                            doc="List[int]; see the storage note in PolynomialRepresentation")

    def __init__(self, modulus: int, degree: int, root: int, inv_root: int, root_order: int,
                 coefficients: List[int]):
        super().__init__(modulus=modulus, degree=degree, root=root, inv_root=inv_root, root_order=root_order)
        if not isinstance(coefficients, list):
            raise TypeError("coefficients must be a list")
        if not all(isinstance(x, int) for x in coefficients):
            raise TypeError("coefficients must be a list of ints")
        if len(coefficients) != degree:
            raise ValueError("coefficients must be of length degree")
        self.coefficients = coefficients

    def __str__(self):
        return (f"PolynomialCoefficientRepresentation(modulus={self.modulus}, degree={self.degree}, "
                f"root={self.root}, inv_root={self.inv_root}, root_order={self.root_order}, "
                f"coefficients={self.coefficients})")

    def __repr__(self):
        return self.__str__()

    def __eq__(self, other):
        # equality of residues, inv_root not compared (:98-112)
        if not isinstance(other, PolynomialCoefficientRepresentation):
            return False
        if (self.modulus, self.degree, self.root, self.root_order) != \
                (other.modulus, other.degree, other.root, other.root_order):
            return False
        q = self.modulus
        if self._arr is not None and other._arr is not None:
            return bool(np.all((self._arr.astype(np.int64) - other._arr) % q == 0)) if self._arr.shape == other._arr.shape else \
                all((x - y) % q == 0 for x, y in zip(self.coefficients, other.coefficients))
        return all((x - y) % q == 0 for x, y in zip(self.coefficients, other.coefficients))

    def __add__(self, other):
        if other == 0:
            return self
        if not isinstance(other, PolynomialCoefficientRepresentation):
            raise NotImplementedError(f"Addition for {type(self)} and {type(other)} not implemented")
        self._same_ring(other, "add")
        return self._like_arr(self._ring().pw_add(self._i32(), other._i32()))

    def __radd__(self, other):
        if other == 0:
            return self
        return self + other

    def __neg__(self):
        # -(x mod q), in [-(q-1), 0]: deliberately NOT centred, as in the reference (:155-163)
        arr, lst = _backend.neg_values(self._ring(), self._i32(), self.modulus)
        return self._like_arr(arr) if lst is None else self._like_list(lst)

    def __sub__(self, other):
        return self + (-other)

    def __rsub__(self, other):
        return other + (-self)

    def __mul__(self, other):
        if other == 0:
            return 0
        if other == 1:
            return self
        if not isinstance(other, PolynomialCoefficientRepresentation):
            raise NotImplementedError(f"Multiplication for {type(self)} and {type(other)} not implemented")
        if self.modulus != other.modulus:
            raise NotImplementedError(f"Multiplication for {type(self)} with different moduli not implemented")
        if self.degree != other.degree:
            raise NotImplementedError(f"Multiplication for {type(self)} with different degrees not implemented")
        if self.root != other.root:
            raise NotImplementedError(f"Multiplication for {type(self)} with different roots of unity not implemented")
        if self.root_order != other.root_order:
            raise NotImplementedError(f"Multiplication for {type(self)} with different root orders not implemented")
        # negacyclic product mod (X^d + 1, q): NTT -> pointwise -> INTT on the device
        return self._like_arr(self._ntt().poly_mul(self._i32(), other._i32()))

    def __rmul__(self, other):
        return self.__mul__(other=other)

    def norm(self, p: Union[int, str]) -> int:
        if p != "infty":
            raise NotImplementedError(f"norm for p={p} not implemented")
        # max |x| over the STORED values (:221-224), not over their residues.  The device type is int32 (int64 on the generic
        # path, q >= 2^32): a stored value outside it (a caller's own list) is a Python int the library never holds -- the
        # reference's answer for it is one exact max() over the list on the host (marshalling, not algebra: no reduction, no
        # arithmetic mod q)
        wide = _backend.is_wide(self.modulus)
        lo, hi = (-(2 ** 63) + 1, 2 ** 63 - 1) if wide else (_backend.INT32_MIN, _backend.INT32_MAX)
        if self._arr is None and any(x < lo or x > hi for x in self.coefficients):
            return max(abs(x) for x in self.coefficients)
        mx, _ = self._ring().norm_weight(self._arr if self._arr is not None
                                         else np.array(self.coefficients, dtype=np.int64 if wide else np.int32))
        return int(mx[0])

    def weight(self) -> int:
        _, wt = self._ring().norm_weight(self._i32())
        return int(wt[0])


class PolynomialNTTRepresentation(PolynomialRepresentation):
    values = property(PolynomialRepresentation._get_data, PolynomialRepresentation._set_data,
                      doc="List[int]; see the storage note in PolynomialRepresentation")

    def __init__(self, modulus: int, degree: int, root: int, inv_root: int, root_order: int, values: List[int]):
        super().__init__(modulus=modulus, degree=degree, root=root, inv_root=inv_root, root_order=root_order)
        if not isinstance(values, list):
            raise TypeError("values must be a list")
        if not all(isinstance(x, int) for x in values):
            raise TypeError("values must be a list of ints")
        if len(values) != degree:
            raise ValueError("values must have length degree")
        self.values = values

    def __str__(self):
        return (f"PolynomialNTTRepresentation(modulus={self.modulus}, degree={self.degree}, root={self.root}, "
                f"inv_root={self.inv_root}, root_order={self.root_order}, values={self.values})")

    def __repr__(self):
        return self.__str__()

    def __eq__(self, other):
        q = self.modulus
        if other == 0:     # (for a polynomial `other` this asks whether IT is zero, as the reference does)
            if self._arr is not None:          # no reason to build 256 Python ints to learn that a row is not zero
                return not bool(np.any(self._arr.astype(np.int64) % q))
            return all(x % q == 0 for x in self.values)
        if not isinstance(other, PolynomialNTTRepresentation):
            return False
        if (self.modulus, self.degree, self.root_order, self.root, self.inv_root) != \
                (other.modulus, other.degree, other.root_order, other.root, other.inv_root):
            return False
        if self._len() != other._len():
            return False
        if self._arr is not None and other._arr is not None:
            return bool(np.all((self._arr.astype(np.int64) - other._arr) % q == 0))
        return all((x - y) % q == 0 for x, y in zip(self.values, other.values))

    def __add__(self, other):
        if other == 0:
            return self
        if not isinstance(other, PolynomialNTTRepresentation):
            raise NotImplementedError(f"Addition for {type(self)} and {type(other)} not implemented")
        self._same_ring(other, "add")
        if self._len() != other._len():
            raise NotImplementedError("Cannot add polynomials with different lengths")
        return self._like_arr(self._ring().pw_add(self._i32(), other._i32()))

    def __radd__(self, other):
        if other == 0:
            return self
        return self + other

    def __neg__(self):
        arr, lst = _backend.neg_values(self._ring(), self._i32(), self.modulus)
        return self._like_arr(arr) if lst is None else self._like_list(lst)

    def __sub__(self, other):
        return self + (-other)

    def __rsub__(self, other):
        return other + (-self)

    def __mul__(self, other):
        if other == 0:
            return 0
        if other == 1:
            return self
        if not isinstance(other, PolynomialNTTRepresentation):
            raise NotImplementedError(f"Multiplication for {type(self)} and {type(other)} not implemented")
        if self.modulus != other.modulus:
            raise NotImplementedError(f"Multiplication for {type(self)} with different moduli not implemented")
        if self.degree != other.degree:
            raise NotImplementedError(f"Multiplication for {type(self)} with different degrees not implemented")
        if self.root != other.root:
            raise NotImplementedError(f"Multiplication for {type(self)} with different roots of unity not implemented")
        if self.root_order != other.root_order:
            raise NotImplementedError(f"Multiplication for {type(self)} with different root orders not implemented")
        if self._len() != other._len():
            raise NotImplementedError(f"Multiplication for {type(self)} with different lengths not implemented")
        return self._like_arr(self._ring().pw_mul(self._i32(), other._i32()))

    def __rmul__(self, other):
        return self.__mul__(other=other)


def _check_transformable(x, n):
    """The argument errors cooley_tukey_ntt / gentleman_sande_intt would raise (ntt.py:255-270)."""
    from .ntt import has_primitive_root_of_unity, is_odd_prime, is_pow_two_geq_two
    if not is_odd_prime(val=x.modulus):
        raise ValueError(f"modulus={x.modulus} must be an odd prime")
    if not has_primitive_root_of_unity(modulus=x.modulus, root_order=x.root_order):
        raise ValueError(f"modulus={x.modulus} does not have a primitive root of order root_order={x.root_order}")
    if not is_pow_two_geq_two(val=n):
        raise ValueError(f"len(val)={n} must be a power of 2 greater than 1")
    if x.root_order != 2 * n and x.root_order != n:
        raise ValueError(f"root_order={x.root_order} must be degree or twice the degree, {n}")
    if x.root_order == n:
        raise NotImplementedError(f"root_order={x.root_order}=degree={n} is not implemented")


def _as(cls, x):
    """an empty object of class `cls` in x's ring (a template for _like / _like_arr)"""
    z = object.__new__(cls)
    z.modulus, z.degree, z.root, z.inv_root, z.root_order = x.modulus, x.degree, x.root, x.inv_root, x.root_order
    z._list, z._arr = [], None
    return z


def transform(x: Union[PolynomialCoefficientRepresentation, PolynomialNTTRepresentation]
              ) -> Union[PolynomialNTTRepresentation, PolynomialCoefficientRepresentation]:
    """Coefficient <-> NTT domain on a copy of the data (polynomials.py:391-433)."""
    if isinstance(x, (PolynomialCoefficientRepresentation, PolynomialNTTRepresentation)):
        _check_transformable(x, x._len())
    # the ring parameters were validated when x was built; the result shares them (the reference re-validates: same outcome)
    if isinstance(x, PolynomialCoefficientRepresentation):
        out = x._ntt().ntt_forward(x._i32())
        return PolynomialRepresentation._like_arr(_as(PolynomialNTTRepresentation, x), out.reshape(-1))
    if isinstance(x, PolynomialNTTRepresentation):
        out = x._ntt().ntt_inverse(x._i32())
        return PolynomialRepresentation._like_arr(_as(PolynomialCoefficientRepresentation, x), out.reshape(-1))
    raise NotImplementedError(f"Transform for {type(x)} not implemented")


def sample_polynomial_coefficient_representation(modulus: int, degree: int, root: int, inv_root: int,
                                                 root_order: int, norm_bound: int, weight_bound: int,
                                                 seed: Optional[int]) -> PolynomialCoefficientRepresentation:
    """Seeded sampler on the process-global ``random`` (polynomials.py:436-467): the same draws in
    the same order, so the same seed gives the same polynomial as the reference."""
    if seed is not None:
        random_seed(seed)
    count = max(0, min(degree, weight_bound))
    bound = max(0, min(modulus // 2, norm_bound))
    coefficients: List[int] = []
    for _ in range(count):
        magnitude = 1 + randrange(bound)
        sign = 1 - 2 * randrange(2)
        coefficients.append(magnitude * sign)
    coefficients.extend([0] * (degree - count))
    if count < degree:   # Fisher-Yates from the top index down
        for i in range(degree - 1, 0, -1):
            j = randrange(i + 1)
            coefficients[i], coefficients[j] = coefficients[j], coefficients[i]
    return PolynomialCoefficientRepresentation(modulus=modulus, degree=degree, root=root, inv_root=inv_root,
                                               root_order=root_order, coefficients=coefficients)


def sample_polynomial_ntt_representation(modulus: int, degree: int, root: int, inv_root: int, root_order: int,
                                         seed: Optional[int]) -> PolynomialNTTRepresentation:
    """Uniform values randrange(q) - q//2 (polynomials.py:470-488)."""
    if seed is not None:
        random_seed(seed)
    shift = modulus // 2
    values = [randrange(modulus) - shift for _ in range(degree)]
    return PolynomialNTTRepresentation(modulus=modulus, degree=degree, root=root, inv_root=inv_root,
                                       root_order=root_order, values=values)
