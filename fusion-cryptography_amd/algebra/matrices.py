"""Drop-in for the reference's ``algebra.matrices.GeneralMatrix`` (matrices.py:10-153).

The container semantics (validation, aliasing of the list it is given, ``str`` format, duck
typing over any class with ``__eq__ __add__ __neg__ __sub__ __mul__``) are the reference's.
When the entries are polynomial objects of ``algebra.polynomials`` the arithmetic is batched:
one device call per matrix operation instead of one Python-level operation per entry --
``A * B`` for the (1 x l).(l x 1) shape the scheme uses is a single fused multiply-accumulate
kernel (fz_matvec), ``M * poly`` and ``M + N`` one pointwise launch over all rows.
Entries of other algebraic classes (e.g. plain ints) take the generic element-operator path,
like the reference.
"""
from typing import List, Union

import numpy as np

from . import _backend
from .polynomials import PolynomialCoefficientRepresentation, PolynomialNTTRepresentation


def is_algebraic_class(cls):
    return all(hasattr(cls, name) for name in ("__eq__", "__add__", "__neg__", "__sub__", "__mul__"))


def _is_poly_class(cls):
    return cls is PolynomialNTTRepresentation or cls is PolynomialCoefficientRepresentation


class GeneralMatrix:
    elem_class: type
    matrix: List[List[object]]

    def __init__(self, matrix: List[list]):
        if not isinstance(matrix, list):
            raise ValueError("Matrix must be a list")
        if not matrix:
            raise ValueError("Matrix must not be empty.")
        if any(not isinstance(row, list) for row in matrix):
            raise ValueError("Matrix must be a list of lists")
        if any(not row for row in matrix):
            raise ValueError("Matrix must not contain empty lists")
        width = len(matrix[0])
        if any(len(row) != width for row in matrix):
            raise ValueError("All rows must have the same length")
        cls = matrix[0][0].__class__
        if not is_algebraic_class(cls=cls) or not all(isinstance(item, cls) for row in matrix for item in row):
            raise ValueError("Matrix must contain only instances of the same algebraic class")
        self.elem_class = cls
        self.matrix = matrix

    def __str__(self):
        return f"GeneralMatrix(elem_class={self.elem_class}, matrix={self.matrix})"

    def __repr__(self):
        return self.__str__()

    def __len__(self):
        return len(self.matrix)

    def __iter__(self):
        return iter(self.matrix)

    def __getitem__(self, item):
        return self.matrix[item]

    def __setitem__(self, key, value):
        self.matrix[key] = value

    def __delitem__(self, key):
        self.matrix[key] = 0

    # ---- batching helpers (polynomial entries only) -------------------------------------------
    def _shape(self):
        return len(self.matrix), len(self.matrix[0])

    def _uniform_ring(self, *others):
        """(q, degree, template) when every polynomial involved shares one parameter tuple and row
        length, else None (then the generic per-element path produces the reference's errors)."""
        if not _is_poly_class(self.elem_class):
            return None
        first = self.matrix[0][0]
        key = (first.modulus, first.degree, first.root, first.inv_root, first.root_order, first._len())
        for m in (self,) + others:
            rows = m.matrix if isinstance(m, GeneralMatrix) else [[m]]
            for row in rows:
                for z in row:
                    if not isinstance(z, self.elem_class) or \
                            (z.modulus, z.degree, z.root, z.inv_root, z.root_order, z._len()) != key:
                        return None
        return first

    def _stack(self):
        return _backend.stack_polys([z for row in self.matrix for z in row], self.matrix[0][0].modulus)

    def _rebuild(self, template, arr, rows, cols):
        # entries are rows of `arr` (views, never written to); their lists are built only if somebody asks for them
        return GeneralMatrix(matrix=[[template._like_arr(arr[i * cols + j]) for j in range(cols)] for i in range(rows)])

    # ---- algebra -------------------------------------------------------------------------------
    def __eq__(self, other):
        if other == 0:
            return all(all(item == 0 for item in row) for row in self.matrix)
        if not isinstance(other, GeneralMatrix) or self.elem_class != other.elem_class:
            return False
        if self._shape() != other._shape():
            return False
        return self.matrix == other.matrix

    def __add__(self, other):
        if other == 0:
            return self
        if not isinstance(other, GeneralMatrix) or self.elem_class != other.elem_class:
            raise NotImplementedError("Can only add GeneralMatrix objects of the same algebraic class")
        if self._shape() != other._shape():
            raise ValueError("Matrix dimensions must match")
        rows, cols = self._shape()
        t = self._uniform_ring(other)
        # an all-zero right entry makes the reference return the LEFT object itself (uncentred);
        # keep that exact behaviour by using the element operators whenever it could matter
        if t is not None and not any(z == 0 for row in other.matrix for z in row):
            ctx = _backend.ring_ctx(t.modulus, t._len())
            return self._rebuild(t, ctx.pw_add(self._stack(), other._stack()), rows, cols)
        return GeneralMatrix(matrix=[[self.matrix[i][j] + other.matrix[i][j] for j in range(cols)]
                                     for i in range(rows)])

    def __radd__(self, other):
        if other == 0:
            return self
        return self + other

    def __neg__(self):
        rows, cols = self._shape()
        t = self._uniform_ring()
        if t is not None:
            ctx = _backend.ring_ctx(t.modulus, t._len())
            st = self._stack()
            if t.modulus < 2 ** 31 or _backend.is_wide(t.modulus):
                return self._rebuild(t, ctx.pw_neg(st), rows, cols)
            # 2^31 <= q < 2^32: -(x mod q) does not fit the int32 rows (see _backend.neg_values): the device's centred negation,
            # the representative shifted on the way back, lists of Python ints
            c = ctx.pw_sub(np.zeros_like(st), st)
            return GeneralMatrix(matrix=[[t._like_list([int(v) if v <= 0 else int(v) - t.modulus for v in c[i * cols + j].tolist()])
                                          for j in range(cols)] for i in range(rows)])
        return GeneralMatrix(matrix=[[-self.matrix[i][j] for j in range(cols)] for i in range(rows)])

    def __sub__(self, other):
        return self + (-other)

    def __mul__(self, other):
        rows, cols = self._shape()
        if isinstance(other, self.elem_class):
            # every entry times one element (matrices.py:109-114)
            t = self._uniform_ring(other)
            if (t is not None and self.elem_class is PolynomialNTTRepresentation and not other == 0
                    and not other == 1):
                ctx = _backend.ring_ctx(t.modulus, t._len())
                a = self._stack()
                b = np.broadcast_to(other._i32(), a.shape)
                return self._rebuild(t, ctx.pw_mul(a, b), rows, cols)
            return GeneralMatrix(matrix=[[self.matrix[i][j] * other for j in range(cols)] for i in range(rows)])
        if not isinstance(other, GeneralMatrix) or self.elem_class != other.elem_class:
            raise TypeError("Can only multiply matrices of the same algebraic class")
        if cols != len(other.matrix):
            raise ValueError("Matrix dimension mismatch")
        ocols = len(other.matrix[0])
        t = self._uniform_ring(other)
        if (t is not None and self.elem_class is PolynomialNTTRepresentation
                and not any(z == 0 for m in (self, other) for row in m.matrix for z in row)):
            # result[i][j] = sum_k self[i][k] * other[k][j]: one fused multiply-accumulate launch per
            # result row (for the scheme's (1 x l).(l x 1) shape: exactly one launch)
            ctx = _backend.ring_ctx(t.modulus, t._len())
            d = t._len()
            A = self._stack().reshape(rows, cols, d)
            B = other._stack().reshape(cols, ocols, d).transpose(1, 0, 2)      # [ocols][cols][d]
            out = [ctx.matvec(A[i], np.ascontiguousarray(B)) for i in range(rows)]   # each [ocols][d]
            return GeneralMatrix(matrix=[[t._like_arr(out[i][j]) for j in range(ocols)] for i in range(rows)])
        result = []
        for i in range(rows):
            out_row = []
            for j in range(ocols):
                acc = self.matrix[i][0] * other.matrix[0][j]
                for k in range(1, cols):
                    acc += self.matrix[i][k] * other.matrix[k][j]
                out_row.append(acc)
            result.append(out_row)
        res = GeneralMatrix.__new__(GeneralMatrix)
        res.elem_class = self.elem_class
        res.matrix = result
        return res

    def __mod__(self, other):
        if not isinstance(other, int):
            raise TypeError("Can only take the remainder of a matrix with an integer")
        if other <= 1:
            raise ValueError("Modulus must be greater than 1")
        rows, cols = self._shape()
        return GeneralMatrix(matrix=[[self.matrix[i][j] % other for j in range(cols)] for i in range(rows)])

    def norm(self, p: Union[int, str]):
        if not all(hasattr(z, "norm") for y in self.matrix for z in y):
            raise NotImplementedError("Matrix elements must have a norm method")
        if p == "infty":
            return max(max(z.norm(p=p) for z in y) for y in self.matrix)

    def weight(self):
        if not all(hasattr(z, "weight") for y in self.matrix for z in y):
            raise NotImplementedError("Matrix elements must have a weight method")
        return max(max(z.weight() for z in y) for y in self.matrix)
