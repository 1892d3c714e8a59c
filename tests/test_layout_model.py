"""The index model of the two-pass kernels (tools/ntt_layout_model.py: strided pass with uniform
twiddles, transpose, contiguous pass with per-lane twiddles, n^-1 folded into the last inverse
stage) agrees with the oracle's Longa-Naehrig loops for every supported degree."""
import os
import sys

import numpy as np
import pytest

from oracle import oracle as O

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import ntt_layout_model as M  # noqa: E402


def root_for(q, d):
    for g in range(2, 2000):
        r = pow(g, (q - 1) // (2 * d), q)
        if pow(r, d, q) == q - 1:
            return r
    raise AssertionError


@pytest.mark.parametrize("q,d", [(O.PRIME, 256), (O.PRIME, 64), (O.PRIME, 128), (O.PRIME, 32), (12289, 256), (257, 32)])
def test_two_pass_model(q, d, coracle):
    root = {(O.PRIME, 256): 3337519, (O.PRIME, 64): 23584283}.get((q, d)) or root_for(q, d)
    inv = pow(root, q - 2, q)
    x = O.splitmix_centered(d, 3 * d, q).reshape(3, d)
    f = coracle.ntt_forward(x, q, root)
    g = coracle.ntt_inverse(x, q, inv)
    for i in range(3):
        row = [int(v) for v in x[i]]
        assert M.fwd_two_pass(row, q, root) == f[i].tolist()
        assert M.inv_two_pass(row, q, inv) == g[i].tolist()


@pytest.mark.parametrize("d", [2, 4, 8, 16])
def test_small_model(d, coracle):
    q = O.PRIME
    root = root_for(q, d)
    x = O.splitmix_centered(d, d, q)
    assert M.fwd_small([int(v) for v in x], q, root) == coracle.ntt_forward(x, q, root).tolist()


@pytest.mark.parametrize("secpar", [128, 256])
def test_radix4_model(secpar, coracle):
    P = O.PARAMS[secpar]
    q, d = P["q"], P["d"]
    x = O.splitmix_centered(secpar, 2 * d, q).reshape(2, d)
    f, g = coracle.ntt_forward(x, q, P["root"]), coracle.ntt_inverse(x, q, P["inv_root"])
    for i in range(2):
        row = [int(v) for v in x[i]]
        assert M.fwd_radix4(row, q, P["root"]) == f[i].tolist()
        assert M.inv_radix4(row, q, P["inv_root"]) == g[i].tolist()
