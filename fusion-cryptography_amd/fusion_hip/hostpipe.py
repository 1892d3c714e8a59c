"""Python face of the host-side challenge pipeline in libfusion_hip.so (csrc/fz_host.cpp):
exact-format serialisation, SHA3-256 / SHAKE-256, byte decoder -- the N1 row of SURVEY.md 8f.
No GPU is needed for anything in this module."""
import ctypes
import os
from ctypes import byref, c_size_t

import numpy as np

from ._lib import SchemeParams, check, load_library

_I32P = ctypes.POINTER(ctypes.c_int32)
_U8P = ctypes.POINTER(ctypes.c_uint8)
_SZP = ctypes.POINTER(c_size_t)


def default_threads():
    return max(1, min(32, os.cpu_count() or 1))


def scheme_params(params) -> SchemeParams:
    """fusion.fusion.Params (or anything with the same attributes) -> fz_scheme_params"""
    P = SchemeParams()
    for f in ("modulus", "root", "inv_root", "degree", "root_order", "secpar", "omega_ch", "omega_ag", "beta_ch",
              "beta_ag", "bytes_for_one_coef_bdd_by_beta_ch", "bytes_for_poly_shuffle"):
        setattr(P, f, getattr(params, f))
    for f in ("sign_pre_hash_dst", "sign_hash_dst", "agg_xof_dst"):
        b = getattr(params, f)
        getattr(P, f)[0], getattr(P, f)[1] = b[0], b[1]
    return P


def _rows(a, d):
    a = np.ascontiguousarray(a, dtype=np.int32).reshape(-1, d)
    return a


def _p(a):
    return a.ctypes.data_as(_I32P)


def _pack_messages(messages):
    """-> (the messages' UTF-8 bytes back to back, offsets [N + 1] uintp).  ASCII-only str messages (byte length = character
    length) are joined and encoded in one go: a per-message encode() costs more than hashing the message does."""
    n = len(messages)
    off = np.zeros(n + 1, dtype=np.uintp)
    try:
        joined = "".join(messages)
        blob = joined.encode("utf-8")
        if len(blob) == len(joined):
            np.cumsum(np.fromiter(map(len, messages), dtype=np.uintp, count=n), out=off[1:])
            return blob, off
    except TypeError:                                   # bytes-like messages among them
        pass
    enc = [m.encode("utf-8") if isinstance(m, str) else bytes(m) for m in messages]
    np.cumsum(np.fromiter(map(len, enc), dtype=np.uintp, count=n), out=off[1:])
    return b"".join(enc), off


def keccak_variant() -> str:
    """which Keccak-f[1600] the host sponges run in this process: "scalar" | "bmi2" | "x64" | "x64v" (FZ_KECCAK forces one)"""
    return load_library().fz_keccak_variant().decode()


def sha3_256(data: bytes) -> bytes:
    lib = load_library()
    out = (ctypes.c_uint8 * 32)()
    check(lib, lib.fz_sha3_256(data, len(data), out))
    return bytes(out)


def shake256(data: bytes, n: int) -> bytes:
    lib = load_library()
    out = (ctypes.c_uint8 * max(n, 1))()
    check(lib, lib.fz_shake256(data, len(data), out, n))
    return bytes(out)[:n]


def format_vk(P: SchemeParams, left, right) -> str:
    lib = load_library()
    d = P.degree
    left, right = _rows(left, d), _rows(right, d)
    n = c_size_t()
    check(lib, lib.fz_format_vk(byref(P), _p(left), _p(right), None, 0, byref(n)))
    buf = ctypes.create_string_buffer(n.value)
    check(lib, lib.fz_format_vk(byref(P), _p(left), _p(right), buf, n.value, byref(n)))
    return buf.raw[:n.value].decode("ascii")


def decode_coefficients(b: bytes, log2_bias, modulus, degree, norm_bound, weight_bound):
    lib = load_library()
    out = np.empty(degree, dtype=np.int32)
    rc = lib.fz_decode_coefficients(b, len(b), log2_bias, modulus, degree, norm_bound, weight_bound, _p(out))
    if rc != 0:
        raise ValueError(lib.fz_last_error().decode())
    return out


def hash_messages(P: SchemeParams, messages):
    """-> [N][32] uint8: SHA3-256 digests = the pre-hashed integers, little-endian"""
    lib = load_library()
    blob, off = _pack_messages(messages)
    out = np.empty((len(messages), 32), dtype=np.uint8)
    check(lib, lib.fz_hash_messages(byref(P), blob, off.ctypes.data_as(_SZP), len(messages), out.ctypes.data_as(_U8P)))
    return out


def challenge_coefficients(P: SchemeParams, vk_left, vk_right, messages, threads=None):
    """-> (coefficient rows [N][d] int32, prehash [N][32] uint8)"""
    lib = load_library()
    d = P.degree
    L, R = _rows(vk_left, d), _rows(vk_right, d)
    N = L.shape[0]
    assert R.shape[0] == N == len(messages)
    blob, off = _pack_messages(messages)
    coefs = np.empty((N, d), dtype=np.int32)
    pre = np.empty((N, 32), dtype=np.uint8)
    rc = lib.fz_challenge_coefficients(byref(P), _p(L), _p(R), blob, off.ctypes.data_as(_SZP), N, _p(coefs),
                                       pre.ctypes.data_as(_U8P), threads or default_threads())
    if rc != 0:
        raise ValueError(lib.fz_last_error().decode())
    return coefs, pre


def sort_by_vk_string(P: SchemeParams, vk_left, vk_right, threads=None):
    lib = load_library()
    d = P.degree
    L, R = _rows(vk_left, d), _rows(vk_right, d)
    order = np.empty(L.shape[0], dtype=np.uintp)
    check(lib, lib.fz_sort_by_vk_string(byref(P), _p(L), _p(R), L.shape[0], order.ctypes.data_as(_SZP),
                                        threads or default_threads()))
    return order.astype(np.int64)


def aggregation_coefficients(P: SchemeParams, vk_left, vk_right, prehash, c_hat, threads=None):
    """inputs in sorted key order -> alpha coefficient rows [N][d]"""
    lib = load_library()
    d = P.degree
    L, R, C = _rows(vk_left, d), _rows(vk_right, d), _rows(c_hat, d)
    pre = np.ascontiguousarray(prehash, dtype=np.uint8).reshape(-1, 32)
    N = L.shape[0]
    out = np.empty((N, d), dtype=np.int32)
    rc = lib.fz_aggregation_coefficients(byref(P), _p(L), _p(R), pre.ctypes.data_as(_U8P), _p(C), N, _p(out),
                                         threads or default_threads())
    if rc != 0:
        raise ValueError(lib.fz_last_error().decode())
    return out


def sample_ntt_values(seed, modulus, degree):
    """sample_polynomial_ntt_representation(...).values for a non-negative int seed"""
    lib = load_library()
    out = np.empty(degree, dtype=np.int32)
    check(lib, lib.fz_sample_ntt_values(seed, modulus, degree, _p(out)))
    return out


def sample_coefficients(seed, modulus, degree, norm_bound, weight_bound):
    """sample_polynomial_coefficient_representation(...).coefficients for a non-negative int seed"""
    lib = load_library()
    out = np.empty(degree, dtype=np.int32)
    check(lib, lib.fz_sample_coefficients(seed, modulus, degree, norm_bound, weight_bound, _p(out)))
    return out


def sample_coefficients_with_state(seed, modulus, degree, norm_bound, weight_bound):
    """-> (coefficients, state): the same polynomial and the generator afterwards as a tuple of 625 ints, the middle element
    of what random.getstate() returns after sample_polynomial_coefficient_representation(..., seed=seed)"""
    lib = load_library()
    out = np.empty(degree, dtype=np.int32)
    st = np.empty(625, dtype=np.uint32)
    check(lib, lib.fz_sample_coefficients_state(seed, modulus, degree, norm_bound, weight_bound, _p(out),
                                                st.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))))
    return out, tuple(st.tolist())


def sample_secret_polys(seeds, modulus, degree, norm_bound, weight_bound, threads=None):
    """[N][2][degree]: the left (seed) and right (seed + 1) secret polynomial of each key"""
    lib = load_library()
    sd = np.ascontiguousarray(seeds, dtype=np.uint64)
    out = np.empty((sd.size, 2, degree), dtype=np.int32)
    check(lib, lib.fz_sample_secret_polys(sd.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), sd.size, modulus, degree,
                                          norm_bound, weight_bound, _p(out), threads or default_threads()))
    return out
