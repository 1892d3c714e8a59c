"""ctypes binding of libfusion_hip.so (C ABI declared in include/fusion_hip.h; diagnostics in include/fusion_hip_diag.h).

The shared object is built in-tree by ``__graft_entry__.build()`` into
``fusion-cryptography_amd/lib/``.  There is NO CPU fallback: if the library or a GPU is
missing every compute entry point raises :class:`FusionHipError`.
"""
import ctypes
import os
import sys
from ctypes import (POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t,
                    c_uint32, c_uint64, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.normpath(os.path.join(_HERE, "..", "lib", "libfusion_hip.so"))

FZ_OK = 0
FZ_E_BADARG = -1
FZ_E_UNSUPPORTED = -2
FZ_E_HIP = -3
FZ_E_NODEVICE = -4
FZ_E_RCCL = -5


class FusionHipError(RuntimeError):
    """Raised when libfusion_hip.so is missing, no GPU is usable, or a call fails."""

    def __init__(self, code, message):
        super().__init__(f"fusion_hip error {code}: {message}")
        self.code = code


_i32p = POINTER(c_int32)
_i64p = POINTER(c_int64)
_u32p = POINTER(c_uint32)
_ctx = c_void_p

# name -> (restype, argtypes); mirrors include/fusion_hip.h one to one
SIGNATURES = {
    "fz_version": (c_char_p, []),
    "fz_last_error": (c_char_p, []),
    "fz_device_count": (c_int, [POINTER(c_int)]),
    "fz_ctx_create": (c_int, [c_int, c_uint32, c_int, c_uint32, c_uint32, POINTER(_ctx)]),
    "fz_ctx_create_tables": (c_int, [c_int, c_uint32, c_int, _u32p, _u32p, POINTER(_ctx)]),
    "fz_ctx_destroy": (c_int, [_ctx]),
    "fz_ctx_set_stream": (c_int, [_ctx, c_void_p]),
    "fz_ctx_synchronize": (c_int, [_ctx]),
    "fz_ctx_twiddles": (c_int, [_ctx, _u32p, _u32p]),
    "fz_runtime_info": (c_int, [_ctx, POINTER(c_int), POINTER(c_int), c_char_p, c_size_t]),
    "fz_aggregate_core_ragged": (c_int, [_ctx, c_void_p, c_void_p, POINTER(c_size_t), c_size_t, c_int, c_void_p]),
    "fz_aggregate_target_partial_ragged": (c_int, [_ctx, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_size_t), c_size_t,
                                                   c_int, c_void_p, c_size_t, c_void_p, c_size_t]),
    "fz_stream_create": (c_int, [_ctx, POINTER(c_void_p)]),
    "fz_stream_create_priority": (c_int, [_ctx, c_int, POINTER(c_void_p)]),
    "fz_stream_destroy": (c_int, [_ctx, c_void_p]),
    "fz_graph_begin": (c_int, [_ctx]),
    "fz_graph_end": (c_int, [_ctx, POINTER(c_void_p)]),
    "fz_graph_launch": (c_int, [_ctx, c_void_p]),
    "fz_graph_destroy": (c_int, [c_void_p]),
    "fz_event_create": (c_int, [_ctx, POINTER(c_void_p)]),
    "fz_event_record": (c_int, [_ctx, c_void_p]),
    "fz_event_wait": (c_int, [_ctx, c_void_p]),
    "fz_event_destroy": (c_int, [c_void_p]),
    "fz_malloc": (c_int, [_ctx, c_size_t, POINTER(c_void_p)]),
    "fz_free": (c_int, [_ctx, c_void_p]),
    "fz_pool_trim": (c_int, [_ctx, c_size_t]),
    "fz_memcpy_h2d": (c_int, [_ctx, c_void_p, c_void_p, c_size_t]),
    "fz_memcpy_d2h": (c_int, [_ctx, c_void_p, c_void_p, c_size_t]),
    "fz_timer_start": (c_int, [_ctx]),
    "fz_timer_stop_ms": (c_int, [_ctx, POINTER(c_float)]),
    "fz_profile_begin": (c_int, [_ctx, c_int, c_int]),
    "fz_profile_end": (c_int, [_ctx, POINTER(ctypes.c_double), POINTER(c_int), POINTER(ctypes.c_double), POINTER(c_int)]),
    "fz_profile_end_samples": (c_int, [_ctx, POINTER(ctypes.c_double), POINTER(c_int), c_int, POINTER(c_int)]),
    "fz_ntt_forward": (c_int, [_ctx, c_void_p, c_void_p, c_size_t]),
    "fz_ntt_inverse": (c_int, [_ctx, c_void_p, c_void_p, c_size_t]),
    "fz_ntt_forward_host": (c_int, [_ctx, _i32p, c_size_t]),
    "fz_ntt_inverse_host": (c_int, [_ctx, _i32p, c_size_t]),
    "fz_pw_mul": (c_int, [_ctx, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fz_pw_add": (c_int, [_ctx, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fz_pw_sub": (c_int, [_ctx, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fz_pw_neg": (c_int, [_ctx, c_void_p, c_void_p, c_size_t]),
    "fz_pw_mulacc": (c_int, [_ctx, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fz_pw_binary_host": (c_int, [_ctx, c_int, _i32p, _i32p, _i32p, c_size_t]),
    "fz_pw_mul_bcast": (c_int, [_ctx, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fz_wide_ntt_host": (c_int, [c_int, c_uint64, c_int, POINTER(c_uint64), c_uint64, c_int, _i64p, _i64p, c_size_t]),
    "fz_wide_pw_host": (c_int, [c_int, c_uint64, c_int, _i64p, _i64p, _i64p, c_size_t]),
    "fz_wide_matvec_host": (c_int, [c_int, c_uint64, c_int, _i64p, _i64p, _i64p, c_size_t, c_int]),
    "fz_wide_norm_weight_host": (c_int, [c_int, _i64p, c_size_t, c_int, POINTER(ctypes.c_uint64), POINTER(c_int32)]),
    "fz_fill_synthetic": (c_int, [_ctx, c_void_p, c_size_t, c_uint64]),
    "fz_poly_mul": (c_int, [_ctx, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fz_poly_mul_host": (c_int, [_ctx, _i32p, _i32p, _i32p, c_size_t]),
    "fz_matvec": (c_int, [_ctx, c_void_p, c_void_p, c_void_p, c_size_t, c_int]),
    "fz_matvec_host": (c_int, [_ctx, _i32p, _i32p, _i32p, c_size_t, c_int]),
    "fz_keygen_core": (c_int, [_ctx, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int]),
    "fz_keygen_core_bcast": (c_int, [_ctx, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int]),
    "fz_sign_core": (c_int, [_ctx, c_void_p, c_void_p, c_void_p, c_size_t, c_int]),
    "fz_aggregate_core": (c_int, [_ctx, c_void_p, c_void_p, c_void_p, c_size_t, c_int]),
    "fz_aggregate_partial": (c_int, [_ctx, c_void_p, c_void_p, c_void_p, c_size_t, c_int]),
    "fz_aggregate_partial_batch": (c_int, [_ctx, c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_int]),
    "fz_aggregate_target_partial_batch": (c_int, [_ctx, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                                  c_void_p, c_size_t, c_size_t, c_size_t, c_int]),
    "fz_sign_aggregate_target_partial_batch": (c_int, [_ctx, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                       c_size_t, c_void_p, c_size_t, c_size_t, c_size_t, c_int]),
    "fz_verify_partials_batch_async": (c_int, [_ctx, c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_size_t, c_int,
                                               c_int64, c_int64, c_void_p]),
    "fz_target_partial_batch": (c_int, [_ctx, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_size_t]),
    "fz_verify_with_target_batch": (c_int, [_ctx, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int64, c_int64,
                                            POINTER(c_int)]),
    "fz_verify_with_target_batch_async": (c_int, [_ctx, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int64, c_int64,
                                                  c_void_p]),
    "fz_target_partial": (c_int, [_ctx, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fz_reduce_i64": (c_int, [_ctx, c_void_p, c_void_p, c_size_t]),
    "fz_verify_core": (c_int, [_ctx, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_size_t, c_int, c_int64, c_int64, POINTER(c_int)]),
    "fz_verify_with_target": (c_int, [_ctx, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int64,
                                      POINTER(c_int)]),
    "fz_norm_weight": (c_int, [_ctx, c_void_p, c_size_t, c_void_p, c_void_p]),
    "fz_norm_weight_host": (c_int, [_ctx, _i32p, c_size_t, _i64p, POINTER(c_int32)]),
}



class SchemeParams(ctypes.Structure):
    """fz_scheme_params of include/fusion_hip.h"""
    _fields_ = [("modulus", c_int64), ("root", c_int64), ("inv_root", c_int64),
                ("degree", c_int32), ("root_order", c_int32), ("secpar", c_int32),
                ("omega_ch", c_int32), ("omega_ag", c_int32),
                ("beta_ch", c_int64), ("beta_ag", c_int64),
                ("bytes_for_one_coef_bdd_by_beta_ch", c_int32), ("bytes_for_poly_shuffle", c_int32),
                ("sign_pre_hash_dst", ctypes.c_uint8 * 2), ("sign_hash_dst", ctypes.c_uint8 * 2),
                ("agg_xof_dst", ctypes.c_uint8 * 2)]


_u8p = POINTER(ctypes.c_uint8)
_szp = POINTER(c_size_t)
_spp = POINTER(SchemeParams)
SIGNATURES.update({
    "fz_keccak_variant": (c_char_p, []),
    "fz_sha3_256": (c_int, [c_char_p, c_size_t, _u8p]),
    "fz_shake256": (c_int, [c_char_p, c_size_t, _u8p, c_size_t]),
    "fz_format_vk": (c_int, [_spp, _i32p, _i32p, c_char_p, c_size_t, _szp]),
    "fz_decode_coefficients": (c_int, [c_char_p, c_size_t, c_int, c_int64, c_int, c_int64, c_int, _i32p]),
    "fz_hash_messages": (c_int, [_spp, c_char_p, _szp, c_size_t, _u8p]),
    "fz_challenge_coefficients": (c_int, [_spp, _i32p, _i32p, c_char_p, _szp, c_size_t, _i32p, _u8p, c_int]),
    "fz_sort_by_vk_string": (c_int, [_spp, _i32p, _i32p, c_size_t, _szp, c_int]),
    "fz_aggregation_coefficients": (c_int, [_spp, _i32p, _i32p, _u8p, _i32p, c_size_t, _i32p, c_int]),
    "fz_challenge_coefficients_dev": (c_int, [_ctx, _spp, c_void_p, _u8p, c_size_t, c_void_p]),
    "fz_challenge_hat_dev": (c_int, [_ctx, _spp, c_void_p, _u8p, c_size_t, c_void_p]),
    "fz_challenge_hat_msgs_dev": (c_int, [_ctx, _spp, c_void_p, c_char_p, _szp, c_size_t, c_void_p, _u8p]),
    "fz_sample_ntt_values": (c_int, [ctypes.c_uint64, c_int64, c_int, _i32p]),
    "fz_sample_coefficients": (c_int, [ctypes.c_uint64, c_int64, c_int, c_int64, c_int64, _i32p]),
    "fz_sample_coefficients_state": (c_int, [ctypes.c_uint64, c_int64, c_int, c_int64, c_int64, _i32p, _u32p]),
    "fz_sample_secret_polys_dev": (c_int, [_ctx, POINTER(ctypes.c_uint64), c_size_t, c_int64, c_int, c_int64, c_int64, c_void_p]),
    "fz_sample_secret_polys": (c_int, [POINTER(ctypes.c_uint64), c_size_t, c_int64, c_int, c_int64, c_int64, _i32p, c_int]),
})



class NttJob(ctypes.Structure):
    """fz_ntt_job of include/fusion_hip.h"""
    _fields_ = [("d_in", c_void_p), ("d_out", c_void_p), ("rows", c_size_t), ("inverse", c_int)]


class UniqueId(ctypes.Structure):
    """fz_unique_id (== ncclUniqueId)"""
    _fields_ = [("internal", ctypes.c_char * 128)]


SIGNATURES.update({
    "fz_ntt_multi": (c_int, [_ctx, POINTER(NttJob), c_size_t]),
    "fz_comm_unique_id": (c_int, [POINTER(UniqueId)]),
    "fz_comm_create": (c_int, [_ctx, c_int, c_int, POINTER(UniqueId), POINTER(c_void_p)]),
    "fz_comm_destroy": (c_int, [c_void_p]),
    "fz_comm_info": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int)]),
    "fz_rccl_version": (c_int, [POINTER(c_int)]),
    "fz_allreduce_i64": (c_int, [_ctx, c_void_p, c_void_p, c_size_t]),
    "fz_reduce_scatter_i64": (c_int, [_ctx, c_void_p, c_void_p, c_size_t]),
    "fz_broadcast_i32": (c_int, [_ctx, c_void_p, c_void_p, c_size_t, c_int]),
    "fz_diag_empty_launch": (c_int, [_ctx]),
    "fz_diag_copy": (c_int, [_ctx, c_void_p, c_void_p, c_size_t]),
    "fz_diag_shader_clock": (c_int, [_ctx, ctypes.c_uint, POINTER(ctypes.c_double)]),
    "fz_diag_ntt_schedule": (c_int, [_ctx, c_size_t, POINTER(c_int)]),
    "fz_diag_delay": (c_int, [_ctx, c_uint32]),
})


class QueueResult(ctypes.Structure):
    """fz_queue_result"""
    _fields_ = [("status", c_int), ("n", c_size_t), ("d_sk_hat", c_void_p), ("d_vk", c_void_p), ("d_sig", c_void_p)]


FZ_QUEUE_KEEP_SK, FZ_QUEUE_DISCARD, FZ_QUEUE_ROWS_ON_DEVICE = 1, 2, 4
SIGNATURES.update({
    "fz_queue_create": (c_int, [c_int, _spp, c_int, c_int64, c_int64, c_void_p, c_int, c_size_t, POINTER(c_void_p)]),
    "fz_queue_destroy": (c_int, [c_void_p]),
    "fz_queue_submit_keygen_sign": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_int,
                                            POINTER(ctypes.c_uint64)]),
    "fz_queue_enable_aggregate": (c_int, [c_void_p, c_int64, c_int64, c_size_t, c_int]),
    "fz_queue_submit_aggregate_verify": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, POINTER(c_int), c_int,
                                                 POINTER(ctypes.c_uint64)]),
    "fz_queue_submit_verify": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, POINTER(c_int), c_int,
                                       POINTER(ctypes.c_uint64)]),
    "fz_queue_wait": (c_int, [c_void_p, ctypes.c_uint64, POINTER(QueueResult)]),
    "fz_queue_release": (c_int, [c_void_p, ctypes.c_uint64]),
    "fz_queue_release_after": (c_int, [c_void_p, ctypes.c_uint64, _ctx]),
    "fz_queue_drain": (c_int, [c_void_p]),
    "fz_queue_stats": (c_int, [c_void_p, POINTER(ctypes.c_uint64), POINTER(ctypes.c_uint64), POINTER(ctypes.c_uint64)]),
    "fz_pinned_alloc": (c_int, [c_size_t, POINTER(c_void_p)]),
    "fz_pinned_free": (c_int, [c_void_p]),
})

# include/fusion_hip_diag.h (timers, per-dispatch profiling, probes, device-side timestamps, runtime / library reports):
# the same table serves both headers; DIAG_NAMES says which entries are diagnostics (tests/test_cabi_symbols.py)
SIGNATURES.update({
    "fz_rccl_library": (c_int, [c_char_p, c_size_t, c_char_p, c_size_t, POINTER(c_int)]),
    "fz_diag_stamps_begin": (c_int, [_ctx, c_size_t, c_size_t]),
    "fz_diag_stamps_stop": (c_int, [_ctx]),
    "fz_diag_stamps_reset": (c_int, [_ctx]),
    "fz_diag_stamps_read": (c_int, [_ctx, POINTER(c_uint64), POINTER(c_uint64), POINTER(c_uint64), POINTER(c_uint32), c_size_t,
                                    POINTER(c_size_t)]),
})
DIAG_NAMES = frozenset(n for n in SIGNATURES if n.startswith(("fz_diag_", "fz_profile_", "fz_timer_")) or
                       n in ("fz_runtime_info", "fz_keccak_variant", "fz_rccl_library"))

_lib = None


RUNTIME_CHOICE = {"runtime": "system (/opt/rocm)", "why": "default"}      # what load_library() decided, for runtime_report()


def _prefer_torch_hip_runtime():
    """A torch wheel carries its own libamdhip64 / libhsa-runtime64 (same sonames as /opt/rocm's) and maps them by path, so
    a process that loaded libfusion_hip.so FIRST (against /opt/rocm) ends up with two HIP runtimes, and the second one
    finds no GPU (`import fusion_hip; ...; import torch; torch.cuda...` -> "No HIP GPUs are available").  If torch is
    installed but not imported yet, map ITS copies first: libfusion_hip.so then binds to them by soname and a later
    `import torch` finds its runtime already alive.
    The wheel's runtime is only accepted when its HIP MAJOR version (read from torch/version.py, nothing is mapped for the
    question) equals the one the library was built with (lib/build_info.json, written by the build); on a mismatch (or
    FZ_HIP_RUNTIME=system) NOTHING of the wheel is mapped, the system runtime serves the library and torch users must import
    torch first.  fusion_hip.runtime_report() says what happened."""
    if os.environ.get("FZ_HIP_RUNTIME", "") == "system":
        RUNTIME_CHOICE.update(runtime="system (/opt/rocm)", why="FZ_HIP_RUNTIME=system")
        return
    if "torch" in sys.modules:
        RUNTIME_CHOICE.update(runtime="torch's (already imported)", why="torch was imported before fusion_hip")
        return
    try:
        import importlib.util
        import json
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        d = os.path.join(os.path.dirname(spec.origin), "lib")
        hip = os.path.join(d, "libamdhip64.so")
        hsa = os.path.join(d, "libhsa-runtime64.so")
        if not os.path.exists(hip):
            return
        want = None
        info = os.path.join(os.path.dirname(LIB_PATH), "build_info.json")
        if os.path.exists(info):
            with open(info) as fh:
                want = json.load(fh).get("hip_major")
        # the wheel's HIP version WITHOUT mapping anything of it: torch/version.py carries `hip = '7.0.51831-...'` (a dlopen
        # probe would leave the wheel's libhsa-runtime64 / libamdhip64 in the process even when they are then rejected, and
        # the system libamdhip64 would bind to the wheel's HSA by soname: a mixed stack -- ADVICE r03)
        got = None
        try:
            import re
            with open(os.path.join(os.path.dirname(spec.origin), "version.py")) as fh:
                m = re.search(r"^hip\s*(?::[^=]*)?=\s*['\"](\d+)\.", fh.read(), re.M)
            got = int(m.group(1)) if m else None
        except OSError:
            pass
        if want is None or got is None or want != got:
            why = (f"torch's HIP runtime is major version {got}, the library was built with {want}: not used" if None not in (want, got)
                   else f"cannot compare HIP versions (torch: {got}, build: {want}): torch's runtime not used")
            RUNTIME_CHOICE.update(runtime="system (/opt/rocm)", why=why + "; nothing of the wheel was mapped")
            import warnings
            warnings.warn(f"fusion_hip: {why}; keeping the system runtime -- import torch BEFORE fusion_hip in processes that use both",
                          RuntimeWarning)
            return
        if os.path.exists(hsa):
            ctypes.CDLL(hsa, mode=ctypes.RTLD_GLOBAL)
        ctypes.CDLL(hip, mode=ctypes.RTLD_GLOBAL)                   # promote: libfusion_hip.so binds to it by soname
        RUNTIME_CHOICE.update(runtime=hip, why=f"torch is installed (HIP major {got}, built with {want}): its runtime is mapped first "
                                               "so that a later `import torch` shares it")
    except Exception as e:      # noqa: BLE001 - any failure here leaves the system runtime, which is the default anyway
        RUNTIME_CHOICE.update(runtime="system (/opt/rocm)", why=f"probing torch's runtime failed: {e!r}")


def runtime_report():
    """-> dict: which HIP runtime serves libfusion_hip.so in this process and why (paths of the mapped libamdhip64 copies),
    and which librccl files are mapped (nothing is bound for the question: fusion_hip.rccl_library() says which one fz_comm_*
    uses).  More than one path in either list = two copies of that library in one process."""
    hip, rccl = [], []
    try:
        with open("/proc/self/maps") as fh:
            paths = {ln.split()[-1] for ln in fh if "/" in ln}
        hip = sorted(p for p in paths if "libamdhip64" in p)
        rccl = sorted(p for p in paths if "librccl" in p)
    except OSError:
        pass
    return dict(RUNTIME_CHOICE, mapped_libamdhip64=hip, mapped_librccl=rccl)


def load_library(path=None):
    """Load (once) and return the ctypes handle; raise FusionHipError if it is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("FUSION_HIP_LIB", LIB_PATH)
    if not os.path.exists(p):
        raise FusionHipError(FZ_E_NODEVICE,
                             f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(there is no CPU fallback)")
    _prefer_torch_hip_runtime()
    try:
        lib = ctypes.CDLL(p)
    except OSError as e:  # pragma: no cover - depends on the machine
        raise FusionHipError(FZ_E_NODEVICE, f"cannot load {p}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError here means header and library disagree
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def check(lib, rc):
    if rc != FZ_OK:
        raise FusionHipError(rc, (lib.fz_last_error() or b"").decode("utf-8", "replace"))
