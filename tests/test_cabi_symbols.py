"""The C-ABI shared object loads on a CPU-only machine and exports exactly the symbols that
include/fusion_hip.h (the surface a reference maintainer binds) and include/fusion_hip_diag.h (timers, profiling, probes,
reports: what bench.py and tools/ use) declare (no compute calls here: those need a GPU)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "fusion_hip.h")
DIAG_HEADER = os.path.join(ROOT, "include", "fusion_hip_diag.h")
GENERIC_HEADER = os.path.join(ROOT, "include", "fusion_hip_generic.h")      # the correctness path for parameters beyond the scheme's


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    import fusion_hip
    return fusion_hip.load_library()


def declared_in(path):
    return sorted(set(re.findall(r"FZ_API\s+[\w\s\*]+?\b(fz_\w+)\s*\(", open(path).read())))


def declared_symbols():
    return sorted(set(declared_in(HEADER)) | set(declared_in(DIAG_HEADER)) | set(declared_in(GENERIC_HEADER)))


def test_the_generic_path_has_its_own_header():
    """fz_wide_* do not follow fusion_hip.h's conventions (no context, host pointers, allocations per call), so
    they are not part of the surface a maintainer binds for work"""
    assert set(declared_in(GENERIC_HEADER)) == {"fz_wide_ntt_host", "fz_wide_pw_host", "fz_wide_matvec_host", "fz_wide_norm_weight_host"}
    assert not [n for n in declared_in(HEADER) if n.startswith("fz_wide_")]


def test_diagnostics_live_in_their_own_header():
    """include/fusion_hip.h is the surface INTEGRATION.md documents; timers, per-dispatch profiling, probes and
    runtime reports are declared in include/fusion_hip_diag.h only, and the ctypes table knows which is which"""
    import fusion_hip
    main, diag = set(declared_in(HEADER)), set(declared_in(DIAG_HEADER))
    assert not main & diag
    assert diag == set(fusion_hip._lib.DIAG_NAMES)
    assert not [n for n in main if n.startswith(("fz_diag_", "fz_profile_", "fz_timer_"))]
    assert "fz_runtime_info" in diag and "fz_keccak_variant" in diag and "fz_rccl_library" in diag


def test_header_declares_the_documented_surface():
    syms = declared_symbols()
    for must in ("fz_ctx_create", "fz_ntt_forward", "fz_ntt_inverse", "fz_pw_mul", "fz_pw_add", "fz_pw_mulacc",
                 "fz_matvec", "fz_keygen_core", "fz_sign_core", "fz_aggregate_core", "fz_aggregate_partial",
                 "fz_verify_core", "fz_norm_weight", "fz_reduce_i64", "fz_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol(lib):
    import fusion_hip
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in fusion_hip.h but not exported"
        assert name in fusion_hip.SIGNATURES, f"{name} has no ctypes signature in fusion_hip/_lib.py"
    assert sorted(fusion_hip.SIGNATURES) == declared_symbols()


def test_no_extra_exports_and_no_oracle_dependency():
    import fusion_hip
    out = subprocess.check_output(["nm", "-D", "--defined-only", fusion_hip.LIB_PATH], text=True)
    exported = sorted(line.split()[-1] for line in out.splitlines() if " T " in line)
    assert [s for s in exported if s.startswith("fz_")] == declared_symbols()
    assert not [s for s in exported if s.startswith("orc_")]
    needed = subprocess.check_output(["ldd", fusion_hip.LIB_PATH], text=True)
    assert "fz_oracle" not in needed


def test_fails_loudly_without_a_device(lib):
    """No CPU fallback: on a machine without a GPU context creation reports FZ_E_NODEVICE."""
    import fusion_hip
    n = ctypes.c_int(-1)
    rc = lib.fz_device_count(ctypes.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(fusion_hip.FusionHipError) as e:
        fusion_hip.Context(2147465729, 256, 3337519, pow(3337519, -1, 2147465729))
    assert e.value.code == -4
    assert lib.fz_version().startswith(b"fusion_hip")
    # argument validation happens before the device is touched
    h = ctypes.c_void_p()
    assert lib.fz_ctx_create(0, 2147465729, 100, 3337519, 1, ctypes.byref(h)) == -1      # not a power of two
    assert lib.fz_ctx_create(0, 2147465728, 256, 3337519, 1, ctypes.byref(h)) == -1      # even modulus
    assert lib.fz_ctx_create(0, 2147465729, 256, 5, 1, ctypes.byref(h)) == -1            # not a primitive root
    assert lib.fz_ctx_create(0, 2147465729, 8192, 3337519, 1, ctypes.byref(h)) == -2     # degree > 4096 (round 5; > 256 before)
    assert b"8192" in lib.fz_last_error()
    assert lib.fz_ctx_create_tables(0, 2147465729, 256, None, None, ctypes.byref(h)) == -1   # tables are required
    # the generic int64 path (fz_wide_*): arguments are checked first, and without a device the call FAILS -- it computes nothing on the host
    i64p = ctypes.POINTER(ctypes.c_int64)
    a = (ctypes.c_int64 * 4)(1, 2, 3, 4)
    out = (ctypes.c_int64 * 4)()
    p64 = lambda z: ctypes.cast(z, i64p)
    assert lib.fz_wide_pw_host(0, 2 ** 63 + 1, 1, p64(a), p64(a), p64(out), 4) == -2 and b"2^63" in lib.fz_last_error()
    assert lib.fz_wide_pw_host(0, 2 ** 40, 1, p64(a), p64(a), p64(out), 4) == -2                     # even
    assert lib.fz_wide_ntt_host(0, 97, 12, None, 0, 0, p64(a), p64(out), 1) == -1                   # not a power of two
    assert lib.fz_wide_pw_host(0, 2 ** 40 + 15, 1, p64(a), p64(a), p64(out), 4) not in (0,)         # no device: an error, not a result
    assert list(out) == [0, 0, 0, 0]
    from algebra.polynomials import PolynomialCoefficientRepresentation as PC
    with pytest.raises(fusion_hip.FusionHipError):
        PC(4294967311, 1, 1, 1, 1, [5]) + PC(4294967311, 1, 1, 1, 1, [7])
    # the batch queue owns contexts: no device, no queue (and nothing left running)
    import fusion.fusion as F
    from fusion_hip.queue import BatchQueue, PackedMessages
    with pytest.raises(fusion_hip.FusionHipError) as e:
        BatchQueue(F.fusion_setup(128, 1), workers=2)
    assert e.value.code == -4
    pm = PackedMessages(["ab", "", "cde"])
    assert pm.blob == b"abcde" and pm.off.tolist() == [0, 2, 2, 5] and pm.n == 3
    q = ctypes.c_void_p()
    assert lib.fz_queue_create(0, None, 83, 52, 256, None, 2, 1024, ctypes.byref(q)) == -1  # NULL arguments: before any device call


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under fusion-cryptography_amd/ may reference it."""
    pkg = os.path.join(ROOT, "fusion-cryptography_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(base, f)).read()
                assert "fz_oracle" not in text and "import oracle" not in text and "from oracle" not in text, f


def test_bench_touches_the_oracle_in_its_cpu_baseline_leg_only():
    """bench.py: every import of `oracle` sits inside _cpu_worker (the cpu_baseline leg); the measured path takes its
    parameters from the drop-in package and its inputs from fz_fill_synthetic"""
    import ast
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    inside = set()
    for fn in ast.walk(tree):
        if isinstance(fn, ast.FunctionDef) and fn.name == "_cpu_worker":
            inside = {id(n) for n in ast.walk(fn)}
    found = 0
    for n in ast.walk(tree):
        names = []
        if isinstance(n, ast.ImportFrom):
            names = [n.module or ""]
        elif isinstance(n, ast.Import):
            names = [a.name for a in n.names]
        if any(x == "oracle" or x.startswith("oracle.") for x in names):
            found += 1
            assert id(n) in inside, f"bench.py imports the oracle outside _cpu_worker (line {n.lineno})"
    assert found >= 1
    # the side legs (tools/bench_legs.py: sign_verify and the --full legs) never touch it at all
    legs = open(os.path.join(ROOT, "tools", "bench_legs.py")).read()
    for n in ast.walk(ast.parse(legs)):
        names = [n.module or ""] if isinstance(n, ast.ImportFrom) else [a.name for a in n.names] if isinstance(n, ast.Import) else []
        assert not any(x == "oracle" or x.startswith("oracle.") for x in names), f"tools/bench_legs.py imports the oracle (line {n.lineno})"
    assert "oracle" not in legs.replace("(tests/test_cabi_symbols.py checks bench.py AND this file)", "").replace("Nothing here touches oracle/", "")


def build_c_example(tmp_path, name="roundtrip"):
    """gcc (not hipcc), strict C99: the header is plain C and the library links without HIP on the caller's side"""
    exe = os.path.join(str(tmp_path), name)
    libdir = os.path.join(ROOT, "fusion-cryptography_amd", "lib")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", name + ".c"), "-o", exe, "-L", libdir, "-lfusion_hip",
                           "-Wl,-rpath," + libdir])
    return exe


@pytest.mark.parametrize("name", ["roundtrip", "scheme_flow", "queue_flow", "wide_roundtrip"])
def test_c_caller_compiles_and_fails_loudly_without_a_device(lib, tmp_path, name):
    exe = build_c_example(tmp_path, name)
    n = ctypes.c_int(-1)
    if lib.fz_device_count(ctypes.byref(n)) == 0 and n.value > 0:
        pytest.skip("a GPU is present (tests/test_gpu_ntt.py runs the example there)")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "no HIP device" in r.stderr
