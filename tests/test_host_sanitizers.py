"""The host half of the challenge pipeline (csrc/fz_host.cpp: serialiser, Keccak, decoder, MT19937 clone) built with
AddressSanitizer + UndefinedBehaviorSanitizer and driven through its C ABI on edge-shaped inputs (SURVEY.md section 5:
sanitizers run on the CPU build only).  Results are compared with CPython's hashlib inside the driver."""
import hashlib
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = r'''
#include "fusion_hip.h"
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
int fz_set_error(int code, const char *fmt, ...) { (void)fmt; return code; }
static fz_scheme_params params(int secpar) {
    fz_scheme_params P;
    memset(&P, 0, sizeof(P));
    P.modulus = 2147465729ll; P.secpar = secpar;
    if (secpar == 256) { P.degree = 256; P.root = 3337519; P.inv_root = 1978410468; P.root_order = 512; P.omega_ch = 60; P.omega_ag = 60; }
    else { P.degree = 64; P.root = 23584283; P.inv_root = 540632852; P.root_order = 128; P.omega_ch = 31; P.omega_ag = 31; }
    P.beta_ch = 1; P.beta_ag = 1;
    P.bytes_for_one_coef_bdd_by_beta_ch = secpar == 256 ? 33 : 17;
    P.bytes_for_poly_shuffle = secpar == 256 ? 8448 : 1088;
    P.sign_pre_hash_dst[0] = 0; P.sign_pre_hash_dst[1] = (uint8_t)(secpar == 256 ? 1 : 0);
    P.sign_hash_dst[0] = 1; P.sign_hash_dst[1] = 1; P.agg_xof_dst[0] = 2; P.agg_xof_dst[1] = 1;
    return P;
}
int main() {
    // SHAKE / SHA3 over every length around the rate boundaries; digests printed for the Python side
    std::vector<uint8_t> data(700);
    for (size_t i = 0; i < data.size(); ++i) data[i] = (uint8_t)(i * 131 + 7);
    for (size_t n : {0u, 1u, 135u, 136u, 137u, 271u, 272u, 273u, 700u}) {
        uint8_t d32[32];
        std::vector<uint8_t> x(301);
        if (fz_sha3_256(data.data(), n, d32) || fz_shake256(data.data(), n, x.data(), x.size())) return 2;
        printf("sha3 %zu ", n);
        for (int i = 0; i < 32; ++i) printf("%02x", d32[i]);
        printf("\nshake %zu ", n);
        for (size_t i = 0; i < x.size(); ++i) printf("%02x", x[i]);
        printf("\n");
    }
    for (int secpar : {128, 256}) {
        fz_scheme_params P = params(secpar);
        const int d = P.degree, N = 37;
        std::vector<int32_t> L((size_t)N * d), R((size_t)N * d), coefs((size_t)N * d), c_hat((size_t)N * d), alpha((size_t)N * d);
        unsigned s = 12345u + (unsigned)secpar;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (int32_t)(s >> 1) - (1 << 30); };
        for (auto &v : L) v = rnd();
        for (auto &v : R) v = rnd();
        L[0] = 0; L[1] = -1; L[2] = 1073732864; R[d - 1] = -1073732864;
        std::string msgs;
        std::vector<size_t> off(N + 1, 0);
        for (int i = 0; i < N; ++i) { msgs += std::string((size_t)(i * 7 % 300), (char)('a' + i % 26)); off[i + 1] = msgs.size(); }   // incl. empty messages
        std::vector<uint8_t> pre((size_t)N * 32);
        if (fz_challenge_coefficients(&P, L.data(), R.data(), msgs.data(), off.data(), N, coefs.data(), pre.data(), 3)) return 3;
        std::vector<size_t> order(N);
        if (fz_sort_by_vk_string(&P, L.data(), R.data(), N, order.data(), 2)) return 4;
        for (auto &v : c_hat) v = rnd();
        if (fz_aggregation_coefficients(&P, L.data(), R.data(), pre.data(), c_hat.data(), N, alpha.data(), 3)) return 5;
        size_t len = 0;
        if (fz_format_vk(&P, L.data(), R.data(), nullptr, 0, &len)) return 6;
        std::vector<char> text(len);
        if (fz_format_vk(&P, L.data(), R.data(), text.data(), len, &len)) return 7;
        if (fz_format_vk(&P, L.data(), R.data(), text.data(), len - 1, &len) == 0) return 8;     // too small: must refuse
        long w = 0;
        for (int i = 0; i < N * d; ++i) w += coefs[i] != 0;
        printf("secpar %d weight %ld\n", secpar, w);
        // decoder on a buffer that is exactly long enough and one byte short
        std::vector<uint8_t> buf(20000, 0x5a);
        std::vector<int32_t> out(d);
        size_t need = (size_t)((P.omega_ch + 7) / 8) + (size_t)(P.bytes_for_one_coef_bdd_by_beta_ch * 2) * P.omega_ch;
        if (fz_decode_coefficients(buf.data(), need, secpar, P.modulus, d, 1, P.omega_ch, out.data())) return 9;
        if (fz_decode_coefficients(buf.data(), need - 1, secpar, P.modulus, d, 1, P.omega_ch, out.data()) == 0) return 10;
        std::vector<uint64_t> seeds = {0, 1, 0xffffffffull, 0x100000000ull, 0x123456789abull};
        std::vector<int32_t> polys(seeds.size() * 2 * d);
        if (fz_sample_secret_polys(seeds.data(), seeds.size(), P.modulus, d, 52, d, polys.data(), 2)) return 11;
        if (fz_sample_ntt_values(77, P.modulus, d, out.data())) return 12;
    }
    printf("done\n");
    return 0;
}
'''


def test_host_pipeline_under_asan_and_ubsan(tmp_path):
    src = tmp_path / "driver.cpp"
    src.write_text(DRIVER)
    exe = tmp_path / "driver"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-pthread",
           "-I", os.path.join(ROOT, "include"), str(src), os.path.join(ROOT, "fusion-cryptography_amd", "csrc", "fz_host.cpp"),
           "-o", str(exe)]
    try:
        subprocess.check_call(cmd)
    except subprocess.CalledProcessError:
        pytest.skip("g++ with sanitizer runtimes not available")
    data = bytes((i * 131 + 7) & 0xff for i in range(700))
    for variant in ("scalar", "bmi2", "x64", "x64v"):          # every Keccak-f[1600] implementation (a CPU without one falls back)
        r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, FZ_KECCAK=variant, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0",
                                    UBSAN_OPTIONS="print_stacktrace=1"))
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
        assert "done" in r.stdout and "ERROR" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
        lines = dict(((ln.split()[0], int(ln.split()[1])), ln.split()[2]) for ln in r.stdout.splitlines() if ln.startswith(("sha3", "shake")))
        for n in (0, 1, 135, 136, 137, 271, 272, 273, 700):
            assert lines[("sha3", n)] == hashlib.sha3_256(data[:n]).hexdigest()
            assert lines[("shake", n)] == hashlib.shake_256(data[:n]).hexdigest(301)
        assert "secpar 128 weight %d" % (37 * 31) in r.stdout and "secpar 256 weight %d" % (37 * 60) in r.stdout


def test_host_pipeline_threads_under_tsan(tmp_path):
    """the same driver under ThreadSanitizer: the host pipeline spreads signers over std::threads (parallel_for) and shares
    read-only inputs, per-signer outputs and one `bad` flag -- no data race may be reported (race
    detection)"""
    src = tmp_path / "driver.cpp"
    src.write_text(DRIVER)
    exe = tmp_path / "driver_tsan"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread",
           "-I", os.path.join(ROOT, "include"), str(src), os.path.join(ROOT, "fusion-cryptography_amd", "csrc", "fz_host.cpp"),
           "-o", str(exe)]
    try:
        subprocess.check_call(cmd)
    except subprocess.CalledProcessError:
        pytest.skip("g++ with the thread sanitizer runtime not available")
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0:report_signal_unsafe=0"))
    if "FATAL: ThreadSanitizer" in r.stderr and "unexpected memory mapping" in r.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow memory in this container")
    assert "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0 and "done" in r.stdout, r.stdout[-1000:] + r.stderr[-2000:]
