// Host Keccak-f[1600] forms on ONE core (what bounds hash_ag: fusion/fusion.py:632-652 is one serial sponge), and the
// instruction latencies that decide between them.  Build:  g++|clang++ -O3 -std=c++17 keccak_host.cpp -o keccak_host
// Every form is checked against the plain C form before it is timed.
//   scalar      25 lanes in general registers (16 of them: ~45 spills + reloads per round), andn / rorx with BMI
//   plane512    a plane per zmm register (csrc/fz_host.cpp's first AVX-512 form): chi register-wise after an in-register pi,
//               a 5 x 5 transpose back to planes every round
//   lane128     ONE lane per xmm register, AVX-512VL: 32 registers, three-input XOR and chi in one vpternlogq, vprolq for
//               rho: 90 instructions per round, no transposes, (almost) no spills
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <immintrin.h>

static const uint64_t RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL, 0x0000000080000001ULL,
    0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL,
    0x000000000000800aULL, 0x800000008000000aULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};

// ---- the round, written once over (T, X3, CHI, ROL, XRC): theta with three-input XORs, rho + pi, chi, iota --------------
#define KROUND(A, E, r) { \
    const T c0 = X3(X3(A##0, A##5, A##10), A##15, A##20), c1 = X3(X3(A##1, A##6, A##11), A##16, A##21), c2 = X3(X3(A##2, A##7, A##12), A##17, A##22), \
            c3 = X3(X3(A##3, A##8, A##13), A##18, A##23), c4 = X3(X3(A##4, A##9, A##14), A##19, A##24); \
    const T r0 = ROL(c0, 1), r1 = ROL(c1, 1), r2 = ROL(c2, 1), r3 = ROL(c3, 1), r4 = ROL(c4, 1); \
    T b0, b1, b2, b3, b4; \
    b0 = X3(A##0, c4, r1); b1 = ROL(X3(A##6, c0, r2), 44); b2 = ROL(X3(A##12, c1, r3), 43); b3 = ROL(X3(A##18, c2, r4), 21); b4 = ROL(X3(A##24, c3, r0), 14); \
    E##0 = XRC(CHI(b0, b1, b2), r); E##1 = CHI(b1, b2, b3); E##2 = CHI(b2, b3, b4); E##3 = CHI(b3, b4, b0); E##4 = CHI(b4, b0, b1); \
    b0 = ROL(X3(A##3, c2, r4), 28); b1 = ROL(X3(A##9, c3, r0), 20); b2 = ROL(X3(A##10, c4, r1), 3); b3 = ROL(X3(A##16, c0, r2), 45); b4 = ROL(X3(A##22, c1, r3), 61); \
    E##5 = CHI(b0, b1, b2); E##6 = CHI(b1, b2, b3); E##7 = CHI(b2, b3, b4); E##8 = CHI(b3, b4, b0); E##9 = CHI(b4, b0, b1); \
    b0 = ROL(X3(A##1, c0, r2), 1); b1 = ROL(X3(A##7, c1, r3), 6); b2 = ROL(X3(A##13, c2, r4), 25); b3 = ROL(X3(A##19, c3, r0), 8); b4 = ROL(X3(A##20, c4, r1), 18); \
    E##10 = CHI(b0, b1, b2); E##11 = CHI(b1, b2, b3); E##12 = CHI(b2, b3, b4); E##13 = CHI(b3, b4, b0); E##14 = CHI(b4, b0, b1); \
    b0 = ROL(X3(A##4, c3, r0), 27); b1 = ROL(X3(A##5, c4, r1), 36); b2 = ROL(X3(A##11, c0, r2), 10); b3 = ROL(X3(A##17, c1, r3), 15); b4 = ROL(X3(A##23, c2, r4), 56); \
    E##15 = CHI(b0, b1, b2); E##16 = CHI(b1, b2, b3); E##17 = CHI(b2, b3, b4); E##18 = CHI(b3, b4, b0); E##19 = CHI(b4, b0, b1); \
    b0 = ROL(X3(A##2, c1, r3), 62); b1 = ROL(X3(A##8, c2, r4), 55); b2 = ROL(X3(A##14, c3, r0), 39); b3 = ROL(X3(A##15, c4, r1), 41); b4 = ROL(X3(A##21, c0, r2), 2); \
    E##20 = CHI(b0, b1, b2); E##21 = CHI(b1, b2, b3); E##22 = CHI(b2, b3, b4); E##23 = CHI(b3, b4, b0); E##24 = CHI(b4, b0, b1); }
#define KBODY(LOAD, STORE) \
    T a0 = LOAD(0), a1 = LOAD(1), a2 = LOAD(2), a3 = LOAD(3), a4 = LOAD(4), a5 = LOAD(5), a6 = LOAD(6), a7 = LOAD(7), a8 = LOAD(8), a9 = LOAD(9), \
      a10 = LOAD(10), a11 = LOAD(11), a12 = LOAD(12), a13 = LOAD(13), a14 = LOAD(14), a15 = LOAD(15), a16 = LOAD(16), a17 = LOAD(17), a18 = LOAD(18), \
      a19 = LOAD(19), a20 = LOAD(20), a21 = LOAD(21), a22 = LOAD(22), a23 = LOAD(23), a24 = LOAD(24); \
    T e0, e1, e2, e3, e4, e5, e6, e7, e8, e9, e10, e11, e12, e13, e14, e15, e16, e17, e18, e19, e20, e21, e22, e23, e24; \
    for (int r = 0; r < 24; r += 2) { KROUND(a, e, r) KROUND(e, a, r + 1) } \
    STORE(0, a0); STORE(1, a1); STORE(2, a2); STORE(3, a3); STORE(4, a4); STORE(5, a5); STORE(6, a6); STORE(7, a7); STORE(8, a8); STORE(9, a9); \
    STORE(10, a10); STORE(11, a11); STORE(12, a12); STORE(13, a13); STORE(14, a14); STORE(15, a15); STORE(16, a16); STORE(17, a17); STORE(18, a18); \
    STORE(19, a19); STORE(20, a20); STORE(21, a21); STORE(22, a22); STORE(23, a23); STORE(24, a24);

// plain C
#define T uint64_t
#define X3(a, b, c) ((a) ^ (b) ^ (c))
#define CHI(a, b, c) ((a) ^ (~(b) & (c)))
#define ROL(x, n) (((x) << (n)) | ((x) >> (64 - (n))))
#define XRC(x, r) ((x) ^ RC[r])
#define LD(i) s[i]
#define ST(i, v) s[i] = v
void keccak_scalar(uint64_t *s) { KBODY(LD, ST) }
__attribute__((target("bmi,bmi2"))) void keccak_bmi2(uint64_t *s) { KBODY(LD, ST) }
#undef T
#undef X3
#undef CHI
#undef ROL
#undef XRC
#undef LD
#undef ST

// one lane per xmm register
#define T __m128i
#define X3(a, b, c) _mm_ternarylogic_epi64((a), (b), (c), 0x96)
#define CHI(a, b, c) _mm_ternarylogic_epi64((a), (b), (c), 0xD2)
#define ROL(x, n) _mm_rol_epi64((x), (n))
#define XRC(x, r) _mm_xor_si128((x), _mm_loadl_epi64(reinterpret_cast<const __m128i *>(RC + (r))))
#define LD(i) _mm_loadl_epi64(reinterpret_cast<const __m128i *>(s + (i)))
#define ST(i, v) _mm_storel_epi64(reinterpret_cast<__m128i *>(s + (i)), (v))
__attribute__((target("avx512f,avx512vl"))) void keccak_lane128(uint64_t *s) { KBODY(LD, ST) }
#undef T
#undef X3
#undef CHI
#undef ROL
#undef XRC
#undef LD
#undef ST

// a plane per zmm register (the form csrc/fz_host.cpp had in rounds 3-5)
__attribute__((target("avx512f,avx512vl,avx512dq,avx512bw"))) void keccak_plane512(uint64_t *s) {
    const __mmask8 k5 = 0x1f;
    __m512i r0 = _mm512_maskz_loadu_epi64(k5, s), r1 = _mm512_maskz_loadu_epi64(k5, s + 5), r2 = _mm512_maskz_loadu_epi64(k5, s + 10),
            r3 = _mm512_maskz_loadu_epi64(k5, s + 15), r4 = _mm512_maskz_loadu_epi64(k5, s + 20);
    const __m512i idx_m = _mm512_setr_epi64(4, 0, 1, 2, 3, 5, 6, 7), idx_p = _mm512_setr_epi64(1, 2, 3, 4, 0, 5, 6, 7);
    const __m512i rho0 = _mm512_setr_epi64(0, 1, 62, 28, 27, 0, 0, 0), rho1 = _mm512_setr_epi64(36, 44, 6, 55, 20, 0, 0, 0),
                  rho2 = _mm512_setr_epi64(3, 10, 43, 25, 39, 0, 0, 0), rho3 = _mm512_setr_epi64(41, 45, 15, 21, 8, 0, 0, 0),
                  rho4 = _mm512_setr_epi64(18, 2, 61, 56, 14, 0, 0, 0);
    const __m512i pi0 = _mm512_setr_epi64(0, 3, 1, 4, 2, 5, 6, 7), pi1 = _mm512_setr_epi64(1, 4, 2, 0, 3, 5, 6, 7), pi2 = _mm512_setr_epi64(2, 0, 3, 1, 4, 5, 6, 7),
                  pi3 = _mm512_setr_epi64(3, 1, 4, 2, 0, 5, 6, 7), pi4 = _mm512_setr_epi64(4, 2, 0, 3, 1, 5, 6, 7);
    const __m512i ra1 = idx_m, ra2 = _mm512_setr_epi64(3, 4, 0, 1, 2, 5, 6, 7), ra3 = _mm512_setr_epi64(2, 3, 4, 0, 1, 5, 6, 7), ra4 = idx_p;
    const __m512i rc1 = idx_p, rc2 = ra3, rc3 = ra2, rc4 = idx_m;
    for (int round = 0; round < 24; ++round) {
        __m512i c = _mm512_ternarylogic_epi64(r0, r1, r2, 0x96);
        c = _mm512_ternarylogic_epi64(c, r3, r4, 0x96);
        const __m512i cm = _mm512_permutexvar_epi64(idx_m, c), cp = _mm512_rol_epi64(_mm512_permutexvar_epi64(idx_p, c), 1);
        r0 = _mm512_ternarylogic_epi64(r0, cm, cp, 0x96); r1 = _mm512_ternarylogic_epi64(r1, cm, cp, 0x96); r2 = _mm512_ternarylogic_epi64(r2, cm, cp, 0x96);
        r3 = _mm512_ternarylogic_epi64(r3, cm, cp, 0x96); r4 = _mm512_ternarylogic_epi64(r4, cm, cp, 0x96);
        const __m512i q0 = _mm512_permutexvar_epi64(pi0, _mm512_rolv_epi64(r0, rho0)), q1 = _mm512_permutexvar_epi64(pi1, _mm512_rolv_epi64(r1, rho1)),
                      q2 = _mm512_permutexvar_epi64(pi2, _mm512_rolv_epi64(r2, rho2)), q3 = _mm512_permutexvar_epi64(pi3, _mm512_rolv_epi64(r3, rho3)),
                      q4 = _mm512_permutexvar_epi64(pi4, _mm512_rolv_epi64(r4, rho4));
        __m512i e0 = _mm512_ternarylogic_epi64(q0, q1, q2, 0xD2);
        const __m512i e1 = _mm512_ternarylogic_epi64(q1, q2, q3, 0xD2), e2 = _mm512_ternarylogic_epi64(q2, q3, q4, 0xD2),
                      e3 = _mm512_ternarylogic_epi64(q3, q4, q0, 0xD2), e4 = _mm512_ternarylogic_epi64(q4, q0, q1, 0xD2);
        e0 = _mm512_xor_si512(e0, _mm512_maskz_set1_epi64(0x01, (long long)RC[round]));
        const __m512i f0 = e0, f1 = _mm512_permutexvar_epi64(ra1, e1), f2 = _mm512_permutexvar_epi64(ra2, e2), f3 = _mm512_permutexvar_epi64(ra3, e3),
                      f4 = _mm512_permutexvar_epi64(ra4, e4);
#define TB(A0, A1, A2, A3, A4) _mm512_mask_blend_epi64(0x10, _mm512_mask_blend_epi64(0x0c, _mm512_mask_blend_epi64(0x02, A0, A1), _mm512_mask_blend_epi64(0x08, A2, A3)), A4)
        const __m512i t0 = TB(f0, f1, f2, f3, f4), t1 = TB(f4, f0, f1, f2, f3), t2 = TB(f3, f4, f0, f1, f2), t3 = TB(f2, f3, f4, f0, f1), t4 = TB(f1, f2, f3, f4, f0);
#undef TB
        r0 = t0; r1 = _mm512_permutexvar_epi64(rc1, t1); r2 = _mm512_permutexvar_epi64(rc2, t2); r3 = _mm512_permutexvar_epi64(rc3, t3); r4 = _mm512_permutexvar_epi64(rc4, t4);
    }
    _mm512_mask_storeu_epi64(s, k5, r0); _mm512_mask_storeu_epi64(s + 5, k5, r1); _mm512_mask_storeu_epi64(s + 10, k5, r2);
    _mm512_mask_storeu_epi64(s + 15, k5, r3); _mm512_mask_storeu_epi64(s + 20, k5, r4);
}


// ---- state in MEMORY, one round per call, one output row at a time -------------------------------------------------------
// The register-resident forms above leave the allocation of 50 + 15 values to 15 general registers to the compiler: ~90 of
// the ~225 instructions it emits per round are spill traffic, and the core retires them at its dispatch width.  Here a round
// reads A[] and writes E[] (two 200-byte arrays in L1: four loads + two stores per cycle on Zen 5), keeps only D[5] and one
// row of B in registers, and never spills.
#define RL(x, n) ((n) == 0 ? (x) : (((x) << ((n) & 63)) | ((x) >> ((64 - (n)) & 63))))
#define MROW(o, i0, x0, n0, i1, x1, n1, i2, x2, n2, i3, x3, n3, i4, x4, n4, rc) { \
    const uint64_t b0 = RL(A[i0] ^ d##x0, n0), b1 = RL(A[i1] ^ d##x1, n1), b2 = RL(A[i2] ^ d##x2, n2), b3 = RL(A[i3] ^ d##x3, n3), b4 = RL(A[i4] ^ d##x4, n4); \
    E[o] = b0 ^ (~b1 & b2) ^ (rc); E[o + 1] = b1 ^ (~b2 & b3); E[o + 2] = b2 ^ (~b3 & b4); E[o + 3] = b3 ^ (~b4 & b0); E[o + 4] = b4 ^ (~b0 & b1); }
#define MROUND(rc) \
    const uint64_t c0 = A[0] ^ A[5] ^ A[10] ^ A[15] ^ A[20], c1 = A[1] ^ A[6] ^ A[11] ^ A[16] ^ A[21], c2 = A[2] ^ A[7] ^ A[12] ^ A[17] ^ A[22], \
                   c3 = A[3] ^ A[8] ^ A[13] ^ A[18] ^ A[23], c4 = A[4] ^ A[9] ^ A[14] ^ A[19] ^ A[24]; \
    const uint64_t d0 = c4 ^ RL(c1, 1), d1 = c0 ^ RL(c2, 1), d2 = c1 ^ RL(c3, 1), d3 = c2 ^ RL(c4, 1), d4 = c3 ^ RL(c0, 1); \
    MROW(0, 0, 0, 0, 6, 1, 44, 12, 2, 43, 18, 3, 21, 24, 4, 14, rc) \
    MROW(5, 3, 3, 28, 9, 4, 20, 10, 0, 3, 16, 1, 45, 22, 2, 61, 0) \
    MROW(10, 1, 1, 1, 7, 2, 6, 13, 3, 25, 19, 4, 8, 20, 0, 18, 0) \
    MROW(15, 4, 4, 27, 5, 0, 36, 11, 1, 10, 17, 2, 15, 23, 3, 56, 0) \
    MROW(20, 2, 2, 62, 8, 3, 55, 14, 4, 39, 15, 0, 41, 21, 1, 2, 0)
__attribute__((noinline)) static void mround_base(const uint64_t *__restrict A, uint64_t *__restrict E, uint64_t rc) { MROUND(rc) }
__attribute__((noinline, target("bmi,bmi2"))) static void mround_bmi2(const uint64_t *__restrict A, uint64_t *__restrict E, uint64_t rc) { MROUND(rc) }
void keccak_mem(uint64_t *s) {
    alignas(64) uint64_t t[25];
    for (int r = 0; r < 24; r += 2) { mround_base(s, t, RC[r]); mround_base(t, s, RC[r + 1]); }
}
void keccak_mem_bmi2(uint64_t *s) {
    alignas(64) uint64_t t[25];
    for (int r = 0; r < 24; r += 2) { mround_bmi2(s, t, RC[r]); mround_bmi2(t, s, RC[r + 1]); }
}
// the same with all 24 rounds in one function (no calls), the arrays made opaque to the optimiser between rounds so that it
// does not promote them to registers again
__attribute__((target("bmi,bmi2"))) void keccak_mem24_bmi2(uint64_t *s) {
    alignas(64) uint64_t t[25];
    for (int r = 0; r < 24; r += 2) {
        { const uint64_t *A = s; uint64_t *E = t; asm volatile("" : : "r"(A), "r"(E) : "memory"); MROUND(RC[r]) }
        { const uint64_t *A = t; uint64_t *E = s; asm volatile("" : : "r"(A), "r"(E) : "memory"); MROUND(RC[r + 1]) }
    }
}

// ---- one assembly routine with a fixed register plan, early parity and the block loop inside (tools/gen_keccak_x64.py)
extern "C" {
extern const uint64_t fz_keccak_rc_x64[24] __attribute__((visibility("hidden")));
const uint64_t fz_keccak_rc_x64[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL, 0x0000000080000001ULL,
    0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL,
    0x000000000000800aULL, 0x800000008000000aULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
void fz_keccak_blocks_x64(uint64_t *s, const uint8_t *in, uint8_t *out, size_t nblocks);
void fz_keccak_blocks_x64v(uint64_t *s, const uint8_t *in, uint8_t *out, size_t nblocks);
}
static void keccak_asmv(uint64_t *s) { fz_keccak_blocks_x64v(s, nullptr, nullptr, 1); }
static void keccak_asm(uint64_t *s) { fz_keccak_blocks_x64(s, nullptr, nullptr, 1); }
#ifndef FZ_INC
#define FZ_INC "../../fusion-cryptography_amd/csrc/fz_keccak_x64.inc"
#endif
#include FZ_INC

// ---- dependent-chain latency of the instructions the vector forms are made of -------------------------------------------
#define CHAIN(NAME, DECL, STEP, SINK) \
    __attribute__((target("avx512f,avx512vl,avx512dq,avx512bw"), noinline)) double NAME(int iters) { \
        DECL; \
        const auto t0 = std::chrono::steady_clock::now(); \
        for (int i = 0; i < iters; ++i) { STEP STEP STEP STEP STEP STEP STEP STEP } \
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); \
        SINK; \
        return dt / iters / 8 * 1e9; }
static volatile long long g_sink;
CHAIN(lat_gpr_xor, uint64_t v = (uint64_t)iters; uint64_t w = 0x9e3779b97f4a7c15ull, asm volatile("xor %1, %0" : "+r"(v) : "r"(w));, g_sink = (long long)v)
CHAIN(lat_gpr_rorx, uint64_t v = (uint64_t)iters, asm volatile("rorx $7, %0, %0" : "+r"(v));, g_sink = (long long)v)
CHAIN(lat_xmm_ternlog, __m128i v = _mm_set1_epi64x(iters); __m128i w = _mm_set1_epi64x(77), asm volatile("vpternlogq $0x96, %1, %1, %0" : "+v"(v) : "v"(w));, g_sink = _mm_cvtsi128_si64(v))
CHAIN(lat_xmm_rol, __m128i v = _mm_set1_epi64x(iters), asm volatile("vprolq $7, %0, %0" : "+v"(v));, g_sink = _mm_cvtsi128_si64(v))
CHAIN(lat_zmm_ternlog, __m512i v = _mm512_set1_epi64(iters); __m512i w = _mm512_set1_epi64(77), asm volatile("vpternlogq $0x96, %1, %1, %0" : "+v"(v) : "v"(w));, g_sink = _mm_cvtsi128_si64(_mm512_castsi512_si128(v)))
CHAIN(lat_zmm_rolv, __m512i v = _mm512_set1_epi64(iters); __m512i w = _mm512_set1_epi64(7), asm volatile("vprolvq %1, %0, %0" : "+v"(v) : "v"(w));, g_sink = _mm_cvtsi128_si64(_mm512_castsi512_si128(v)))
CHAIN(lat_zmm_permq, __m512i v = _mm512_set1_epi64(iters & 7); __m512i w = _mm512_setr_epi64(1, 2, 3, 4, 0, 5, 6, 7), asm volatile("vpermq %0, %1, %0" : "+v"(v) : "v"(w));, g_sink = _mm_cvtsi128_si64(_mm512_castsi512_si128(v)))
CHAIN(lat_zmm_permt2q, __m512i v = _mm512_set1_epi64(iters & 7); __m512i w = _mm512_setr_epi64(1, 2, 3, 4, 0, 5, 6, 7), asm volatile("vpermt2q %1, %1, %0" : "+v"(v) : "v"(w));, g_sink = _mm_cvtsi128_si64(_mm512_castsi512_si128(v)))
CHAIN(lat_zmm_unpck, __m512i v = _mm512_set1_epi64(iters); __m512i w = _mm512_set1_epi64(77), asm volatile("vpunpcklqdq %1, %0, %0" : "+v"(v) : "v"(w));, g_sink = _mm_cvtsi128_si64(_mm512_castsi512_si128(v)))
CHAIN(lat_zmm_blend, __m512i v = _mm512_set1_epi64(iters); __m512i w = _mm512_set1_epi64(77); __mmask8 k = 0x55, asm volatile("vpblendmq %1, %0, %0 %{%2%}" : "+v"(v) : "v"(w), "Yk"(k));, g_sink = _mm_cvtsi128_si64(_mm512_castsi512_si128(v)))

// throughput of independent xmm ternlogs / rotates (how many vector pipes take them)
__attribute__((target("avx512f,avx512vl"), noinline)) double thr_xmm(int iters, int what) {
    __m128i v0 = _mm_set1_epi64x(1), v1 = v0, v2 = v0, v3 = v0, v4 = v0, v5 = v0, v6 = v0, v7 = v0, w = _mm_set1_epi64x(77);
    const auto t0 = std::chrono::steady_clock::now();
    if (what == 0)
        for (int i = 0; i < iters; ++i)
            asm volatile("vpternlogq $0x96, %8, %8, %0\n vpternlogq $0x96, %8, %8, %1\n vpternlogq $0x96, %8, %8, %2\n vpternlogq $0x96, %8, %8, %3\n"
                         "vpternlogq $0x96, %8, %8, %4\n vpternlogq $0x96, %8, %8, %5\n vpternlogq $0x96, %8, %8, %6\n vpternlogq $0x96, %8, %8, %7"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(w));
    else
        for (int i = 0; i < iters; ++i)
            asm volatile("vprolq $7, %0, %0\n vprolq $7, %1, %1\n vprolq $7, %2, %2\n vprolq $7, %3, %3\n vprolq $7, %4, %4\n vprolq $7, %5, %5\n vprolq $7, %6, %6\n vprolq $7, %7, %7"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    g_sink = _mm_cvtsi128_si64(_mm_xor_si128(_mm_xor_si128(_mm_xor_si128(v0, v1), _mm_xor_si128(v2, v3)), _mm_xor_si128(_mm_xor_si128(v4, v5), _mm_xor_si128(v6, v7))));
    return dt / iters / 8 * 1e9;
}

int main() {
    __builtin_cpu_init();
    const bool bmi = __builtin_cpu_supports("bmi2"), avx512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl") &&
                                                               __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512bw");
    struct V { const char *name; void (*fn)(uint64_t *); bool ok; } vs[] = {
        {"scalar", keccak_scalar, true}, {"scalar + bmi2", keccak_bmi2, bmi}, {"plane512", keccak_plane512, avx512}, {"lane128", keccak_lane128, avx512},
        {"memory", keccak_mem, true}, {"memory + bmi2", keccak_mem_bmi2, bmi}, {"memory24 + bmi2", keccak_mem24_bmi2, bmi}, {"assembly (bmi2)", keccak_asm, bmi}, {"assembly hybrid", keccak_asmv, bmi && avx512}};
    uint64_t ref[25];
    for (int i = 0; i < 25; ++i) ref[i] = 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1);
    keccak_scalar(ref);
    keccak_scalar(ref);
    for (const V &v : vs) {
        if (!v.ok) { printf("%-16s not supported by this CPU\n", v.name); continue; }
        uint64_t st[25];
        for (int i = 0; i < 25; ++i) st[i] = 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1);
        v.fn(st);
        v.fn(st);
        if (memcmp(st, ref, sizeof(st)) != 0) { printf("%-16s WRONG RESULT\n", v.name); continue; }
        double best = 1e30;
        for (int pass = 0; pass < 7; ++pass) {
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < 20000; ++i) v.fn(st);
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (dt < best) best = dt;
        }
        printf("%-16s %7.1f ns per permutation (dependent chain of 20000, state through memory)\n", v.name, best / 20000 * 1e9);
    }
    if (bmi) {       // the block loop: 4096 blocks absorbed per call (the state stays in the routine's frame), checked against the C form
        static uint8_t data[4096 * 136];
        for (size_t i = 0; i < sizeof(data); ++i) data[i] = (uint8_t)(i * 131 + (i >> 7));
        uint64_t a[25] = {0}, b[25] = {0};
        for (int blk = 0; blk < 4096; ++blk) {
            for (int i = 0; i < 17; ++i) { uint64_t w; memcpy(&w, data + 136 * blk + 8 * i, 8); a[i] ^= w; }
            keccak_bmi2(a);
        }
        fz_keccak_blocks_x64(b, data, nullptr, 4096);
        if (memcmp(a, b, sizeof(a)) != 0) printf("assembly block loop WRONG RESULT\n");
        else {
            if (avx512) {
                uint64_t c[25] = {0};
                fz_keccak_blocks_x64v(c, data, nullptr, 4096);
                if (memcmp(a, c, sizeof(a)) != 0) printf("hybrid block loop WRONG RESULT\n");
                double bestv = 1e30;
                for (int pass = 0; pass < 7; ++pass) {
                    auto t0 = std::chrono::steady_clock::now();
                    fz_keccak_blocks_x64v(c, data, nullptr, 4096);
                    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    if (dt < bestv) bestv = dt;
                }
                printf("absorbing 4096 blocks of 136 bytes: hybrid (general + xmm registers) block loop %.1f ns per block (%.3f GB/s)\n", bestv / 4096 * 1e9, 4096 * 136 / bestv / 1e9);
            }
            double best = 1e30, best_c = 1e30;
            for (int pass = 0; pass < 7; ++pass) {
                auto t0 = std::chrono::steady_clock::now();
                fz_keccak_blocks_x64(b, data, nullptr, 4096);
                double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (dt < best) best = dt;
                t0 = std::chrono::steady_clock::now();
                for (int blk = 0; blk < 4096; ++blk) {
                    for (int i = 0; i < 17; ++i) { uint64_t w; memcpy(&w, data + 136 * blk + 8 * i, 8); a[i] ^= w; }
                    keccak_bmi2(a);
                }
                dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (dt < best_c) best_c = dt;
            }
            printf("absorbing 4096 blocks of 136 bytes: assembly block loop %.1f ns per block (%.3f GB/s), C + bmi2 %.1f ns per block (%.3f GB/s)\n",
                   best / 4096 * 1e9, 4096 * 136 / best / 1e9, best_c / 4096 * 1e9, 4096 * 136 / best_c / 1e9);
        }
    }
    if (avx512) {
        const int n = 2000000;
        printf("dependent-chain latency, ns per instruction (x core GHz = cycles):\n");
        printf("  xor r64 %.3f  rorx %.3f | xmm vpternlogq %.3f  xmm vprolq %.3f | zmm vpternlogq %.3f  vprolvq %.3f  vpermq %.3f  vpermt2q %.3f  vpunpcklqdq %.3f  vpblendmq %.3f\n",
               lat_gpr_xor(n), lat_gpr_rorx(n), lat_xmm_ternlog(n), lat_xmm_rol(n), lat_zmm_ternlog(n), lat_zmm_rolv(n), lat_zmm_permq(n), lat_zmm_permt2q(n), lat_zmm_unpck(n),
               lat_zmm_blend(n));
        printf("throughput, ns per instruction over 8 independent chains: xmm vpternlogq %.3f  xmm vprolq %.3f\n", thr_xmm(n, 0), thr_xmm(n, 1));
    }
    return 0;
}
