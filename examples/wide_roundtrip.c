/* A C caller of the GENERIC-parameter path of libfusion_hip.so (fz_wide_*: any odd modulus below 2^63, any power-of-two
 * length, int64 rows on the host): the forward transform of a few rows over a 62-bit prime, checked against a direct
 * evaluation of the definition -- NTT(x)[brv(k)] = sum_j x[j] * psi^((2k+1) j) (algebra/ntt.py:216-291: natural order in,
 * bit-reversed out) -- then the inverse back to the input, the pointwise product and the reference's negation.
 * Plain C99 (the host-side check multiplies mod q by shift-and-add: no 128-bit type needed), no HIP headers:
 *   gcc -std=c99 -Iinclude examples/wide_roundtrip.c -o wide_roundtrip -Lfusion-cryptography_amd/lib -lfusion_hip \
 *       -Wl,-rpath,$PWD/fusion-cryptography_amd/lib
 * Exit code 0 = every value equal.  (tests/test_cabi_symbols.py compiles it; tests/test_gpu_wide.py runs it.) */
#include <stdio.h>
#include <stdlib.h>
#include "fusion_hip_generic.h"

#define CHECK(call)                                                        \
    do {                                                                   \
        int rc_ = (call);                                                  \
        if (rc_ != FZ_OK) {                                                \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, fz_last_error()); \
            return 1;                                                      \
        }                                                                  \
    } while (0)

static const uint64_t Q = 4611686018427322369ull;         /* 2^62 - 65535: prime, Q - 1 divisible by 2^16 */

static uint64_t mulq(uint64_t a, uint64_t b) {             /* a * b mod Q for a, b < Q < 2^62: shift and add, sums stay below 2^63 */
    uint64_t r = 0;
    for (; b; b >>= 1, a = (a << 1) % Q)
        if (b & 1) r = (r + a) % Q;
    return r;
}
static uint64_t powq(uint64_t a, uint64_t e) {
    uint64_t r = 1;
    for (; e; e >>= 1, a = mulq(a, a))
        if (e & 1) r = mulq(r, a);
    return r;
}
static int64_t cent(uint64_t v) { return v > (Q - 1) / 2 ? (int64_t)v - (int64_t)Q : (int64_t)v; }
static uint64_t canon(int64_t x) { return x < 0 ? (uint64_t)(x + (int64_t)Q) : (uint64_t)x; }
static unsigned brv(unsigned i, int bits) {
    unsigned r = 0;
    int b;
    for (b = 0; b < bits; ++b) r |= ((i >> b) & 1u) << (bits - 1 - b);
    return r;
}

enum { D = 64, LOGD = 6, ROWS = 3 };

int main(void) {
    static uint64_t tab[D], itab[D];
    static int64_t x[ROWS][D], y[ROWS][D], z[ROWS][D], p[ROWS][D], neg[ROWS][D];
    uint64_t g, psi = 0, ipsi, s = 88172645463325252ull;
    int r, j, k;
    size_t bad = 0;
    for (g = 2; g < 1000 && !psi; ++g) {                   /* a primitive 2D-th root of unity */
        const uint64_t c = powq(g, (Q - 1) / (2 * D));
        if (powq(c, D) == Q - 1) psi = c;
    }
    if (!psi) return 2;
    ipsi = powq(psi, Q - 2);
    for (j = 0; j < D; ++j) {                              /* bit_reverse_copy of the powers (polynomials.py:396-397) */
        tab[brv((unsigned)j, LOGD)] = powq(psi, (uint64_t)j);
        itab[brv((unsigned)j, LOGD)] = powq(ipsi, (uint64_t)j);
    }
    for (r = 0; r < ROWS; ++r)
        for (j = 0; j < D; ++j) {                          /* xorshift64: centred residues */
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            x[r][j] = cent(s % Q);
        }
    x[0][0] = (int64_t)((Q - 1) / 2);
    x[0][1] = -(int64_t)((Q - 1) / 2);
    CHECK(fz_wide_ntt_host(0, Q, D, tab, 0, 0, &x[0][0], &y[0][0], ROWS));
    for (r = 0; r < ROWS; ++r)
        for (k = 0; k < D; ++k) {
            uint64_t acc = 0;
            const uint64_t w = powq(psi, (uint64_t)(2 * k + 1));
            uint64_t wj = 1;
            for (j = 0; j < D; ++j, wj = mulq(wj, w)) acc = (acc + mulq(canon(x[r][j]), wj)) % Q;
            if (y[r][brv((unsigned)k, LOGD)] != cent(acc)) ++bad;
        }
    CHECK(fz_wide_ntt_host(0, Q, D, itab, powq(D, Q - 2), 1, &y[0][0], &z[0][0], ROWS));
    CHECK(fz_wide_pw_host(0, Q, 0 /* mul */, &x[0][0], &y[0][0], &p[0][0], (size_t)ROWS * D));
    CHECK(fz_wide_pw_host(0, Q, 3 /* neg */, &x[0][0], NULL, &neg[0][0], (size_t)ROWS * D));
    for (r = 0; r < ROWS; ++r)
        for (j = 0; j < D; ++j) {
            if (z[r][j] != x[r][j]) ++bad;
            if (p[r][j] != cent(mulq(canon(x[r][j]), canon(y[r][j])))) ++bad;
            if (neg[r][j] != -(int64_t)canon(x[r][j])) ++bad;
        }
    if (bad) { fprintf(stderr, "%zu values differ\n", bad); return 1; }
    printf("wide_roundtrip OK: %d rows of length %d over q = %llu: transform == definition, inverse == input, product, negation\n",
           ROWS, D, (unsigned long long)Q);
    return 0;
}
