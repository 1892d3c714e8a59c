/*
 * fz_oracle.c -- CPU restatement of the reference's algebra hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker.  The product path (libfusion_hip.so) never
 * links, loads or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function here
 * against golden vectors produced by importing the reference itself
 * (tests/golden/gen_golden.py, run in the dev container against /root/reference)
 * and against the reference's own reproducible KAT rows
 * (KATs/KAT_values/intermediate_hash_ch_KAT_128.csv pins the d=64 forward NTT).
 *
 * Each function cites the reference lines it follows (paths relative to the
 * reference checkout).  The arithmetic is deliberately naive: 64-bit products and
 * the C '%' operator, one centred reduction per reference `cent` call, loops in the
 * reference's order.
 *
 * Validity range (coefficients are stored as int32, as the HIP path stores them): every odd
 * modulus q < 2^32 and every int32 input, reduced or not.  The 64-bit products hold because
 * |a| <= 2^31 and a table entry s < 2^32 give |a * s| <= 2^63 - 2^31, and adding |u| <= 2^31
 * stays inside int64; the two places where an operand is a DIFFERENCE of coefficients
 * (the inverse butterfly's (u - v) * s, up to 2^32 * 2^32) go through __int128.  Pinned for
 * q in [2^31, 2^32) by tests/golden/generic.npz (reference outputs at q = 4294828033,
 * d = 256 and 2048) in tests/test_oracle_golden.py.  Moduli of 2^32 and more are the
 * pure-Python port's (oracle.py py_*: Python integers), pinned by the same file at q ~ 2^62.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* floor-mod like Python's % for positive q (algebra/ntt.py:120) */
static inline int64_t pymod(int64_t v, int64_t q) {
    int64_t y = v % q;
    return y < 0 ? y + q : y;
}

/* cent(): algebra/ntt.py:93-123.  y = val % q; z = y - q if y > q//2 else y
 * (the branch-free shift form at :121-122 selects exactly this for every odd q). */
ORC_API int64_t orc_cent(int64_t val, int64_t q) {
    int64_t y = pymod(val, q);
    return (y > q / 2) ? y - q : y;
}

/* the same for a product that may pass 2^63 */
static inline int64_t cent128(__int128 val, int64_t q) {
    int64_t y = (int64_t)(val % q);
    if (y < 0) y += q;
    return (y > q / 2) ? y - q : y;
}

ORC_API int64_t orc_powmod(int64_t b, int64_t e, int64_t q) {
    __int128 r = 1, x = pymod(b, q);
    while (e > 0) {
        if (e & 1) r = (r * x) % q;
        x = (x * x) % q;
        e >>= 1;
    }
    return (int64_t)r;
}

static unsigned bitrev(unsigned i, int k) {
    unsigned r = 0;
    for (int b = 0; b < k; ++b) r |= ((i >> b) & 1u) << (k - 1 - b);
    return r;
}

/* bit_reverse_copy([pow(root, i, q) for i in range(n)]):
 * algebra/ntt.py:74-90 with algebra/polynomials.py:396-397 / :416-417. */
ORC_API void orc_twiddle_table(int64_t root, int64_t q, int n, int64_t *out) {
    int k = 0;
    while ((1 << k) < n) ++k;
    for (int i = 0; i < n; ++i) out[i] = orc_powmod(root, bitrev((unsigned)i, k), q);
}

/* cooley_tukey_ntt loop: algebra/ntt.py:273-291 (natural in, bit-reversed out). */
ORC_API void orc_ntt_forward(int32_t *a, int n, int64_t q, const int64_t *tw) {
    int t = n, m = 1;
    while (m < n) {
        t /= 2;
        for (int i = 0; i < m; ++i) {
            int j1 = 2 * i * t, j2 = j1 + t - 1;
            int64_t s = tw[m + i];
            for (int j = j1; j <= j2; ++j) {
                int64_t u = a[j], v = (int64_t)a[j + t] * s;
                a[j] = (int32_t)orc_cent(u + v, q);
                a[j + t] = (int32_t)orc_cent(u - v, q);
            }
        }
        m *= 2;
    }
}

/* gentleman_sande_intt loop: algebra/ntt.py:352-377 (bit-reversed in, natural out,
 * final scale by n^{-1} = n^(q-2) mod q). */
ORC_API void orc_ntt_inverse(int32_t *a, int n, int64_t q, const int64_t *itw) {
    int64_t n_inv = orc_powmod(n, q - 2, q);
    int t = 1, m = n;
    while (m > 1) {
        int j1 = 0, h = m / 2;
        for (int i = 0; i < h; ++i) {
            int j2 = j1 + t - 1;
            int64_t s = itw[h + i];
            for (int j = j1; j <= j2; ++j) {
                int64_t u = a[j], v = a[j + t];
                a[j] = (int32_t)orc_cent(u + v, q);
                /* (u - v) * s: |u - v| < 2^32, s < q < 2^32 -> up to 2^64: through 128 bits */
                a[j + t] = (int32_t)cent128((__int128)(u - v) * s, q);
            }
            j1 += 2 * t;
        }
        t *= 2;
        m = h;
    }
    for (int j = 0; j < n; ++j) a[j] = (int32_t)orc_cent((int64_t)a[j] * n_inv, q);
}

/* batch forms: `batch` independent rows of n int32, row-major */
ORC_API void orc_ntt_forward_batch(int32_t *a, int64_t batch, int n, int64_t q, int64_t root) {
    int64_t *tw = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    orc_twiddle_table(root, q, n, tw);
    for (int64_t b = 0; b < batch; ++b) orc_ntt_forward(a + b * n, n, q, tw);
    free(tw);
}
ORC_API void orc_ntt_inverse_batch(int32_t *a, int64_t batch, int n, int64_t q, int64_t inv_root) {
    int64_t *tw = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    orc_twiddle_table(inv_root, q, n, tw);
    for (int64_t b = 0; b < batch; ++b) orc_ntt_inverse(a + b * n, n, q, tw);
    free(tw);
}

/* Pointwise ops on NTT-domain rows: PolynomialNTTRepresentation.__mul__ / __add__
 * (algebra/polynomials.py:376-384, :309-317); __neg__ (:325-333) returns -(x % q),
 * NOT centred; __sub__ = self + (-other) (:335-336). */
ORC_API void orc_pw_mul(const int32_t *a, const int32_t *b, int32_t *out, int64_t count, int64_t q) {
    for (int64_t i = 0; i < count; ++i) out[i] = (int32_t)orc_cent((int64_t)a[i] * b[i], q);
}
ORC_API void orc_pw_add(const int32_t *a, const int32_t *b, int32_t *out, int64_t count, int64_t q) {
    for (int64_t i = 0; i < count; ++i) out[i] = (int32_t)orc_cent((int64_t)a[i] + b[i], q);
}
ORC_API void orc_pw_neg(const int32_t *a, int32_t *out, int64_t count, int64_t q) {
    for (int64_t i = 0; i < count; ++i) out[i] = (int32_t)(-pymod(a[i], q));
}
ORC_API void orc_pw_sub(const int32_t *a, const int32_t *b, int32_t *out, int64_t count, int64_t q) {
    for (int64_t i = 0; i < count; ++i) out[i] = (int32_t)orc_cent((int64_t)a[i] - pymod(b[i], q), q);
}
/* acc = cent(acc + cent(a*b)): one GeneralMatrix.__mul__ inner step, algebra/matrices.py:127-129 */
ORC_API void orc_pw_mulacc(int32_t *acc, const int32_t *a, const int32_t *b, int64_t count, int64_t q) {
    for (int64_t i = 0; i < count; ++i)
        acc[i] = (int32_t)orc_cent((int64_t)acc[i] + orc_cent((int64_t)a[i] * b[i], q), q);
}

/* Coefficient-domain schoolbook negacyclic product:
 * PolynomialCoefficientRepresentation.__mul__, algebra/polynomials.py:196-208. */
ORC_API void orc_schoolbook_negacyclic(const int32_t *f, const int32_t *g, int32_t *out, int n, int64_t q) {
    __int128 *c = (__int128 *)calloc((size_t)(2 * n), sizeof(__int128));
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) c[i + j] += (__int128)f[i] * g[j];
    for (int k = 0; k < n; ++k) {
        __int128 v = c[k] - c[k + n];
        int64_t y = (int64_t)(v % q);
        if (y < 0) y += q;
        out[k] = (int32_t)((y > q / 2) ? y - q : y);
    }
    free(c);
}

/* (1 x l) . (l x 1) product of NTT-domain polynomials:
 * GeneralMatrix.__mul__, algebra/matrices.py:125-130 -- first term, then
 * `next_data += A[0][k] * S[k][0]`, every partial sum centred.
 * A: [l][n]; S: [batch][l][n]; out: [batch][n]. */
ORC_API void orc_matvec(const int32_t *A, const int32_t *S, int32_t *out,
                        int64_t batch, int l, int n, int64_t q) {
    for (int64_t b = 0; b < batch; ++b) {
        const int32_t *s = S + b * (int64_t)l * n;
        int32_t *o = out + b * n;
        for (int j = 0; j < n; ++j) o[j] = (int32_t)orc_cent((int64_t)A[j] * s[j], q);
        for (int k = 1; k < l; ++k)
            for (int j = 0; j < n; ++j)
                o[j] = (int32_t)orc_cent((int64_t)o[j] +
                                         orc_cent((int64_t)A[(int64_t)k * n + j] * s[(int64_t)k * n + j], q), q);
    }
}

/* keygen arithmetic: fusion/fusion.py:363-370.
 * coef: [batch][2][l][n] coefficient-domain secret polys (left then right);
 * sk_hat: same shape, NTT of each; vk: [batch][2][n] = A . sk_hat. */
ORC_API void orc_keygen_core(const int32_t *A, const int32_t *coef, int32_t *sk_hat, int32_t *vk,
                             int64_t batch, int l, int n, int64_t q, int64_t root) {
    int64_t total = batch * 2 * (int64_t)l * n;
    memcpy(sk_hat, coef, sizeof(int32_t) * (size_t)total);
    orc_ntt_forward_batch(sk_hat, batch * 2 * l, n, q, root);
    orc_matvec(A, sk_hat, vk, batch * 2, l, n, q);
}

/* sign arithmetic: sk.left_sk_hat * c_hat + sk.right_sk_hat, fusion/fusion.py:557
 * -> scalar product (matrices.py:109-114) then elementwise add (matrices.py:87-91).
 * L,R: [batch][l][n]; c_hat: [batch][n]; sig: [batch][l][n]. */
ORC_API void orc_sign_core(const int32_t *L, const int32_t *R, const int32_t *c_hat, int32_t *sig,
                           int64_t batch, int l, int n, int64_t q) {
    for (int64_t b = 0; b < batch; ++b)
        for (int k = 0; k < l; ++k)
            for (int j = 0; j < n; ++j) {
                int64_t idx = (b * l + k) * (int64_t)n + j;
                int64_t p = orc_cent((int64_t)L[idx] * c_hat[b * n + j], q);
                sig[idx] = (int32_t)orc_cent(p + R[idx], q);
            }
}

/* aggregate arithmetic: fusion/fusion.py:670-676.
 * sig: [N][l][n]; alpha_hat: [N][n]; out: [l][n] = sum_i sig_i (.) alpha_i, each
 * partial sum centred exactly as the `+=` chain does. */
ORC_API void orc_aggregate_core(const int32_t *sig, const int32_t *alpha_hat, int32_t *out,
                                int64_t N, int l, int n, int64_t q) {
    for (int k = 0; k < l; ++k)
        for (int j = 0; j < n; ++j) {
            int64_t acc = orc_cent((int64_t)sig[(int64_t)k * n + j] * alpha_hat[j], q);
            for (int64_t i = 1; i < N; ++i) {
                int64_t p = orc_cent((int64_t)sig[(i * l + k) * (int64_t)n + j] * alpha_hat[i * n + j], q);
                acc = orc_cent(acc + p, q);
            }
            out[(int64_t)k * n + j] = (int32_t)acc;
        }
}

/* norm("infty") and weight() of coefficient rows:
 * algebra/polynomials.py:221-227 -- max |x| over STORED values; #{x : x % q != 0}. */
ORC_API void orc_norm_weight(const int32_t *coef, int64_t batch, int n, int64_t q,
                             int64_t *max_abs, int32_t *weight) {
    for (int64_t b = 0; b < batch; ++b) {
        int64_t mx = 0;
        int32_t w = 0;
        for (int j = 0; j < n; ++j) {
            int64_t x = coef[b * n + j];
            int64_t ax = x < 0 ? -x : x;
            if (ax > mx) mx = ax;
            if (pymod(x, q) != 0) ++w;
        }
        max_abs[b] = mx;
        weight[b] = w;
    }
}

/* verify arithmetic: fusion/fusion.py:690-727.  Returns the reference's verdict:
 *   0 = (True, "")
 *   3 = "Target doesn't match image of aggregate signature."   (:721)
 *   4 = "Norm of aggregate signature too large."               (:725)
 *   5 = "Weight of aggregate signature too large."             (:727)
 * (codes 1, 2 -- "Too many keys." / "Number of keys and messages must be equal." --
 *  are host-side length checks, :686-689.)
 * A: [l][n]; sig: [l][n] aggregate; vkL, vkR, c_hat, alpha_hat: [N][n]. */
ORC_API int orc_verify_core(const int32_t *A, const int32_t *sig,
                            const int32_t *vkL, const int32_t *vkR,
                            const int32_t *c_hat, const int32_t *alpha_hat,
                            int64_t N, int l, int n, int64_t q, int64_t inv_root,
                            int64_t beta_vf, int64_t omega_vf) {
    /* target = sum_i (vkL_i * c_i + vkR_i) * alpha_i   (:706-714) */
    int32_t *target = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    for (int j = 0; j < n; ++j) {
        int64_t acc = 0;
        for (int64_t i = 0; i < N; ++i) {
            int64_t t = orc_cent((int64_t)vkL[i * n + j] * c_hat[i * n + j], q);
            t = orc_cent(t + vkR[i * n + j], q);
            t = orc_cent(t * alpha_hat[i * n + j], q);
            acc = (i == 0) ? t : orc_cent(acc + t, q);
        }
        target[j] = (int32_t)acc;
    }
    /* observed = A . sig  (:715-717) */
    int32_t *observed = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    orc_matvec(A, sig, observed, 1, l, n, q);
    int mismatch = 0;
    for (int j = 0; j < n; ++j)
        if (pymod((int64_t)target[j] - observed[j], q) != 0) mismatch = 1;   /* __eq__ is mod q, polynomials.py:278-280 */
    free(target);
    free(observed);
    if (mismatch) return 3;
    /* coefficient representation of the aggregate (:690-692), then bounds (:722-727) */
    int32_t *coef = (int32_t *)malloc(sizeof(int32_t) * (size_t)l * n);
    memcpy(coef, sig, sizeof(int32_t) * (size_t)l * n);
    orc_ntt_inverse_batch(coef, l, n, q, inv_root);
    int64_t *mx = (int64_t *)malloc(sizeof(int64_t) * (size_t)l);
    int32_t *wt = (int32_t *)malloc(sizeof(int32_t) * (size_t)l);
    orc_norm_weight(coef, l, n, q, mx, wt);
    int rc = 0;
    for (int k = 0; k < l && rc == 0; ++k) if (mx[k] > beta_vf) rc = 4;
    if (rc == 0) for (int k = 0; k < l && rc == 0; ++k) if (wt[k] > omega_vf) rc = 5;
    free(coef); free(mx); free(wt);
    return rc;
}
