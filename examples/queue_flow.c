/* The asynchronous batch queue from plain C through the C ABI only (no Python, no HIP headers): CALLS calls of PER keys +
 * signatures each (keygen + sign, fusion/fusion.py:338-373, :534-557) are submitted back to back from this one thread, worker
 * threads inside the library run whatever is pending as one batch, and every call's verification keys and signatures must be
 * the rows the direct entry points (fz_sample_secret_polys_dev + fz_keygen_core_bcast + fz_challenge_hat_msgs_dev + fz_sign_core)
 * give for that call alone.  Then every call's signers are aggregated and verified THROUGH THE SAME QUEUE
 * (fz_queue_submit_aggregate_verify on the queue's own device rows: aggregate() + verify(), fusion.py:655-677, :680-728 -- the twelve
 * pending calls share one ragged launch for the sums and one for the verdicts), and a tampered aggregate is rejected by a queued
 * verify().
 *   gcc -std=c99 -Iinclude examples/queue_flow.c -o queue_flow -Lfusion-cryptography_amd/lib -lfusion_hip \
 *       -Wl,-rpath,$PWD/fusion-cryptography_amd/lib
 * Exit code 0 = every row equal.  (tests/test_cabi_symbols.py compiles it; tests/test_gpu_queue.py runs it.) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fusion_hip.h"

#define CHECK(call)                                                        \
    do {                                                                   \
        int rc_ = (call);                                                  \
        if (rc_ != FZ_OK) {                                                \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, fz_last_error()); \
            return 1;                                                      \
        }                                                                  \
    } while (0)

enum { CALLS = 12, PER = 40, D = 64, L = 195 };          /* secpar 128: degree 64, rank 195 */

int main(void) {
    fz_scheme_params P;
    const int64_t beta_sk = 52, omega_sk = D;            /* fusion/fusion.py:30, :87-90 (secpar 128) */
    fz_ctx *ctx = NULL;
    fz_queue *q = NULL;
    static int32_t A[L][D], vk_ref[PER][2][D];
    static char msgs[CALLS][PER * 24];
    static size_t off[CALLS][PER + 1];
    static uint64_t seeds[CALLS][PER];
    int32_t *vk_out[CALLS], *sig_ref, *sig_got;
    uint64_t ticket[CALLS], agg_ticket[CALLS], bad_ticket = 0, done_calls = 0, batches = 0, rows = 0;
    const int32_t *sig_rows[CALLS];
    static int32_t agg[CALLS][L][D];
    int verdict[CALLS], bad_verdict = -1;
    void *d_A = NULL, *d_coef = NULL, *d_sk = NULL, *d_vk = NULL, *d_c = NULL, *d_sig = NULL;
    size_t i;
    int c, k;

    memset(&P, 0, sizeof P);
    P.modulus = 2147465729; P.root = 23584283; P.inv_root = 540632852;
    P.degree = D; P.root_order = 2 * D; P.secpar = 128;
    P.omega_ch = 27; P.omega_ag = 35; P.beta_ch = 1; P.beta_ag = 1;
    P.bytes_for_one_coef_bdd_by_beta_ch = 17;            /* ceil(ceil(log2(3) / 8) + 128 / 8) */
    P.bytes_for_poly_shuffle = D * 17;
    P.sign_pre_hash_dst[0] = 1; P.sign_pre_hash_dst[1] = 0;
    P.sign_hash_dst[0] = 1;     P.sign_hash_dst[1] = 1;
    P.agg_xof_dst[0] = 1;       P.agg_xof_dst[1] = 2;

    for (k = 0; k < L; ++k) CHECK(fz_sample_ntt_values(7000u + (uint64_t)k, P.modulus, D, A[k]));   /* public challenge */
    for (c = 0; c < CALLS; ++c) {
        off[c][0] = 0;
        for (i = 0; i < PER; ++i) {
            seeds[c][i] = 900000u + 1000u * (uint64_t)c + 2u * (uint64_t)i;
            off[c][i + 1] = off[c][i] + (size_t)sprintf(msgs[c] + off[c][i], "call %d msg %zu", c, i);
        }
    }
    /* the queue: two workers, batches of at most 256 rows (so the 12 calls of 40 rows need several batches) */
    CHECK(fz_queue_create(0, &P, L, beta_sk, omega_sk, &A[0][0], 2, 256, &q));
    for (c = 0; c < CALLS; ++c) {
        CHECK(fz_pinned_alloc(sizeof vk_ref, (void **)&vk_out[c]));
        CHECK(fz_queue_submit_keygen_sign(q, seeds[c], PER, msgs[c], off[c], vk_out[c], 0, &ticket[c]));
    }
    /* the same calls one by one through the direct entry points, on a context of our own */
    CHECK(fz_ctx_create(0, (uint32_t)P.modulus, D, (uint32_t)P.root, (uint32_t)P.inv_root, &ctx));
    CHECK(fz_malloc(ctx, sizeof A, &d_A));
    CHECK(fz_malloc(ctx, sizeof vk_ref, &d_coef));
    CHECK(fz_malloc(ctx, (size_t)PER * 2 * L * D * 4, &d_sk));
    CHECK(fz_malloc(ctx, sizeof vk_ref, &d_vk));
    CHECK(fz_malloc(ctx, (size_t)PER * D * 4, &d_c));
    CHECK(fz_malloc(ctx, (size_t)PER * L * D * 4, &d_sig));
    CHECK(fz_memcpy_h2d(ctx, d_A, A, sizeof A));
    sig_ref = (int32_t *)malloc((size_t)PER * L * D * 4);
    sig_got = (int32_t *)malloc((size_t)PER * L * D * 4);
    if (!sig_ref || !sig_got) return 1;
    for (c = 0; c < CALLS; ++c) {
        fz_queue_result r;
        CHECK(fz_sample_secret_polys_dev(ctx, seeds[c], PER, P.modulus, D, beta_sk, omega_sk, (int32_t *)d_coef));
        CHECK(fz_keygen_core_bcast(ctx, (const int32_t *)d_A, (const int32_t *)d_coef, (int32_t *)d_sk, (int32_t *)d_vk, PER, L));
        CHECK(fz_challenge_hat_msgs_dev(ctx, &P, (const int32_t *)d_vk, msgs[c], off[c], PER, (int32_t *)d_c, NULL));
        CHECK(fz_sign_core(ctx, (const int32_t *)d_sk, (const int32_t *)d_c, (int32_t *)d_sig, PER, L));
        CHECK(fz_memcpy_d2h(ctx, vk_ref, d_vk, sizeof vk_ref));
        CHECK(fz_memcpy_d2h(ctx, sig_ref, d_sig, (size_t)PER * L * D * 4));
        CHECK(fz_queue_wait(q, ticket[c], &r));
        if (r.status != FZ_OK || r.n != PER || !r.d_sig || r.d_sk_hat) { fprintf(stderr, "call %d: unexpected result\n", c); return 1; }
        CHECK(fz_memcpy_d2h(ctx, sig_got, r.d_sig, (size_t)PER * L * D * 4));      /* the queue's rows, read through OUR context */
        if (memcmp(vk_out[c], vk_ref, sizeof vk_ref) != 0) { fprintf(stderr, "call %d: verification keys differ\n", c); return 1; }
        if (memcmp(sig_got, sig_ref, (size_t)PER * L * D * 4) != 0) { fprintf(stderr, "call %d: signatures differ\n", c); return 1; }
        sig_rows[c] = r.d_sig;                             /* stays the queue's until the call is released (below) */
    }
    /* aggregate() + verify() of every call's signers, queued: verification bounds and capacity of secpar 128 (fusion.py:24, :63-68) */
    CHECK(fz_queue_enable_aggregate(q, 536070080, D, 1796, 4));
    for (c = 0; c < CALLS; ++c) {
        verdict[c] = -1;
        CHECK(fz_queue_submit_aggregate_verify(q, vk_out[c], msgs[c], off[c], PER, sig_rows[c], &agg[c][0][0], &verdict[c],
                                               FZ_QUEUE_ROWS_ON_DEVICE, &agg_ticket[c]));
    }
    for (c = 0; c < CALLS; ++c) {
        CHECK(fz_queue_wait(q, agg_ticket[c], NULL));
        if (verdict[c] != FZ_VERDICT_OK) { fprintf(stderr, "call %d: aggregate rejected with verdict %d\n", c, verdict[c]); return 1; }
    }
    agg[3][7][5] += 1;                                     /* a tampered aggregate: the reference's tamper test (tests/test_fusion.py:860-873) */
    CHECK(fz_queue_submit_verify(q, vk_out[3], msgs[3], off[3], PER, &agg[3][0][0], &bad_verdict, 0, &bad_ticket));
    CHECK(fz_queue_wait(q, bad_ticket, NULL));
    if (bad_verdict != FZ_VERDICT_TARGET_MISMATCH) { fprintf(stderr, "tampered aggregate: verdict %d\n", bad_verdict); return 1; }
    for (c = 0; c < CALLS; ++c) {
        CHECK(fz_queue_release(q, ticket[c]));
        CHECK(fz_queue_release(q, ticket[c]));             /* idempotent */
    }
    CHECK(fz_queue_drain(q));
    CHECK(fz_queue_stats(q, &done_calls, &batches, &rows));
    if (done_calls != 2 * CALLS + 1 || rows != (uint64_t)(2 * CALLS + 1) * PER || batches == 0 || batches > 2 * CALLS + 1) { fprintf(stderr, "stats %llu %llu %llu\n",
        (unsigned long long)done_calls, (unsigned long long)batches, (unsigned long long)rows); return 1; }
    CHECK(fz_queue_destroy(q));
    for (c = 0; c < CALLS; ++c) CHECK(fz_pinned_free(vk_out[c]));
    free(sig_ref); free(sig_got);
    CHECK(fz_free(ctx, d_A)); CHECK(fz_free(ctx, d_coef)); CHECK(fz_free(ctx, d_sk)); CHECK(fz_free(ctx, d_vk));
    CHECK(fz_free(ctx, d_c)); CHECK(fz_free(ctx, d_sig));
    CHECK(fz_ctx_destroy(ctx));
    printf("queue_flow OK: %d calls of %d keys + signatures, their %d aggregates + verifications and one tampered verification in %llu "
           "batches; every row equal to the direct calls, every verdict the expected one\n", CALLS, PER, CALLS, (unsigned long long)batches);
    return 0;
}
