/* The scheme's flow from plain C through the C ABI only (no Python, no HIP headers): keygen -> sign -> aggregate -> verify
 * for N one-time keys at secpar 256 (fusion/fusion.py:338-373, :534-557, :655-677, :680-728).
 *   keys        fz_sample_secret_polys_dev (the reference's seeded MT19937 sampler, on the device) + fz_keygen_core_bcast
 *   challenges  fz_challenge_hat_msgs_dev (SHA3-256 of the messages, text of str(vk), SHAKE-256, decoder, NTT: device)
 *   signatures  fz_sign_core
 *   aggregator  fz_sort_by_vk_string + fz_aggregation_coefficients (hash_ag: ONE serial XOF, host) + fz_ntt_forward_host,
 *               fz_aggregate_core
 *   verifier    fz_verify_core -> verdict code; a tampered aggregate must be rejected
 *   many aggregates in one launch: fz_aggregate_core_ragged over signer blocks of different sizes (linearity check)
 *   signing service: fz_sign_aggregate_target_partial_batch (signatures + the aggregate's and the target's sums in one pass)
 *   gcc -std=c99 -Iinclude examples/scheme_flow.c -o scheme_flow -Lfusion-cryptography_amd/lib -lfusion_hip \
 *       -Wl,-rpath,$PWD/fusion-cryptography_amd/lib
 * Exit code 0 = the aggregate verifies and the tampered one does not.  (tests/test_cabi_symbols.py compiles it;
 * tests/test_gpu_scheme.py runs it.) */
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

enum { N = 64, D = 256, L = 83 };

int main(void) {
    /* the secpar-256 parameter set (fusion/fusion.py:25-68, :97-141) */
    fz_scheme_params P;
    const int64_t beta_sk = 52, omega_sk = D, beta_vf = 536321760, omega_vf = D;
    fz_ctx *ctx = NULL;
    void *d_A = NULL, *d_coef = NULL, *d_sk = NULL, *d_vk = NULL, *d_c = NULL, *d_sig = NULL, *d_al = NULL, *d_agg = NULL,
         *d_vkL = NULL, *d_vkR = NULL;
    static int32_t A[L][D], vk[N][2][D], vkL[N][D], vkR[N][D], sL[N][D], sR[N][D], c_hat[N][D], sc[N][D], al_sorted[N][D],
                   al[N][D], agg[L][D];
    static uint8_t pre[N][32], spre[N][32];
    static char msgs[N * 32];
    size_t off[N + 1], order[N], i;
    uint64_t seeds[N];
    int k, verdict = -1, tampered = -1;

    memset(&P, 0, sizeof P);
    P.modulus = 2147465729; P.root = 3337519; P.inv_root = 1978410468;
    P.degree = D; P.root_order = 2 * D; P.secpar = 256;
    P.omega_ch = 60; P.omega_ag = 60; P.beta_ch = 1; P.beta_ag = 1;
    P.bytes_for_one_coef_bdd_by_beta_ch = 33;           /* ceil(ceil(log2(3) / 8) + 256 / 8) */
    P.bytes_for_poly_shuffle = D * 33;                   /* degree * ceil(ceil(log2(degree) / 8) + 256 / 8) */
    P.sign_pre_hash_dst[0] = 3; P.sign_pre_hash_dst[1] = 0;
    P.sign_hash_dst[0] = 3;     P.sign_hash_dst[1] = 1;
    P.agg_xof_dst[0] = 3;       P.agg_xof_dst[1] = 2;

    CHECK(fz_ctx_create(0, (uint32_t)P.modulus, D, (uint32_t)P.root, (uint32_t)P.inv_root, &ctx));
    for (k = 0; k < L; ++k) CHECK(fz_sample_ntt_values(1000u + (uint64_t)k, P.modulus, D, A[k]));   /* public challenge */
    off[0] = 0;
    for (i = 0; i < N; ++i) {
        seeds[i] = 424242u + 2u * (uint64_t)i;
        off[i + 1] = off[i] + (size_t)sprintf(msgs + off[i], "message %zu", i);
    }
    CHECK(fz_malloc(ctx, sizeof A, &d_A));
    CHECK(fz_malloc(ctx, sizeof vk, &d_coef));
    CHECK(fz_malloc(ctx, (size_t)N * 2 * L * D * 4, &d_sk));
    CHECK(fz_malloc(ctx, sizeof vk, &d_vk));
    CHECK(fz_malloc(ctx, sizeof c_hat, &d_c));
    CHECK(fz_malloc(ctx, (size_t)N * L * D * 4, &d_sig));
    CHECK(fz_malloc(ctx, sizeof al, &d_al));
    CHECK(fz_malloc(ctx, sizeof agg, &d_agg));
    CHECK(fz_malloc(ctx, sizeof vkL, &d_vkL));
    CHECK(fz_malloc(ctx, sizeof vkR, &d_vkR));
    CHECK(fz_memcpy_h2d(ctx, d_A, A, sizeof A));

    /* signers: keys, challenges, signatures -- nothing but seeds, message bytes and digests crosses PCIe */
    CHECK(fz_sample_secret_polys_dev(ctx, seeds, N, P.modulus, D, beta_sk, omega_sk, (int32_t *)d_coef));
    CHECK(fz_keygen_core_bcast(ctx, (const int32_t *)d_A, (const int32_t *)d_coef, (int32_t *)d_sk, (int32_t *)d_vk, N, L));
    CHECK(fz_challenge_hat_msgs_dev(ctx, &P, (const int32_t *)d_vk, msgs, off, N, (int32_t *)d_c, &pre[0][0]));
    CHECK(fz_sign_core(ctx, (const int32_t *)d_sk, (const int32_t *)d_c, (int32_t *)d_sig, N, L));

    /* aggregator: the aggregation coefficients come from ONE hash over the key-sorted list (hash_ag) */
    CHECK(fz_memcpy_d2h(ctx, vk, d_vk, sizeof vk));
    CHECK(fz_memcpy_d2h(ctx, c_hat, d_c, sizeof c_hat));
    for (i = 0; i < N; ++i) {
        memcpy(vkL[i], vk[i][0], sizeof vkL[i]);
        memcpy(vkR[i], vk[i][1], sizeof vkR[i]);
    }
    CHECK(fz_sort_by_vk_string(&P, &vkL[0][0], &vkR[0][0], N, order, 4));
    for (i = 0; i < N; ++i) {
        memcpy(sL[i], vkL[order[i]], sizeof sL[i]);
        memcpy(sR[i], vkR[order[i]], sizeof sR[i]);
        memcpy(sc[i], c_hat[order[i]], sizeof sc[i]);
        memcpy(spre[i], pre[order[i]], 32);
    }
    CHECK(fz_aggregation_coefficients(&P, &sL[0][0], &sR[0][0], &spre[0][0], &sc[0][0], N, &al_sorted[0][0], 4));
    for (i = 0; i < N; ++i) memcpy(al[order[i]], al_sorted[i], sizeof al[i]);     /* back to the signers' order */
    CHECK(fz_ntt_forward_host(ctx, &al[0][0], N));
    CHECK(fz_memcpy_h2d(ctx, d_al, al, sizeof al));
    CHECK(fz_aggregate_core(ctx, (const int32_t *)d_sig, (const int32_t *)d_al, (int32_t *)d_agg, N, L));

    /* verifier */
    CHECK(fz_memcpy_h2d(ctx, d_vkL, vkL, sizeof vkL));
    CHECK(fz_memcpy_h2d(ctx, d_vkR, vkR, sizeof vkR));
    CHECK(fz_verify_core(ctx, (const int32_t *)d_A, (const int32_t *)d_agg, (const int32_t *)d_vkL, (const int32_t *)d_vkR,
                         (const int32_t *)d_c, (const int32_t *)d_al, N, L, beta_vf, omega_vf, &verdict));
    CHECK(fz_memcpy_d2h(ctx, agg, d_agg, sizeof agg));
    {   /* blocks of 24 and 40 signers as two aggregates of ONE launch: with the same coefficients their sum is the aggregate */
        static int32_t two[2][L][D];
        const size_t blocks[3] = {0, 24, N};
        void *d_two = NULL;
        int j;
        CHECK(fz_malloc(ctx, sizeof two, &d_two));
        CHECK(fz_aggregate_core_ragged(ctx, (const int32_t *)d_sig, (const int32_t *)d_al, blocks, 2, L, (int32_t *)d_two));
        CHECK(fz_memcpy_d2h(ctx, two, d_two, sizeof two));
        CHECK(fz_free(ctx, d_two));
        for (k = 0; k < L; ++k)
            for (j = 0; j < D; ++j)
                if (((int64_t)two[0][k][j] + two[1][k][j] - agg[k][j]) % P.modulus != 0) {
                    fprintf(stderr, "ragged aggregation: block sums differ from the aggregate at [%d][%d]\n", k, j);
                    return 4;
                }
    }
    {   /* a service that signs AND aggregates: one pass (fz_sign_aggregate_target_partial_batch) -- the same signatures, and int64
         * sums of the aggregate and of the verification target from which the verdict comes without a centring pass */
        static int32_t sig_a[N][L][D], sig_b[N][L][D];
        void *d_sig2 = NULL, *d_sums = NULL, *d_verd = NULL;
        int v = -1;
        CHECK(fz_malloc(ctx, sizeof sig_b, &d_sig2));
        CHECK(fz_malloc(ctx, (size_t)(L * D + D) * 8, &d_sums));
        CHECK(fz_malloc(ctx, sizeof v, &d_verd));
        CHECK(fz_sign_aggregate_target_partial_batch(ctx, (const int32_t *)d_sk, (const int32_t *)d_c, (const int32_t *)d_al,
                                                     (const int32_t *)d_vkL, (const int32_t *)d_vkR, (int32_t *)d_sig2,
                                                     (int64_t *)d_sums, (size_t)L * D, (int64_t *)d_sums + (size_t)L * D, D, 1, N, L));
        CHECK(fz_verify_partials_batch_async(ctx, (const int32_t *)d_A, (const int64_t *)d_sums, (size_t)L * D,
                                             (const int64_t *)d_sums + (size_t)L * D, D, 1, L, beta_vf, omega_vf, (int *)d_verd));
        CHECK(fz_memcpy_d2h(ctx, sig_a, d_sig, sizeof sig_a));
        CHECK(fz_memcpy_d2h(ctx, sig_b, d_sig2, sizeof sig_b));
        CHECK(fz_memcpy_d2h(ctx, &v, d_verd, sizeof v));
        CHECK(fz_free(ctx, d_sig2)); CHECK(fz_free(ctx, d_sums)); CHECK(fz_free(ctx, d_verd));
        if (memcmp(sig_a, sig_b, sizeof sig_a) != 0 || v != FZ_VERDICT_OK) {
            fprintf(stderr, "one-pass signing + aggregation: signatures differ or verdict %d\n", v);
            return 5;
        }
    }
    agg[L - 1][D - 1] += 1;
    CHECK(fz_memcpy_h2d(ctx, d_agg, agg, sizeof agg));
    CHECK(fz_verify_core(ctx, (const int32_t *)d_A, (const int32_t *)d_agg, (const int32_t *)d_vkL, (const int32_t *)d_vkR,
                         (const int32_t *)d_c, (const int32_t *)d_al, N, L, beta_vf, omega_vf, &tampered));
    printf("%s: %d signatures aggregated; verdict %d (0 = accepted), tampered aggregate: verdict %d (%d = target mismatch)\n",
           fz_version(), (int)N, verdict, tampered, FZ_VERDICT_TARGET_MISMATCH);
    CHECK(fz_free(ctx, d_A)); CHECK(fz_free(ctx, d_coef)); CHECK(fz_free(ctx, d_sk)); CHECK(fz_free(ctx, d_vk));
    CHECK(fz_free(ctx, d_c)); CHECK(fz_free(ctx, d_sig)); CHECK(fz_free(ctx, d_al)); CHECK(fz_free(ctx, d_agg));
    CHECK(fz_free(ctx, d_vkL)); CHECK(fz_free(ctx, d_vkR));
    CHECK(fz_ctx_destroy(ctx));
    return (verdict == FZ_VERDICT_OK && tampered == FZ_VERDICT_TARGET_MISMATCH) ? 0 : 3;
}
