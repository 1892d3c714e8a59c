/* A C caller of libfusion_hip.so: forward + inverse transform of a small batch through the device-pointer entry
 * points, replayed from a captured graph, checked against the input.  Plain C99, no HIP headers:
 *   gcc -std=c99 -Iinclude examples/roundtrip.c -o roundtrip -Lfusion-cryptography_amd/lib -lfusion_hip \
 *       -Wl,-rpath,$PWD/fusion-cryptography_amd/lib
 * Exit code 0 = round trip exact.  (tests/test_cabi_symbols.py compiles it; tests/test_gpu_ntt.py runs it.) */
#include <stdio.h>
#include <stdlib.h>
#include "fusion_hip.h"

#define CHECK(call)                                                        \
    do {                                                                   \
        int rc_ = (call);                                                  \
        if (rc_ != FZ_OK) {                                                \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, fz_last_error()); \
            return 1;                                                      \
        }                                                                  \
    } while (0)

int main(void) {
    const uint32_t q = 2147465729u, root = 3337519u, inv_root = 1978410468u;   /* secpar 256 (fusion/fusion.py:71-96) */
    const int degree = 256;
    const size_t rows = 1000, n = rows * (size_t)degree;
    fz_ctx *ctx = NULL;
    fz_graph *graph = NULL;
    void *stream = NULL, *d_x = NULL, *d_y = NULL, *d_z = NULL;
    int32_t *x = malloc(n * sizeof(int32_t)), *z = malloc(n * sizeof(int32_t));
    uint64_t s = 88172645463325252ull;
    size_t i, bad = 0;
    if (!x || !z) return 2;
    for (i = 0; i < n; ++i) {                     /* xorshift64: centred residues */
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        x[i] = (int32_t)((int64_t)(s % q) - (int64_t)(q / 2));
    }
    CHECK(fz_ctx_create(0, q, degree, root, inv_root, &ctx));
    CHECK(fz_stream_create(ctx, &stream));
    CHECK(fz_ctx_set_stream(ctx, stream));
    CHECK(fz_malloc(ctx, n * sizeof(int32_t), &d_x));
    CHECK(fz_malloc(ctx, n * sizeof(int32_t), &d_y));
    CHECK(fz_malloc(ctx, n * sizeof(int32_t), &d_z));
    CHECK(fz_memcpy_h2d(ctx, d_x, x, n * sizeof(int32_t)));
    CHECK(fz_graph_begin(ctx));
    CHECK(fz_ntt_forward(ctx, (const int32_t *)d_x, (int32_t *)d_y, rows));
    CHECK(fz_ntt_inverse(ctx, (const int32_t *)d_y, (int32_t *)d_z, rows));
    CHECK(fz_graph_end(ctx, &graph));
    CHECK(fz_graph_launch(ctx, graph));
    CHECK(fz_memcpy_d2h(ctx, z, d_z, n * sizeof(int32_t)));      /* synchronises the stream */
    for (i = 0; i < n; ++i) bad += (x[i] != z[i]);
    printf("%s: INTT(NTT(x)) %s x on %zu rows of degree %d (%zu mismatches)\n", fz_version(), bad ? "!=" : "==", rows,
           degree, bad);
    CHECK(fz_graph_destroy(graph));
    CHECK(fz_free(ctx, d_x));
    CHECK(fz_free(ctx, d_y));
    CHECK(fz_free(ctx, d_z));
    CHECK(fz_ctx_set_stream(ctx, NULL));
    CHECK(fz_stream_destroy(ctx, stream));
    CHECK(fz_ctx_destroy(ctx));
    free(x);
    free(z);
    return bad ? 3 : 0;
}
