/* fusion_hip_generic.h -- the GENERIC-PARAMETER path of libfusion_hip.so: a correctness path, not for throughput.
 *
 * The reference transforms any power-of-two length over any odd modulus with the table it is handed (algebra/ntt.py:239-290,
 * :345-377) and its polynomial classes add, negate and multiply over any modulus (algebra/polynomials.py:140-216, :272-333).
 * The int32 entry points of fusion_hip.h cover q < 2^32 and lengths <= 4096 (every parameter set of the scheme) at full speed;
 * the four entries here cover the rest, so that the drop-in packages refuse nothing below 2^63 and compute nothing on the CPU.
 * They are kept out of fusion_hip.h because they do NOT follow that header's conventions: no fz_ctx (a device index), host
 * pointers only, device buffers allocated and freed inside every call, synchronous copies on the null stream, one workgroup per
 * row with every stage through global memory.  Bind them for completeness of the parameter space; bind fusion_hip.h for work.
 *
 * Rows are int64 on the host, in and out.  ANY int64 value is accepted as input and reduced mod q first (the int32 entry
 * points accept unreduced rows in the same way); outputs are centred residues, |x| <= (q-1)/2, except neg (below).
 */
#ifndef FUSION_HIP_GENERIC_H
#define FUSION_HIP_GENERIC_H
#include "fusion_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* h_table: `degree` residues in [0, q), used exactly as the reference uses its `bit_rev_root_powers` /
 * `bit_rev_inv_root_powers` argument (entry m + i for block i of the stage with m blocks); n_inv: degree^{-1} mod q (inverse
 * only).  At most 2^31 - 1 rows per call. */
FZ_API int fz_wide_ntt_host(int device, uint64_t q, int degree, const uint64_t *h_table, uint64_t n_inv, int inverse,
                            const int64_t *h_in, int64_t *h_out, size_t batch);
/* op as for fz_pw_binary_host (0 mul, 1 add, 2 sub, 3 neg); neg returns -(x mod q) in [-(q-1), 0] as the reference's __neg__ does */
FZ_API int fz_wide_pw_host(int device, uint64_t q, int op, const int64_t *h_a, const int64_t *h_b, int64_t *h_out, size_t count);
/* out[b][j] = cent(sum_k A[k][j] * S[b][k][j]): A [l][degree], S [batch][l][degree] (algebra/matrices.py:143-181) */
FZ_API int fz_wide_matvec_host(int device, uint64_t q, int degree, const int64_t *h_A, const int64_t *h_S, int64_t *h_out,
                               size_t batch, int l);
/* per row: max |x| over the STORED values (as an unsigned 64-bit number: |INT64_MIN| = 2^63 is representable) and the number of
 * non-zero values (algebra/polynomials.py:221-229) */
FZ_API int fz_wide_norm_weight_host(int device, const int64_t *h_rows, size_t batch, int degree, uint64_t *h_max_abs,
                                    int32_t *h_weight);

#ifdef __cplusplus
}
#endif
#endif
