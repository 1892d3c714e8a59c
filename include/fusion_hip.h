/*
 * fusion_hip.h -- C ABI of libfusion_hip.so, the MI355X (gfx950) implementation of the
 * algebra hot path of geometry-labs/fusion-cryptography.
 *
 * The reference is pure Python and has no FFI; its boundary for this path is the Python
 * surface of algebra/ntt.py, algebra/polynomials.py, algebra/matrices.py as called from
 * fusion/fusion.py.  Every entry point below names the reference code it replaces
 * (paths relative to the reference checkout).  INTEGRATION.md shows the ctypes stub a
 * reference maintainer would add to bind them.
 *
 * Conventions
 *  - Plain C: pointers, sizes, int status codes.  No C++/torch types.
 *  - All polynomial data are int32, row-major [rows][degree].  Inputs may be ANY int32
 *    (e.g. the non-centred output of the reference's __neg__); outputs are always the
 *    centred representative in [-(q-1)/2, (q-1)/2], exactly what the reference's
 *    cent() (algebra/ntt.py:93-123) returns.
 *  - Pointers named d_* are DEVICE pointers (hipMalloc / torch.Tensor.data_ptr()).
 *    Pointers named h_* are HOST pointers; the *_host entry points stage through
 *    device memory internally (PCIe-inclusive; convenience for the object API).
 *  - Every call returns FZ_OK (0) or a negative FZ_E_* code; fz_last_error() returns a
 *    thread-local message.  The library never frees or retains caller buffers.
 *  - Kernels are enqueued on the context's stream (fz_ctx_set_stream) and the device
 *    entry points do NOT synchronise; *_host entry points return after completion.
 *  - One context is used by one host thread at a time (one context + stream per GPU).
 *
 * Timers, per-dispatch profiling, launch-floor probes, device-side launch timestamps and runtime / library
 * reports are NOT part of this surface: they live in fusion_hip_diag.h (same library, same conventions) and
 * are what bench.py and tools/ use; a reference maintainer binds this header only.
 */
#ifndef FUSION_HIP_H
#define FUSION_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FZ_API __attribute__((visibility("default")))

/* status codes */
#define FZ_OK              0
#define FZ_E_BADARG      (-1)   /* null pointer, bad size, q/degree/root inconsistent   */
#define FZ_E_UNSUPPORTED (-2)   /* parameters outside what the kernels implement        */
#define FZ_E_HIP         (-3)   /* a HIP runtime call failed (message in fz_last_error) */
#define FZ_E_NODEVICE    (-4)   /* no usable gfx950 device                              */
#define FZ_E_RCCL        (-5)   /* RCCL missing or an RCCL call failed (fz_comm_*, fz_allreduce_i64) */

/* verify verdict codes (fusion/fusion.py:686-728) */
#define FZ_VERDICT_OK              0  /* (True, "")                                              */
#define FZ_VERDICT_TOO_MANY_KEYS   1  /* "Too many keys."                              :687      */
#define FZ_VERDICT_LEN_MISMATCH    2  /* "Number of keys and messages must be equal."  :689      */
#define FZ_VERDICT_TARGET_MISMATCH 3  /* "Target doesn't match image of aggregate signature." :721 */
#define FZ_VERDICT_NORM            4  /* "Norm of aggregate signature too large."      :725      */
#define FZ_VERDICT_WEIGHT          5  /* "Weight of aggregate signature too large."    :727      */

typedef struct fz_ctx fz_ctx;

/* ---- library / device ----------------------------------------------------------------- */
FZ_API const char *fz_version(void);
FZ_API const char *fz_last_error(void);
FZ_API int fz_device_count(int *out_count);

/* ---- context: one per (device, modulus, degree, root) ------------------------------------
 * Replaces the per-call parameter plumbing of the reference (modulus, degree, root,
 * inv_root, root_order carried by every PolynomialRepresentation, algebra/polynomials.py:16-50)
 * and the per-call twiddle rebuild `bit_reverse_copy([pow(root, i, q) ...])`
 * (algebra/polynomials.py:396-397, :414-417; algebra/ntt.py:443-449).
 * Requirements: q odd, 3 <= q < 2^32 (centred residues are int32 for every such q; primality is the caller's check, as in
 * the reference), degree a power of two in [2, 4096] (64 and 256, the scheme's, take the tuned kernels; up to 256 the register /
 * LDS schedules; 512 .. 4096 one workgroup per polynomial), root a primitive 2*degree-th root of unity mod q,
 * root*inv_root == 1 mod q (the same conditions PolynomialRepresentation.__init__ checks, polynomials.py:36-45).
 * root == 0: a ring-only context (pointwise operations, no transforms). */
FZ_API int fz_ctx_create(int device_id, uint32_t q, int degree, uint32_t root, uint32_t inv_root,
                         fz_ctx **out);
/* The same with the twiddle tables GIVEN: cooley_tukey_ntt / gentleman_sande_intt take their table as an argument and use
 * whatever they are handed (algebra/ntt.py:216-291 `s = bit_rev_root_powers[m + i]`, :294-377) -- a list that is not the
 * bit-reversed power table of one root is transformed with it all the same.  h_fwd / h_inv: `degree` entries each (reduced mod
 * q), used by fz_ntt_forward / fz_ntt_inverse and everything built on them; both required (pass one table twice when only one
 * direction will be used). */
FZ_API int fz_ctx_create_tables(int device_id, uint32_t q, int degree, const uint32_t *h_fwd, const uint32_t *h_inv, fz_ctx **out);
FZ_API int fz_ctx_destroy(fz_ctx *ctx);
FZ_API int fz_ctx_set_stream(fz_ctx *ctx, void *hip_stream);   /* NULL = default stream */
FZ_API int fz_ctx_synchronize(fz_ctx *ctx);
/* a HIP stream on the context's device for host languages without a HIP binding (pass it to fz_ctx_set_stream;
 * detach it from every context before destroying it) */
FZ_API int fz_stream_create(fz_ctx *ctx, void **out_stream);
FZ_API int fz_stream_destroy(fz_ctx *ctx, void *hip_stream);
/* the same with a scheduling priority: high != 0 asks for the device's highest stream priority -- the workgroups of its kernels
 * are dispatched ahead of those of normal streams, which is what a small latency-bound launch (the exchange step's all-reduce,
 * a verification of a few aggregates) needs beside a launch that fills the chip; 0 asks for the lowest */
FZ_API int fz_stream_create_priority(fz_ctx *ctx, int high, void **out_stream);
/* copies the bit-reversed twiddle tables the context uses (each `degree` uint32 in [0,q));
 * equal to bit_reverse_copy([pow(root,i,q)]) / ([pow(inv_root,i,q)]). Either may be NULL. */
FZ_API int fz_ctx_twiddles(fz_ctx *ctx, uint32_t *h_fwd, uint32_t *h_inv);

/* ---- graph capture --------------------------------------------------------------------------
 * The reference runs its algebra as a long sequence of small calls (one cooley_tukey_ntt /
 * gentleman_sande_intt per polynomial: fusion/fusion.py:363-370, :557, :670-676); at the batch sizes
 * of BASELINE configs[1] each device call is only a few microseconds and the per-launch cost is a
 * fifth of it.  A sequence of device-pointer calls (transforms, pointwise, matvec, *_core, *_partial,
 * *_async; nothing that copies to the host, synchronises, or grows the scratch -- run the sequence
 * once un-captured first) issued between fz_graph_begin and fz_graph_end is recorded into a hipGraph
 * instead of executed; fz_graph_launch replays it on the context's stream with one call
 * (measured: 3.85 us per 4096 x 256 transform instead of 4.2-4.4 us).  The context needs a non-default
 * stream (fz_ctx_set_stream).  The recorded pointers and sizes are fixed; contents may change. */
typedef struct fz_graph fz_graph;
FZ_API int fz_graph_begin(fz_ctx *ctx);
FZ_API int fz_graph_end(fz_ctx *ctx, fz_graph **out_graph);
FZ_API int fz_graph_launch(fz_ctx *ctx, fz_graph *graph);      /* asynchronous on the context's stream */
FZ_API int fz_graph_destroy(fz_graph *graph);

/* ---- events: ordering between the streams of two contexts ------------------------------------
 * The reference is one Python thread with no streams; a host that keeps several contexts busy (one stream each) needs an
 * order between them in exactly one place of this path: the exchange step of a sharded aggregation or verification
 * (the sums of fusion/fusion.py:670-676 and :706-714 across GPUs, fz_allreduce_i64) can run on a second context's stream while
 * the first context signs the next batch (fusion.py:557) -- record after the partial sums, let the exchanging context wait,
 * record after the exchange, let the verifying context wait.  fz_event_record marks the point the context's stream has
 * reached; fz_event_wait makes the context's stream wait for the marked point without blocking the host.  Both can be issued
 * inside a capture (fz_graph_begin): waiting for an event recorded in another context's capture draws the waiting context's
 * stream into that capture (fork), and the capturing context must wait for an event recorded on it again before
 * fz_graph_end (join); while its stream is part of a capture a context obeys the rules of a capturing one. */
typedef struct fz_event fz_event;
FZ_API int fz_event_create(fz_ctx *ctx, fz_event **out);
FZ_API int fz_event_record(fz_ctx *ctx, fz_event *ev);       /* on the context's stream */
FZ_API int fz_event_wait(fz_ctx *ctx, fz_event *ev);         /* the context's stream waits; asynchronous */
FZ_API int fz_event_destroy(fz_event *ev);

/* ---- device memory helpers (so a host language needs no HIP binding of its own) --------
 * fz_free keeps blocks of 256 KiB or more for reuse by later fz_malloc calls of this context; ONE budget of FZ_POOL_MB
 * megabytes (default 4096; 0 = every fz_free is a hipFree, which for a large block takes ~180 us and synchronises the
 * device) covers the idle blocks of ALL contexts of the process.
 * STREAM-ORDER REQUIREMENT OF fz_free: the last work that uses the block must already be queued on the context's stream
 * (or have completed) when fz_free is called -- exactly what holds for memory handed to this context's entry points.
 * fz_free records an event on that stream and whichever stream takes the block out of the pool waits for it, so a reused
 * block's previous users finish before its next ones start even across fz_ctx_set_stream (which also drains the old stream
 * on every change).  A block still in use by ANOTHER stream or context (torch, a second fz_ctx) must be synchronised by the
 * caller before fz_free: unlike hipFree, a pooled free does not synchronise the device.  fz_malloc / fz_free may be called
 * from any thread (the pool is locked); fz_ctx_destroy releases the pool, fz_pool_trim releases it down to keep_bytes,
 * and a failed allocation flushes the idle blocks of every context on the device before it is retried. */
FZ_API int fz_malloc(fz_ctx *ctx, size_t bytes, void **d_out);
FZ_API int fz_free(fz_ctx *ctx, void *d_ptr);
FZ_API int fz_pool_trim(fz_ctx *ctx, size_t keep_bytes);
FZ_API int fz_memcpy_h2d(fz_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);   /* async on ctx stream */
FZ_API int fz_memcpy_d2h(fz_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);   /* returns after completion */

/* ---- transforms ---------------------------------------------------------------------------
 * fz_ntt_forward: cooley_tukey_ntt (algebra/ntt.py:216-291) on `batch` rows.
 *   natural order in, bit-reversed order out, out[i] = cent(sum_j x[j] psi^((2 brv(i)+1) j)).
 * fz_ntt_inverse: gentleman_sande_intt (algebra/ntt.py:294-377) incl. the n^{-1} scaling.
 * d_in == d_out (in place) is allowed.  These are also what transform()
 * (algebra/polynomials.py:391-433) runs for each polynomial. */
FZ_API int fz_ntt_forward(fz_ctx *ctx, const int32_t *d_in, int32_t *d_out, size_t batch);
FZ_API int fz_ntt_inverse(fz_ctx *ctx, const int32_t *d_in, int32_t *d_out, size_t batch);
FZ_API int fz_ntt_forward_host(fz_ctx *ctx, int32_t *h_data, size_t batch);   /* in place */
FZ_API int fz_ntt_inverse_host(fz_ctx *ctx, int32_t *h_data, size_t batch);   /* in place */

/* Many independent transforms in ONE dispatch.  The reference issues its transforms polynomial by polynomial
 * (transform(), algebra/polynomials.py:391-433, called 2*rank times per key in fusion/fusion.py:363-368, rank times
 * per verification in :690-692, once per challenge in :499-507); a device launch of a few thousand rows is mostly
 * dispatch floor.  A job is (d_in, d_out, rows, direction); jobs must be independent of each other (d_in == d_out
 * inside a job is allowed).  The job table travels in the kernel arguments: nothing is copied to the device, the
 * call is asynchronous and can be recorded by fz_graph_*.  Degree 64 / 256: one launch per 32 jobs; other degrees:
 * one launch per job.  h_jobs is read before the call returns. */
typedef struct fz_ntt_job {
    const int32_t *d_in;
    int32_t *d_out;
    size_t rows;
    int inverse;               /* 0: cooley_tukey_ntt (ntt.py:216-291), 1: gentleman_sande_intt (ntt.py:294-377) */
} fz_ntt_job;
FZ_API int fz_ntt_multi(fz_ctx *ctx, const fz_ntt_job *h_jobs, size_t n_jobs);

/* ---- pointwise ring operations on `count` int32 values -------------------------------------
 * PolynomialNTTRepresentation.__mul__/__add__/__neg__/__sub__ (algebra/polynomials.py:341-385,
 * :282-318, :325-336); same formulas serve PolynomialCoefficientRepresentation.__add__/__neg__
 * (:114-163).  neg returns -(x mod q) in [-(q-1), 0] like the reference (NOT centred);
 * sub = cent(a + neg(b)).  mulacc: acc = cent(acc + cent(a*b)), the inner step of
 * GeneralMatrix.__mul__ (algebra/matrices.py:127-129). */
FZ_API int fz_pw_mul(fz_ctx *ctx, const int32_t *d_a, const int32_t *d_b, int32_t *d_out, size_t count);
FZ_API int fz_pw_add(fz_ctx *ctx, const int32_t *d_a, const int32_t *d_b, int32_t *d_out, size_t count);
FZ_API int fz_pw_sub(fz_ctx *ctx, const int32_t *d_a, const int32_t *d_b, int32_t *d_out, size_t count);
FZ_API int fz_pw_neg(fz_ctx *ctx, const int32_t *d_a, int32_t *d_out, size_t count);
FZ_API int fz_pw_mulacc(fz_ctx *ctx, int32_t *d_acc, const int32_t *d_a, const int32_t *d_b, size_t count);
/* host-pointer forms used by the object API (op: 0 mul, 1 add, 2 sub, 3 neg (h_b ignored)) */
FZ_API int fz_pw_binary_host(fz_ctx *ctx, int op, const int32_t *h_a, const int32_t *h_b,
                             int32_t *h_out, size_t count);
/* out[row] = cent(a[row] * s) for one NTT-domain polynomial s broadcast over rows:
 * GeneralMatrix.__mul__(element), algebra/matrices.py:109-114 */
FZ_API int fz_pw_mul_bcast(fz_ctx *ctx, const int32_t *d_a, const int32_t *d_s, int32_t *d_out, size_t rows);

/* (The generic-parameter path -- any odd modulus below 2^63, any power-of-two length: fz_wide_* -- is declared in
 * fusion_hip_generic.h: a correctness path for the parameters the reference accepts beyond the scheme's, not for throughput.) */

/* ---- synthetic batches (benchmarks, tests) ----------------------------------------------------------
 * d_out[i] = SplitMix64(seed + (i + 1) * 0x9E3779B97F4A7C15) mod q, centred to [-(q-1)/2, (q-1)/2]: a seeded uniform
 * batch generated where it is used, instead of on the host and across PCIe (a 2^20 x 256 batch is 1 GiB). */
FZ_API int fz_fill_synthetic(fz_ctx *ctx, int32_t *d_out, size_t count, uint64_t seed);

/* ---- negacyclic product of coefficient-domain polynomials ---------------------------------------
 * out[b] = INTT(NTT(f[b]) (.) NTT(g[b])), centred: ntt_poly_mult (algebra/ntt.py:380-484) and the value of
 * PolynomialCoefficientRepresentation.__mul__ (algebra/polynomials.py:171-216, schoolbook there).
 * One launch for degrees 32 .. 256 (both forward transforms, the product and the inverse stay on chip:
 * 12*degree bytes of HBM traffic per product; degrees 32 and 128 need 16-byte aligned buffers for it); other
 * degrees compose the transform and pointwise kernels (16-byte aligned buffers).
 * d_out may alias d_f or d_g.  Rows are [batch][degree], 4-byte aligned, any int32 in, centred out. */
FZ_API int fz_poly_mul(fz_ctx *ctx, const int32_t *d_f, const int32_t *d_g, int32_t *d_out, size_t batch);
FZ_API int fz_poly_mul_host(fz_ctx *ctx, const int32_t *h_f, const int32_t *h_g, int32_t *h_out, size_t batch);

/* ---- (1 x l) . (l x 1) polynomial matrix-vector product ------------------------------------
 * GeneralMatrix.__mul__(GeneralMatrix), algebra/matrices.py:115-131, for the only shape the
 * scheme uses (fusion/fusion.py:369-370, :715-717): out[b] = cent(sum_k A[k] (.) S[b][k]).
 * A: [l][degree] shared; S: [batch][l][degree]; out: [batch][degree]. */
FZ_API int fz_matvec(fz_ctx *ctx, const int32_t *d_A, const int32_t *d_S, int32_t *d_out,
                     size_t batch, int l);
FZ_API int fz_matvec_host(fz_ctx *ctx, const int32_t *h_A, const int32_t *h_S, int32_t *h_out,
                          size_t batch, int l);

/* ---- fused scheme cores (the arithmetic inside fusion/fusion.py) ---------------------------
 * keygen (fusion.py:363-370): coef [batch][2][l][degree] (left rows then right rows, coefficient
 *   domain) -> sk_hat same shape (NTT of every row) and vk [batch][2][degree] = A . sk_hat.
 * sign (fusion.py:557): sig[b][k] = cent(cent(L[b][k] (.) c_hat[b]) + R[b][k]).
 *   sk_hat: [batch][2][l][degree] as produced by keygen; c_hat: [batch][degree]; sig: [batch][l][degree].
 * aggregate (fusion.py:670-676): out[k] = cent(sum_i sig[i][k] (.) alpha_hat[i]); sig [N][l][degree],
 *   alpha_hat [N][degree], out [l][degree].
 * aggregate_partial: the same sum WITHOUT the final reduction, as int64 per coefficient
 *   (|value| < N * q/2), for cross-GPU summation (ncclSum on int64) followed by
 *   fz_reduce_i64 -- the one exchange step of the path.
 * verify (fusion.py:690-727): returns the verdict code in *h_verdict (FZ_VERDICT_*).
 *   A [l][degree], sig [l][degree] aggregate, vkL/vkR/c_hat/alpha_hat [N][degree]. */
FZ_API int fz_keygen_core(fz_ctx *ctx, const int32_t *d_A, const int32_t *d_coef,
                          int32_t *d_sk_hat, int32_t *d_vk, size_t batch, int l);
/* the same with ONE secret polynomial per (key, half), d_coef [batch][2][degree], used for all l rows: what the
 * reference's keygen really computes, because its sampler is re-seeded with the same seed for every matrix entry
 * (fusion/fusion.py:159-170, :189-199 -> algebra/polynomials.py:447-448) -- 83x less data to upload */
FZ_API int fz_keygen_core_bcast(fz_ctx *ctx, const int32_t *d_A, const int32_t *d_coef,
                                int32_t *d_sk_hat, int32_t *d_vk, size_t batch, int l);
FZ_API int fz_sign_core(fz_ctx *ctx, const int32_t *d_sk_hat, const int32_t *d_c_hat,
                        int32_t *d_sig, size_t batch, int l);
FZ_API int fz_aggregate_core(fz_ctx *ctx, const int32_t *d_sig, const int32_t *d_alpha_hat,
                             int32_t *d_out, size_t N, int l);
FZ_API int fz_aggregate_partial(fz_ctx *ctx, const int32_t *d_sig, const int32_t *d_alpha_hat,
                                int64_t *d_partial, size_t N, int l);
/* `groups` independent aggregates in one launch: sig [groups][N][l][degree], alpha_hat [groups][N][degree],
 * partial of group g at d_partial + g * partial_stride (int64 elements). */
FZ_API int fz_aggregate_partial_batch(fz_ctx *ctx, const int32_t *d_sig, const int32_t *d_alpha_hat,
                                      int64_t *d_partial, size_t partial_stride, size_t groups, size_t N, int l);
/* aggregate partials AND the verification target's partials (fusion.py:706-714) in one pass over the signers
 * (two launches instead of five): vkL, vkR, c_hat [groups][N][degree]; target partial of group g at
 * d_target_partial + g * target_stride.  All inputs 16-byte aligned. */
FZ_API int fz_aggregate_target_partial_batch(fz_ctx *ctx, const int32_t *d_sig, const int32_t *d_alpha_hat,
                                             const int32_t *d_vkL, const int32_t *d_vkR, const int32_t *d_c_hat,
                                             int64_t *d_partial, size_t partial_stride,
                                             int64_t *d_target_partial, size_t target_stride,
                                             size_t groups, size_t N, int l);
/* sign AND aggregate in one pass (a signing service that also aggregates: the reference's sign(), fusion.py:534-557, called once
 * per key, then aggregate(), :655-677, and verify()'s target, :706-714): sigma_i = L_i * c_i + R_i is written to d_sig
 * [groups][N][l][degree] as it is computed and enters the aggregate's int64 partial sums from registers, so the l rows of a
 * signature are never read back -- (3l + 2) rows of traffic per signature instead of (3l + 1) + (l + 1) for fz_sign_core
 * followed by fz_aggregate_target_partial_batch, whose results it reproduces exactly.  d_sk_hat [groups][N][2][l][degree],
 * d_c_hat / d_alpha_hat [groups][N][degree]; d_vkL, d_vkR and d_target_partial are given together (the target's partial sums
 * in the same launch) or all NULL.  Power-of-two degrees <= 256 with 16-byte aligned rows take the one-launch form; anything
 * else runs the two launches. */
FZ_API int fz_sign_aggregate_target_partial_batch(fz_ctx *ctx, const int32_t *d_sk_hat, const int32_t *d_c_hat,
                                                  const int32_t *d_alpha_hat, const int32_t *d_vkL, const int32_t *d_vkR,
                                                  int32_t *d_sig, int64_t *d_partial, size_t partial_stride,
                                                  int64_t *d_target_partial, size_t target_stride, size_t groups, size_t N, int l);
/* MANY aggregates of DIFFERENT sizes in one launch -- the reference is called once per aggregate (fusion.py:655, :680;
 * benchmarks/benchmarks.py:37-141 loops over them): aggregate g's signers are rows [h_offsets[g], h_offsets[g+1]) of the
 * concatenated arrays d_sig [sum N][l][degree], d_alpha_hat / d_vkL / d_vkR / d_c_hat [sum N][degree]; h_offsets (HOST,
 * groups + 1 entries, read before the call returns) travels in the kernel arguments, so the call is asynchronous and
 * capturable.  fz_aggregate_core_ragged writes the centred aggregates d_out [groups][l][degree];
 * fz_aggregate_target_partial_ragged the int64 partial sums of aggregates and verification targets (as
 * fz_aggregate_target_partial_batch does for equal sizes); with d_sig == NULL and d_partial == NULL only the targets
 * (fusion.py:706-714) are computed.  Power-of-two degree <= 256, 16-byte aligned rows. */
FZ_API int fz_aggregate_core_ragged(fz_ctx *ctx, const int32_t *d_sig, const int32_t *d_alpha_hat, const size_t *h_offsets,
                                    size_t groups, int l, int32_t *d_out);
FZ_API int fz_aggregate_target_partial_ragged(fz_ctx *ctx, const int32_t *d_sig, const int32_t *d_alpha_hat,
                                              const int32_t *d_vkL, const int32_t *d_vkR, const int32_t *d_c_hat,
                                              const size_t *h_offsets, size_t groups, int l,
                                              int64_t *d_partial, size_t partial_stride,
                                              int64_t *d_target_partial, size_t target_stride);
/* target partial for verify: sum_i (vkL_i (.) c_i + vkR_i) (.) alpha_i as int64 [degree] */
FZ_API int fz_target_partial(fz_ctx *ctx, const int32_t *d_vkL, const int32_t *d_vkR,
                             const int32_t *d_c_hat, const int32_t *d_alpha_hat,
                             int64_t *d_partial, size_t N);
FZ_API int fz_target_partial_batch(fz_ctx *ctx, const int32_t *d_vkL, const int32_t *d_vkR,
                                   const int32_t *d_c_hat, const int32_t *d_alpha_hat,
                                   int64_t *d_partial, size_t partial_stride, size_t groups, size_t N);
/* out[i] = cent(in[i]) for int64 sums */
FZ_API int fz_reduce_i64(fz_ctx *ctx, const int64_t *d_in, int32_t *d_out, size_t count);
FZ_API int fz_verify_core(fz_ctx *ctx, const int32_t *d_A, const int32_t *d_sig,
                          const int32_t *d_vkL, const int32_t *d_vkR,
                          const int32_t *d_c_hat, const int32_t *d_alpha_hat,
                          size_t N, int l, int64_t beta_vf, int64_t omega_vf, int *h_verdict);
/* verify with a precomputed target [degree] (multi-GPU: target summed across ranks first) */
FZ_API int fz_verify_with_target(fz_ctx *ctx, const int32_t *d_A, const int32_t *d_sig,
                                 const int32_t *d_target, int l,
                                 int64_t beta_vf, int64_t omega_vf, int *h_verdict);
/* `groups` aggregates against one public challenge: sig [groups][l][degree], target [groups][degree],
 * h_verdicts [groups]. */
FZ_API int fz_verify_with_target_batch(fz_ctx *ctx, const int32_t *d_A, const int32_t *d_sig,
                                       const int32_t *d_target, size_t groups, int l,
                                       int64_t beta_vf, int64_t omega_vf, int *h_verdicts);

/* verification straight from int64 partial sums (what the all-reduce leaves): aggregate g at d_partial +
 * g * partial_stride, its target at d_target_partial + g * target_stride; values are centred on load, so no
 * fz_reduce_i64 pass and no int32 copy of the aggregate is needed.  Asynchronous; degree 64 or 256. */
FZ_API int fz_verify_partials_batch_async(fz_ctx *ctx, const int32_t *d_A, const int64_t *d_partial, size_t partial_stride,
                                          const int64_t *d_target_partial, size_t target_stride, size_t groups, int l,
                                          int64_t beta_vf, int64_t omega_vf, int *d_verdicts);

/* the same without the device-to-host copy: verdict codes are written to d_verdicts [groups] on the context's
 * stream and nothing is synchronised (pipelined verification; degree 64 or 256) */
FZ_API int fz_verify_with_target_batch_async(fz_ctx *ctx, const int32_t *d_A, const int32_t *d_sig,
                                             const int32_t *d_target, size_t groups, int l,
                                             int64_t beta_vf, int64_t omega_vf, int *d_verdicts);

/* ---- the exchange step across GPUs: RCCL all-reduce of the int64 partial sums -------------------
 * aggregate() (fusion.py:670-676) and verify()'s target (:706-714) are sums over signers; with the signers sharded over
 * GPUs each rank holds exact int64 partials (fz_aggregate_partial*, fz_target_partial*, fz_aggregate_target_partial_batch)
 * and ONE ncclAllReduce(ncclInt64, ncclSum) over xGMI completes them (8 centred values already overflow int32).
 * One process per GPU: rank 0 calls fz_comm_unique_id and hands the 128 bytes to the other ranks by any means
 * (a file, a socket, MPI, torch.distributed); every rank calls fz_comm_create (collective: returns when all ranks
 * have joined).  fz_allreduce_i64 runs in place on the context's stream, asynchronously, ordered with the kernels
 * issued around it, and can be recorded by fz_graph_* together with them.  RCCL is bound at the first fz_comm_* call
 * (dlopen "librccl.so.1"); without it these entries return FZ_E_RCCL and everything else keeps working. */
typedef struct fz_unique_id { char internal[128]; } fz_unique_id;      /* == ncclUniqueId */
typedef struct fz_comm fz_comm;
FZ_API int fz_comm_unique_id(fz_unique_id *out_id);
FZ_API int fz_comm_create(fz_ctx *ctx, int nranks, int rank, const fz_unique_id *id, fz_comm **out);
FZ_API int fz_comm_destroy(fz_comm *comm);
/* what the communicator itself reports (ncclCommCount) and this rank's index */
FZ_API int fz_comm_info(fz_comm *comm, int *out_nranks, int *out_rank);
/* the version code of the RCCL this process bound (ncclGetVersion, e.g. 22703): bench.py prints it per rank so that a
 * multi-GPU record shows which library carried the exchange step */
FZ_API int fz_rccl_version(int *out_version);
FZ_API int fz_allreduce_i64(fz_ctx *ctx, fz_comm *comm, int64_t *d_buf, size_t count);
/* the same sums when every rank needs only ITS block of them (a rank verifies only the aggregates it owns, fusion.py:690-727
 * from the sums of :670-676 and :706-714): ncclReduceScatter(ncclInt64, ncclSum) in place -- d_buf holds nranks blocks of
 * count_per_rank elements, block r of the summed buffer arrives in block r of rank r's d_buf, the other blocks of d_buf are
 * left undefined.  Half the traffic and half the steps of the all-reduce.  Asynchronous, capturable, ordered like it. */
FZ_API int fz_reduce_scatter_i64(fz_ctx *ctx, fz_comm *comm, int64_t *d_buf, size_t count_per_rank);
/* rank `root`'s d_buf [count] int32 to every rank's d_buf (ncclBroadcast, in place, on the context's stream): how the rank
 * that ran hash_ag's serial sponge (fusion.py:632-652) hands the aggregation-coefficient rows to the others
 * (fusion_hip.dist.ShardedScheme, alpha_mode "root") */
FZ_API int fz_broadcast_i32(fz_ctx *ctx, fz_comm *comm, int32_t *d_buf, size_t count, int root);

/* ---- norm / weight of coefficient rows -------------------------------------------------------
 * PolynomialCoefficientRepresentation.norm("infty") / weight(), algebra/polynomials.py:221-227:
 * max |x| over the STORED values and #{x : x mod q != 0}, per row. */
FZ_API int fz_norm_weight(fz_ctx *ctx, const int32_t *d_coef, size_t batch,
                          int64_t *d_max_abs, int32_t *d_weight);
FZ_API int fz_norm_weight_host(fz_ctx *ctx, const int32_t *h_coef, size_t batch,
                               int64_t *h_max_abs, int32_t *h_weight);

/* ---- challenge pipeline on the host -----------------------------------------
 * The scheme's hash -> challenge path: exact-format serialisation of the key objects, SHA3-256 /
 * SHAKE-256 (FIPS 202) and the byte decoder.  Host memory only (h_*), no GPU involved; `threads` > 1
 * spreads independent signers over host threads.  The forward NTT of the decoded coefficient rows is
 * fz_ntt_forward.
 *   fz_scheme_params mirrors the fields of fusion.fusion.Params that the path reads (fusion.py:204-282). */
typedef struct fz_scheme_params {
    int64_t modulus, root, inv_root;
    int32_t degree, root_order, secpar;
    int32_t omega_ch, omega_ag;
    int64_t beta_ch, beta_ag;
    int32_t bytes_for_one_coef_bdd_by_beta_ch, bytes_for_poly_shuffle;
    uint8_t sign_pre_hash_dst[2], sign_hash_dst[2], agg_xof_dst[2];
} fz_scheme_params;

FZ_API int fz_sha3_256(const uint8_t *h_data, size_t len, uint8_t *h_out32);
FZ_API int fz_shake256(const uint8_t *h_data, size_t len, uint8_t *h_out, size_t out_len);
/* str(OneTimeVerificationKey) (fusion.py:328-329 -> matrices.py:40-41 -> polynomials.py:257-258) of a key
 * given as its two degree-long rows; h_out may be NULL to query the length. */
FZ_API int fz_format_vk(const fz_scheme_params *P, const int32_t *h_vk_left, const int32_t *h_vk_right,
                        char *h_out, size_t cap, size_t *out_len);
/* decode_bytes_to_polynomial_coefficients (fusion.py:422-481) */
FZ_API int fz_decode_coefficients(const uint8_t *h_bytes, size_t len, int log2_bias, int64_t modulus, int degree,
                                  int64_t norm_bound, int weight_bound, int32_t *h_out);
/* hash_message_to_int (fusion.py:405-409) for N messages (UTF-8, concatenated; h_msg_off has N+1 entries):
 * the 32-byte digests, i.e. the integers in little-endian byte order. */
FZ_API int fz_hash_messages(const fz_scheme_params *P, const char *h_msgs, const size_t *h_msg_off, size_t N,
                            uint8_t *h_prehash);
/* hash_ch up to (not including) the NTT (fusion.py:511-531 = :405-409, :412-419, :484-506): coefficient rows
 * [N][degree] of the challenges of N (key, message) pairs; h_prehash [N][32] may be NULL. */
FZ_API int fz_challenge_coefficients(const fz_scheme_params *P, const int32_t *h_vk_left, const int32_t *h_vk_right,
                                     const char *h_msgs, const size_t *h_msg_off, size_t N, int32_t *h_coefs,
                                     uint8_t *h_prehash, int threads);
/* the order sorted(..., key=str(vk)) puts the keys in (fusion.py:661-663, :693): h_order[i] = index of the
 * i-th key in sorted order (stable). */
FZ_API int fz_sort_by_vk_string(const fz_scheme_params *P, const int32_t *h_vk_left, const int32_t *h_vk_right,
                                size_t N, size_t *h_order, int threads);
/* hash_ag up to the NTTs (fusion.py:573-629): keys, pre-hashed messages and NTT-domain challenges c_hat
 * [N][degree], all already in sorted key order -> coefficient rows [N][degree] of the aggregation coefficients. */
FZ_API int fz_aggregation_coefficients(const fz_scheme_params *P, const int32_t *h_vk_left,
                                       const int32_t *h_vk_right, const uint8_t *h_prehash,
                                       const int32_t *h_c_hat, size_t N, int32_t *h_coefs, int threads);

/* ---- the per-signer challenge pipeline on the DEVICE (device half) ------------------------------
 * hash_ch (fusion.py:511-531) for N independent (key, message) pairs without the keys ever leaving the device as text:
 * the exact str(OneTimeVerificationKey) serialisation (fusion.py:328-329 -> matrices.py:40-41 -> polynomials.py:257-258),
 * SHAKE-256 (fusion.py:412-419), the byte decoder (fusion.py:422-481) and -- fz_challenge_hat_dev -- the forward
 * transform (fusion.py:499-507).  d_vk [N][2][degree] as fz_keygen_core writes it (left row, right row); h_prehash
 * [N][32] from fz_hash_messages (HOST memory: 32 bytes per message, converted to decimal text and uploaded by the call);
 * d_coefs / d_c_hat [N][degree].  Asynchronous on the context's stream; the caller's HOST arrays have been consumed when
 * the call returns (up to 4608 signers per call -- a wave per signer, one fused kernel -- they are copied into pinned staging
 * that the kernel reads in place: nothing is uploaded and nothing waited for; larger batches upload them and synchronise
 * once).  Only as many XOF bytes are squeezed as the decoder consumes (a prefix of the reference's n).  Supported: ternary
 * challenges (norm bound 1: both parameter sets of the reference), degree 4..256; otherwise FZ_E_UNSUPPORTED -- use
 * fz_challenge_coefficients.
 * hash_ag (fusion.py:632-652) stays on the host: it is ONE serial XOF over all signers by construction. */
FZ_API int fz_challenge_coefficients_dev(fz_ctx *ctx, const fz_scheme_params *P, const int32_t *d_vk, const uint8_t *h_prehash,
                                         size_t N, int32_t *d_coefs);
FZ_API int fz_challenge_hat_dev(fz_ctx *ctx, const fz_scheme_params *P, const int32_t *d_vk, const uint8_t *h_prehash,
                                size_t N, int32_t *d_c_hat);
/* The same with hash_message_to_int (fusion.py:405-409: SHA3-256 of dst + "," + message) on the device as well: h_msgs the
 * N messages' bytes back to back (UTF-8, as the reference's .encode()), h_msg_off [N + 1] their offsets (as
 * fz_hash_messages takes them); h_prehash_out (optional) receives the [N][32] digests, which hash_ag needs on the host
 * (the call then synchronises the stream before it returns: the digests are the kernel's). */
FZ_API int fz_challenge_hat_msgs_dev(fz_ctx *ctx, const fz_scheme_params *P, const int32_t *d_vk, const char *h_msgs,
                                     const size_t *h_msg_off, size_t N, int32_t *d_c_hat, uint8_t *h_prehash_out);

/* ---- reference-exact sampling on the host -----------------------------------------
 * CPython's MT19937 `random` exactly as the reference's samplers drive it (random.seed(int), randrange):
 * the same seed yields the same polynomial as algebra/polynomials.py:436-488.  Seeds are the non-negative integers below
 * 2^64 (fz_sample_secret_polys: below 2^64 - 1, because the right half uses seed + 1); callers map other Python seeds
 * themselves (random.seed(int) uses abs(seed); larger ones need more key words than these entries take). */
FZ_API int fz_sample_ntt_values(uint64_t seed, int64_t modulus, int degree, int32_t *h_out);
FZ_API int fz_sample_coefficients(uint64_t seed, int64_t modulus, int degree, int64_t norm_bound,
                                  int64_t weight_bound, int32_t *h_out);
/* the same, and the generator's state afterwards in h_state[625] (624 words + position: what random.getstate()[1] holds), for a
 * caller that replaces the Python function and must leave the process-global `random` where that function leaves it */
FZ_API int fz_sample_coefficients_state(uint64_t seed, int64_t modulus, int degree, int64_t norm_bound,
                                        int64_t weight_bound, int32_t *h_out, uint32_t *h_state);
/* the two distinct secret polynomials of keygen(params, seed) for N keys: [N][2][degree] */
FZ_API int fz_sample_secret_polys(const uint64_t *h_seeds, size_t N, int64_t modulus, int degree,
                                  int64_t norm_bound, int64_t weight_bound, int32_t *h_out, int threads);
/* The same on the DEVICE (csrc/fz_sample.hip): one lane per polynomial runs CPython's MT19937 exactly (init_by_array
 * seeding, getrandbits, rejection) and writes d_out [N][2][degree] in device memory -- what fz_keygen_core_bcast reads, so
 * the secret polynomials of keygen(params, seed) never exist on the host.  Supported: weight_bound >= degree (both parameter
 * sets: no shuffle), seeds < 2^64 - 1 (the right half is seeded with seed + 1, which must not wrap); else
 * FZ_E_UNSUPPORTED.  Synchronous (it reads back a completion flag). */
FZ_API int fz_sample_secret_polys_dev(fz_ctx *ctx, const uint64_t *h_seeds, size_t N, int64_t modulus, int degree,
                                      int64_t norm_bound, int64_t weight_bound, int32_t *d_out);

/* ---- asynchronous batch queue (round 4) -------------------------------------------------------------------------
 * The reference is called once per key / signature (fusion.py:338-373, :534-557).  A call of BASELINE's size (1024 keys +
 * 1024 signatures) is a latency chain that leaves most of the chip idle; the queue takes such calls from ONE host thread
 * without blocking it and runs whatever is pending as ONE batch on worker threads that each own a context and a stream
 * (csrc/fz_queue.hip): every row of the path is independent, so the results are those of separate calls, bit for bit.
 *   fz_queue_create   workers (1..16) contexts + streams on `device`; h_A [rank][degree] the public challenge; max_rows:
 *                     keys per coalesced batch (and per call).
 *   fz_queue_submit_keygen_sign   keygen(params, seed_i) + sign(params, key_i, message_i) for i < n: copies the inputs,
 *                     returns at once with a ticket.  h_vk_out (optional, ideally from fz_pinned_alloc) receives the
 *                     verification keys [n][2][degree]; it must stay valid until the call has finished.
 *                     flags: FZ_QUEUE_KEEP_SK keeps the secret keys on the device too; FZ_QUEUE_DISCARD drops every device
 *                     result when the call finishes (throughput runs; its verification keys still reach h_vk_out).
 *   fz_queue_wait     blocks until the call has finished; device pointers of its rows (owned by the queue, valid until
 *                     released; read them on any stream: the producing work has completed).
 *   fz_queue_release  gives the call's device rows back (idempotent).  The rows are recycled at once: every read of them
 *                     must have COMPLETED (the reading stream synchronised), not merely been queued.
 *   fz_queue_release_after   the same when reads of the rows are still QUEUED on `consumer`'s stream (an asynchronous
 *                     fz_aggregate_* / fz_verify_* on d_sig, say): an event recorded on that stream now is what the owning
 *                     worker waits for before the rows are reused, so the caller need not synchronise.  Several consumers:
 *                     synchronise all but the last, or call fz_queue_release after synchronising.
 *   fz_queue_drain waits for everything submitted and reports the first failure among discarded calls.  fz_queue_destroy
 *                     finishes what was submitted, releases everything (the caller has synchronised its consumers) and joins
 *                     the workers.
 * Any thread may submit / wait / release; seeds must be < 2^64 - 1 (as fz_sample_secret_polys_dev). */
typedef struct fz_queue fz_queue;
typedef struct fz_queue_result {
    int status;                  /* FZ_OK or the error code of the batch the call ran in */
    size_t n;                    /* rows of the call (0: discarded or released) */
    const int32_t *d_sk_hat;     /* [n][2][rank][degree], NULL without FZ_QUEUE_KEEP_SK */
    const int32_t *d_vk;         /* [n][2][degree] */
    const int32_t *d_sig;        /* [n][rank][degree] */
} fz_queue_result;
#define FZ_QUEUE_KEEP_SK 1
#define FZ_QUEUE_DISCARD 2
#define FZ_QUEUE_ROWS_ON_DEVICE 4   /* aggregate / verify calls: the signature (or aggregate) rows are a DEVICE pointer */
FZ_API int fz_queue_create(int device, const fz_scheme_params *P, int rank, int64_t beta_sk, int64_t omega_sk,
                           const int32_t *h_A, int workers, size_t max_rows, fz_queue **out);
FZ_API int fz_queue_destroy(fz_queue *queue);
FZ_API int fz_queue_submit_keygen_sign(fz_queue *queue, const uint64_t *h_seeds, size_t n, const char *h_msgs,
                                       const size_t *h_msg_off, int32_t *h_vk_out, int flags, uint64_t *out_ticket);
/* aggregate() + verify(), and verify() alone, as queued calls (round 5; fusion.py:655-677, :680-728 -- the reference is called
 * once per aggregate, benchmarks/benchmarks.py:37-141 loops over them).  An aggregate of a few dozen signers is a launch at the
 * dispatch floor behind milliseconds of host hashing; pending calls of one kind are run as ONE batch: hash_ch of all signers in
 * one device pipeline, hash_ag's serial SHAKE-256 of every aggregate on its own host thread, ONE ragged launch for all partial
 * sums (fz_aggregate_target_partial_ragged), ONE for all verdicts -- the values of separate calls, bit for bit.
 *   fz_queue_enable_aggregate   once, before the first such call: the verification bounds and capacity of the parameter set
 *                     (fusion.py:24-25, :63-68) and how many host threads a batch may use for its sponges.
 *   fz_queue_submit_aggregate_verify   h_vk [n][2][degree] and the messages are copied; `sig` [n][rank][degree] (host, ideally
 *                     pinned, or device with FZ_QUEUE_ROWS_ON_DEVICE) is read by the worker and must stay valid and unchanged
 *                     until the call has finished, as must the outputs: h_agg_out [rank][degree] (optional) receives
 *                     aggregate(...).signature_hat, *h_verdict_out (optional) the FZ_VERDICT_* code of verify(...) of it.
 *   fz_queue_submit_verify   the same for verify(params, keys, messages, aggregate) of a given aggregate [rank][degree].
 *   fz_queue_wait on such a ticket blocks until the outputs have been written (out->n = signers; no device rows). */
FZ_API int fz_queue_enable_aggregate(fz_queue *queue, int64_t beta_vf, int64_t omega_vf, size_t capacity, int host_threads);
FZ_API int fz_queue_submit_aggregate_verify(fz_queue *queue, const int32_t *h_vk, const char *h_msgs, const size_t *h_msg_off, size_t n,
                                            const int32_t *sig, int32_t *h_agg_out, int *h_verdict_out, int flags, uint64_t *out_ticket);
FZ_API int fz_queue_submit_verify(fz_queue *queue, const int32_t *h_vk, const char *h_msgs, const size_t *h_msg_off, size_t n,
                                  const int32_t *aggregate, int *h_verdict_out, int flags, uint64_t *out_ticket);
FZ_API int fz_queue_wait(fz_queue *queue, uint64_t ticket, fz_queue_result *out);
FZ_API int fz_queue_release(fz_queue *queue, uint64_t ticket);
FZ_API int fz_queue_release_after(fz_queue *queue, uint64_t ticket, fz_ctx *consumer);
FZ_API int fz_queue_drain(fz_queue *queue);
FZ_API int fz_queue_stats(fz_queue *queue, uint64_t *out_calls, uint64_t *out_batches, uint64_t *out_rows);
/* page-locked host memory (hipHostMalloc): device-to-host copies into it are asynchronous and run at PCIe speed */
FZ_API int fz_pinned_alloc(size_t bytes, void **h_out);
FZ_API int fz_pinned_free(void *h_ptr);

#ifdef __cplusplus
}
#endif
#endif /* FUSION_HIP_H */
