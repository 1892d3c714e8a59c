// fz_internal.h -- shared declarations between the translation units of libfusion_hip.so.
#ifndef FZ_INTERNAL_H
#define FZ_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <mutex>
#include "fz_arith.h"

// Wave-uniform twiddles of the strided pass, passed BY VALUE so they live in the kernarg
// segment and reach the kernel as scalar loads (SGPR operands of the fp64 multiplies).
struct FzTwA {
    double w[16];      // entries 1..15 of the bit-reversed power table (entry 0 unused)
    double w2[16];     // w[i] * K / q (quotient twiddles of fz_mulmod4)
    double n_inv;      // degree^{-1} mod q            (inverse only)
    double w1_n_inv;   // w[1] * degree^{-1} mod q      (inverse only: n^{-1} folded into last stage)
    double n_inv2, w1_n_inv2;   // their quotient twiddles
};

// the same for the radix-4 kernels, which use entries 1..3 only: 96 bytes of kernel arguments instead of 288 per direction
// (the multi-job launch carries both directions: a launch's scalar loads are in front of its first data load)
struct FzTw4 {
    double w[4], w2[4];
    double n_inv, w1_n_inv, n_inv2, w1_n_inv2;
};
static inline FzTw4 fz_tw4(const FzTwA &a) {
    FzTw4 t;
    for (int i = 0; i < 4; ++i) { t.w[i] = a.w[i]; t.w2[i] = a.w2[i]; }
    t.n_inv = a.n_inv; t.w1_n_inv = a.w1_n_inv; t.n_inv2 = a.n_inv2; t.w1_n_inv2 = a.w1_n_inv2;
    return t;
}

struct fz_ctx {
    int device;
    int num_cu;
    hipStream_t stream;
    hipEvent_t ev0, ev1;
    uint32_t q, root, inv_root;
    int degree, logd;
    FzMod mod;
    // host tables, bit-reversed powers in [0,q)
    uint32_t *h_tw, *h_itw;
    // device tables
    double *d_tw, *d_itw;        // [degree] as doubles (generic / small kernels)
    double *d_twB, *d_itwB;      // per-lane tables of the contiguous pass, [NE][L] pairs (w, w*K/q)
    double *d_twAB;              // {twA, itwA} as two FzTwA in device memory (polymul16 reads them as constants, per direction)
    double *d_tw2, *d_itw2;      // full tables as (w, w*K/q) pairs, [degree] (radix-4 kernels)
    int small_batch_rows;        // below this many rows the radix-4 (4 coefficients per lane) kernels run
    int force_kernel;            // 0 auto, 4 radix-4, 16 sixteen-per-lane (env FZ_NTT_KERNEL; tests and A/B runs)
    FzTwA twA, itwA;
    // growable device scratch (host-pointer entry points, int64 partial sums)
    void *d_scratch, *d_scratch2;
    size_t scratch_bytes, scratch2_bytes;
    int *d_verdict;              // [verdict_cap]
    size_t verdict_cap;
    int grid_fwd, grid_inv;      // resident-grid caps for the persistent NTT kernels
    int grid_pm;                 // resident grid of the fused product kernel (0 = not queried yet)
    int grid_pm16;               // ... of its 16-per-lane form
    int knob_ntt_rows;           // FZ_NTT_ROWS = 1 | 2 | 4: row groups per wave of the radix-4 kernels (0 = by batch size)
    // per-dispatch timing of the NTT kernels (fz_profile_begin/end): event pairs bound to the
    // dispatch itself via hipExtLaunchKernelGGL, i.e. kernel begin -> kernel end on its own stream
    int prof_on, prof_cap, prof_n, prof_every, prof_seen[2];
    hipEvent_t *prof_ev;         // 2 * prof_cap events
    unsigned char *prof_kind;    // 0 forward, 1 inverse
    int capturing;               // between fz_graph_begin and fz_graph_end: nothing may allocate or synchronise
    // fused verification: per-aggregate fp64 accumulators of `observed` [groups][degree] and state words
    // (arrival / failure counts); all zero between launches -- the kernel re-arms them itself
    double *d_vpart;
    int *d_vstate;
    size_t vpart_doubles, vstate_groups;
    int verify_dirty;            // a verify launch failed: accumulators / state words are re-zeroed before the next one
    // one-pass aggregation: per (aggregate, column block) accumulator words [tiles][tile_words] (running sum + arrival
    // count, see aggregate_onepass); all zero between launches (the last adder of a word re-arms it)
    unsigned long long *d_aggacc;
    size_t aggacc_tiles;
    int agg_dirty;
    uint32_t *d_mt_init;         // MT19937 state after init_genrand(19650218) (fz_sample_secret_polys_dev), lazily
    uint32_t *d_chal_tab;        // weight table of the challenge decoder (fz_challenge.hip), built on first use
    int chal_tab_ib, chal_tab_degree;
    // Pinned, device-visible staging for the SMALL host inputs of the fused challenge kernel (message bytes + offsets, or the
    // digests): the kernel reads them in place over the host link (three coalesced requests per wave), so a call uploads
    // nothing and does not synchronise.  Two slots in turn; a slot's event is recorded after the launch that reads it and
    // waited for before the slot is written again.
    struct FzStage { uint8_t *h; size_t bytes; hipEvent_t ev; int busy; };
    FzStage chal_stage[2];
    int chal_stage_next;
    // benchmarking knobs, read ONCE at context creation
    int knob_agg_direct;         // FZ_AGG_DIRECT: -1 = never the slice-free aggregation kernel, 2 | 4 = always, with that many rows per tile (0 = by size)
    int knob_shake_full;         // FZ_SHAKE_FORM: 1 = lane pairs, 2 = whole state per lane, 3 = a wave per signer (0 = by batch size)
    int knob_verify_ordered;     // FZ_VERIFY_ORDERED=1 (and every device that is not gfx950): acquire / release on verify_fused's arrival atomic
    int knob_polymul_form;       // FZ_POLYMUL_FORM: 1 = the radix-4 product kernel, 2 = the 16-per-lane one (0 = by batch size)
    int knob_unfused;            // FZ_UNFUSED=1: the multi-launch paths of keygen / verification / the coefficient-domain product (what degrees other than 64 / 256 take anyway)
    hipStream_t diag_stream;     // fz_diag_shader_clock: the probe's private stream and result words (created on first use)
    unsigned long long *d_diag;
    int knob_matvec_slices;      // FZ_MATVEC_SLICES = 1 | 2 | 4: k-range slices per column of the integer matvec kernel (0 = by batch size, -1 = the fp64 kernel)
    int knob_no_imad;            // FZ_NO_IMAD=1: A (.) y through the general fp64 multiply instead of integer multiply-adds (A/B runs)
    int knob_verify_cent;        // FZ_VERIFY_CENT=1: centre the inverse transform's outputs before the norm test even when beta allows skipping it
    // device allocations replaced by a larger one while a captured graph may still hold their address: kept until
    // fz_ctx_destroy (a replay must never touch freed memory)
    int graphs_captured;
    void **retired;
    int n_retired, cap_retired;
    // fz_malloc / fz_free keep large blocks for reuse (hipFree of anything from 16 MiB up costs ~180 us AND synchronises the
    // device): live blocks with their sizes, and the blocks handed back, at most pool_cap bytes of them (FZ_POOL_MB, 0 = off)
    // `ev`: recorded on the context's stream when the block came back (fz_free), waited for by the stream that reuses it.
    // The arrays are guarded by pool_mu: a DeviceBuffer.__del__ may run on any thread (cyclic garbage collection).
    struct FzBlock { void *p; size_t bytes; hipEvent_t ev; };
    FzBlock *live_blocks, *pool_blocks;
    int n_live, cap_live, n_pool, cap_pool;
    size_t pool_bytes, pool_cap;
    std::mutex pool_mu;
    // RCCL (fz_comm_*): communicators are owned by the caller; nothing here
    // fz_diag_stamps_*: device-side launch timestamps of the multi-job transform ({entry, exit} of the 100 MHz reference counter
    // per workgroup); launch k of the recording owns slots [stamp_first[k], stamp_first[k] + stamp_count[k])
    int stamp_on, stamp_n, stamp_launch_cap;
    size_t stamp_used, stamp_wg_cap;
    unsigned long long *d_stamp;
    size_t *stamp_first;
    unsigned *stamp_count;
};

// group table of a ragged aggregation launch (kernarg segment): signers of aggregate g are rows [off[g], off[g+1]) of the
// concatenated signature / coefficient arrays; base / extra = its signers per slice and the remainder
constexpr int kFzRaggedMax = 64;
struct FzRagged {
    unsigned off[kFzRaggedMax + 1];
    unsigned base[kFzRaggedMax], extra[kFzRaggedMax];
};

// the job table of one fz_ntt_multi launch travels in the kernarg segment (no device copy, capturable in a graph)
constexpr int kFzMultiMax = 32;
struct FzMultiJobs {
    const int32_t *in[kFzMultiMax];
    int32_t *out[kFzMultiMax];
    unsigned end[kFzMultiMax];   // running total of WORKGROUPS up to and including job j (filled by fz_launch_ntt_multi)
    unsigned rows[kFzMultiMax];  // rows of job j; bit 31 set: inverse transform
    int n;
};

// the table as the kernel takes it: N = 4 | 8 | 32 entries, 24 bytes of kernel arguments each (a launch of four jobs
// carries 96 bytes of table instead of 772)
template <int N> struct FzJobsN {
    static constexpr int kJobs = N;
    const int32_t *in[N];
    int32_t *out[N];
    unsigned end[N];
    unsigned rows[N];
};

struct fz_graph {
    hipGraph_t graph;
    hipGraphExec_t exec;
    int device;
};

// largest transform length: up to 256 the register / LDS schedules of fz_ntt.hip, beyond it one workgroup per polynomial through LDS
constexpr int kFzMaxDegree = 4096;

// error plumbing (fz_capi.hip)
int fz_set_error(int code, const char *fmt, ...);
int fz_check_hip(hipError_t e, const char *what);
int fz_scratch(fz_ctx *ctx, size_t bytes, void **out);
int fz_scratch2(fz_ctx *ctx, size_t bytes, void **out);
int fz_verify_scratch(fz_ctx *ctx, size_t groups, size_t doubles_per_group, double **part, int **state);
int fz_agg_scratch(fz_ctx *ctx, size_t tiles, size_t tile_words, unsigned long long **acc);
int fz_retire(fz_ctx *ctx, void *d_ptr, const char *what);       // hipFree, or keep until destroy when graphs were captured

// launchers (fz_ntt.hip)
int fz_launch_ntt(fz_ctx *ctx, const int32_t *d_in, int32_t *d_out, size_t batch, bool inverse);
int fz_ntt_query_grid(fz_ctx *ctx);
int fz_launch_ntt_multi(fz_ctx *ctx, FzMultiJobs &jobs);           // degree 64 / 256; fills jobs.end
int fz_launch_diag_clock(hipStream_t stream, unsigned long long ticks, unsigned long long *d_out);
int fz_launch_diag(fz_ctx *ctx, int what, const void *src, void *dst, size_t bytes);

// challenge pipeline on the device (fz_challenge.hip) and the pieces it shares with the host serialiser (fz_host.cpp)
struct fz_scheme_params;
int fz_launch_challenge(fz_ctx *ctx, const fz_scheme_params *P, const int32_t *d_vk, const uint8_t *d_pre, const uint32_t *d_dec, size_t N,
                        uint8_t *d_text, size_t text_stride, int *d_nblocks, uint32_t *d_xof, size_t xstride, int out_blocks,
                        const uint32_t *d_tab, int32_t *d_coefs);
void fz_mt_init_table(uint32_t *h_tab);                                              // 624 words
// d_state: [2 * nkeys][624] words of scratch for the two-kernel form, or NULL for the one-kernel form
int fz_launch_mt_sample(fz_ctx *ctx, const unsigned long long *d_seeds, size_t nkeys, int degree, uint32_t bound, int kbits,
                        const uint32_t *d_init, int32_t *d_out, int *d_fail, uint32_t *d_state);
bool fz_challenge_wave_ok(const fz_scheme_params *P);        // the fused one-wave-per-signer form takes these parameters
int fz_launch_challenge_wave(fz_ctx *ctx, const fz_scheme_params *P, const int32_t *d_vk, const uint8_t *d_pre, const uint8_t *d_msgs,
                             const unsigned long long *d_off, uint8_t *d_pre_out, size_t N, size_t text_stride, int out_blocks,
                             const uint32_t *d_tab, int32_t *d_coefs);
int fz_launch_prehash(fz_ctx *ctx, const fz_scheme_params *P, const uint8_t *d_msgs, const unsigned long long *d_off, size_t N,
                      uint8_t *d_pre, uint32_t *d_dec);          // d_dec [N][16]: the integers in base 10^9 + chunk count
void fz_challenge_weight_table(int index_bytes, int degree, uint32_t *h_tab);      // (degree + 1) * 16 words
void fz_host_vk_text_parts(const fz_scheme_params *P, char *s0, int *n0, char *s1, int *n1, char *s2, int *n2, int cap);
size_t fz_host_challenge_needed_bytes(const fz_scheme_params *P, int *sign_bytes, int *coef_bytes, int *index_bytes);
bool fz_host_params_ok(const fz_scheme_params *P);

int fz_launch_polymul_fused(fz_ctx *ctx, const int32_t *f, const int32_t *g, int32_t *out, size_t batch);
bool fz_polymul16_ok(const fz_ctx *ctx, const int32_t *f, const int32_t *g, const int32_t *out, size_t batch);
int fz_launch_keygen_fused(fz_ctx *ctx, const int32_t *A, const int32_t *coef, int32_t *sk_hat, int32_t *vk, size_t segments,
                           int l, bool broadcast = false);
int fz_launch_fill_synthetic(fz_ctx *ctx, int32_t *out, size_t count, unsigned long long seed);
int fz_launch_bcast_rows(fz_ctx *ctx, const int32_t *in, int32_t *out, size_t segments, int l);
int fz_launch_verify_fused_i64(fz_ctx *ctx, const int32_t *A, const int64_t *sig, size_t sig_stride, const int64_t *target,
                               size_t target_stride, size_t groups, int l, int64_t beta, int64_t omega, int *d_verdict);
int fz_launch_verify_fused(fz_ctx *ctx, const int32_t *A, const int32_t *sig, const int32_t *target, size_t groups, int l,
                           int64_t beta, int64_t omega, int *d_verdict);

// launchers (fz_pointwise.hip)
enum { FZ_OP_MUL = 0, FZ_OP_ADD = 1, FZ_OP_SUB = 2, FZ_OP_NEG = 3, FZ_OP_MULACC = 4 };
int fz_launch_pw(fz_ctx *ctx, int op, const int32_t *a, const int32_t *b, int32_t *out, size_t count);
int fz_launch_pw_bcast(fz_ctx *ctx, const int32_t *a, const int32_t *s, int32_t *out, size_t rows);
int fz_launch_matvec(fz_ctx *ctx, const int32_t *A, const int32_t *S, int32_t *out, size_t batch, int l);
int fz_launch_sign(fz_ctx *ctx, const int32_t *sk_hat, const int32_t *c_hat, int32_t *sig, size_t batch, int l);
int fz_launch_aggregate(fz_ctx *ctx, const int32_t *sig, const int32_t *alpha, int64_t *out64, size_t pstride,
                        int32_t *out32, size_t groups, size_t N, int l, const int32_t *vkL = nullptr,
                        const int32_t *vkR = nullptr, const int32_t *c = nullptr, int64_t *tout64 = nullptr, size_t tstride = 0,
                        const size_t *h_offsets = nullptr, const int32_t *sk_hat = nullptr, int32_t *sig_out = nullptr);
int fz_launch_target_partial(fz_ctx *ctx, const int32_t *vkL, const int32_t *vkR, const int32_t *c, const int32_t *alpha,
                             int64_t *partial, size_t pstride, size_t groups, size_t N);
int fz_launch_reduce_i64(fz_ctx *ctx, const int64_t *in, int32_t *out, size_t count);
int fz_launch_norm_weight(fz_ctx *ctx, const int32_t *coef, size_t batch, int64_t *max_abs, int32_t *weight);
int fz_launch_verdict(fz_ctx *ctx, const int32_t *target, const int32_t *observed, const int64_t *max_abs,
                      const int32_t *weight, size_t groups, int l, int64_t beta, int64_t omega, int *d_verdict);

#endif
