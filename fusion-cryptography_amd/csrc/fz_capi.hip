// fz_capi.hip -- the C ABI of libfusion_hip.so (include/fusion_hip.h): context management,
// table construction, error reporting and the thin wrappers that enqueue kernels.
#include "fz_internal.h"
#include "../../include/fusion_hip.h"
#include "../../include/fusion_hip_diag.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <algorithm>
#include <vector>
#include <string>
#include <dlfcn.h>
#include <atomic>
#include <mutex>

// block pool + context registry (defined with fz_malloc / fz_free below)
static void pool_release_locked(fz_ctx *ctx, size_t keep);
static void fz_registry_add(fz_ctx *c);
static void fz_registry_remove(fz_ctx *c);

static thread_local char g_err[512] = "";

int fz_set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int fz_check_hip(hipError_t e, const char *what) {
    if (e == hipSuccess) return FZ_OK;
    return fz_set_error(FZ_E_HIP, "%s: %s", what, hipGetErrorString(e));
}

#define FZ_TRY(x) do { int rc_ = (x); if (rc_ != FZ_OK) return rc_; } while (0)
#define FZ_HIP(x, what) FZ_TRY(fz_check_hip((x), what))
#define FZ_REQUIRE(cond, ...) do { if (!(cond)) return fz_set_error(FZ_E_BADARG, __VA_ARGS__); } while (0)
// Every entry point that touches the device makes the context's device current first: a process may hold contexts on
// several GPUs (and other code -- torch -- may have changed the current device behind our back); kernels, scratch
// allocations and events must land on ctx->device whatever stream the caller attached.
#define FZ_DEV(ctx) FZ_HIP(hipSetDevice((ctx)->device), "hipSetDevice")

// Is work on this context being recorded rather than executed?  Either the context opened a capture itself (fz_graph_begin) or
// its stream was drawn into another context's capture by fz_event_wait on an event recorded there (the fork / join of a
// two-stream capture): the runtime knows, so ask it -- nothing may allocate, copy to the host or synchronise in either case.
static bool fz_capturing(fz_ctx *ctx) {
    if (ctx->capturing) return true;
    if (!ctx->stream) return false;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(ctx->stream, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
    return st == hipStreamCaptureStatusActive;
}

static uint64_t powmod_u64(uint64_t b, uint64_t e, uint64_t q) {
    unsigned __int128 r = 1, x = b % q;
    while (e) {
        if (e & 1) r = (r * x) % q;
        x = (x * x) % q;
        e >>= 1;
    }
    return (uint64_t)r;
}

static unsigned bitrev(unsigned i, int k) {
    unsigned r = 0;
    for (int b = 0; b < k; ++b) r |= ((i >> b) & 1u) << (k - 1 - b);
    return r;
}

// A device allocation that is being replaced by a larger one.  A graph captured on this context may hold its address
// (fz_graph_*: recorded pointers are fixed), and a replay must never touch freed memory: once any graph was captured the
// old allocation is kept until fz_ctx_destroy instead of being freed.
int fz_retire(fz_ctx *ctx, void *d_ptr, const char *what) {
    if (!d_ptr) return FZ_OK;
    if (!ctx->graphs_captured) return fz_check_hip(hipFree(d_ptr), what);
    if (ctx->n_retired == ctx->cap_retired) {
        const int cap = ctx->cap_retired ? 2 * ctx->cap_retired : 16;
        void **r = (void **)realloc(ctx->retired, sizeof(void *) * (size_t)cap);
        if (!r) return fz_set_error(FZ_E_HIP, "out of host memory");
        ctx->retired = r;
        ctx->cap_retired = cap;
    }
    ctx->retired[ctx->n_retired++] = d_ptr;
    return FZ_OK;
}

int fz_scratch(fz_ctx *ctx, size_t bytes, void **out) {
    if (bytes > ctx->scratch_bytes) {
        if (fz_capturing(ctx))
            return fz_set_error(FZ_E_BADARG, "scratch would grow during graph capture: run the sequence once before fz_graph_begin");
        // previous users of the scratch are stream-ordered before this point
        FZ_HIP(hipStreamSynchronize(ctx->stream), "scratch sync");
        FZ_TRY(fz_retire(ctx, ctx->d_scratch, "scratch free"));
        ctx->d_scratch = nullptr;
        ctx->scratch_bytes = 0;
        size_t want = bytes + bytes / 4 + 4096;
        FZ_HIP(hipMalloc(&ctx->d_scratch, want), "scratch alloc");
        ctx->scratch_bytes = want;
    }
    *out = ctx->d_scratch;
    return FZ_OK;
}

int fz_scratch2(fz_ctx *ctx, size_t bytes, void **out) {
    if (bytes > ctx->scratch2_bytes) {
        if (fz_capturing(ctx))
            return fz_set_error(FZ_E_BADARG, "scratch would grow during graph capture: run the sequence once before fz_graph_begin");
        FZ_HIP(hipStreamSynchronize(ctx->stream), "scratch2 sync");
        FZ_TRY(fz_retire(ctx, ctx->d_scratch2, "scratch2 free"));
        ctx->d_scratch2 = nullptr;
        ctx->scratch2_bytes = 0;
        size_t want = bytes + bytes / 4 + 4096;
        FZ_HIP(hipMalloc(&ctx->d_scratch2, want), "scratch2 alloc");
        ctx->scratch2_bytes = want;
    }
    *out = ctx->d_scratch2;
    return FZ_OK;
}

int fz_verify_scratch(fz_ctx *ctx, size_t groups, size_t doubles_per_group, double **part, int **state) {
    const size_t need = groups * doubles_per_group;
    if (need > ctx->vpart_doubles || groups > ctx->vstate_groups) {
        if (fz_capturing(ctx))
            return fz_set_error(FZ_E_BADARG, "verify scratch would grow during graph capture: run the sequence once before fz_graph_begin");
        FZ_HIP(hipStreamSynchronize(ctx->stream), "verify scratch sync");
        if (need > ctx->vpart_doubles) {
            FZ_TRY(fz_retire(ctx, ctx->d_vpart, "verify scratch free"));
            ctx->d_vpart = nullptr;
            ctx->vpart_doubles = 0;
            FZ_HIP(hipMalloc((void **)&ctx->d_vpart, (need + need / 4) * sizeof(double)), "verify scratch alloc");
            // on the context's stream: a null-stream memset is not ordered with a non-blocking stream (found by tools/soak.py)
            FZ_HIP(hipMemsetAsync(ctx->d_vpart, 0, (need + need / 4) * sizeof(double), ctx->stream), "verify scratch clear");
            ctx->vpart_doubles = need + need / 4;
        }
        if (groups > ctx->vstate_groups) {
            FZ_TRY(fz_retire(ctx, ctx->d_vstate, "verify state free"));
            ctx->d_vstate = nullptr;
            ctx->vstate_groups = 0;
            const size_t cap = groups + groups / 4 + 16;
            FZ_HIP(hipMalloc((void **)&ctx->d_vstate, cap * 2 * sizeof(int)), "verify state alloc");
            FZ_HIP(hipMemsetAsync(ctx->d_vstate, 0, cap * 2 * sizeof(int), ctx->stream), "verify state clear");
            ctx->vstate_groups = cap;
        }
    }
    if (ctx->verify_dirty) {                       // an earlier launch failed: do not trust "zero between launches"
        if (fz_capturing(ctx)) return fz_set_error(FZ_E_BADARG, "verify scratch must be re-zeroed: not during graph capture");
        FZ_HIP(hipMemsetAsync(ctx->d_vpart, 0, ctx->vpart_doubles * sizeof(double), ctx->stream), "verify scratch clear");
        FZ_HIP(hipMemsetAsync(ctx->d_vstate, 0, ctx->vstate_groups * 2 * sizeof(int), ctx->stream), "verify state clear");
        ctx->verify_dirty = 0;
    }
    *part = ctx->d_vpart;
    *state = ctx->d_vstate;
    return FZ_OK;
}

// accumulator words of the one-pass aggregation (zero between launches; see aggregate_onepass)
int fz_agg_scratch(fz_ctx *ctx, size_t tiles, size_t tile_words, unsigned long long **acc) {
    if (tiles > ctx->aggacc_tiles) {
        if (fz_capturing(ctx))
            return fz_set_error(FZ_E_BADARG, "aggregation scratch would grow during graph capture: run the sequence once before fz_graph_begin");
        FZ_HIP(hipStreamSynchronize(ctx->stream), "aggregation scratch sync");
        FZ_TRY(fz_retire(ctx, ctx->d_aggacc, "aggregation scratch free"));     // a captured aggregation keeps a valid (if stale) accumulator
        ctx->d_aggacc = nullptr;
        ctx->aggacc_tiles = 0;
        const size_t cap = tiles + tiles / 4 + 8;
        FZ_HIP(hipMalloc((void **)&ctx->d_aggacc, cap * tile_words * sizeof(unsigned long long)), "aggregation scratch alloc");
        ctx->aggacc_tiles = cap;
        ctx->agg_dirty = 1;
    }
    if (ctx->agg_dirty) {
        if (fz_capturing(ctx)) return fz_set_error(FZ_E_BADARG, "aggregation scratch must be re-zeroed: not during graph capture");
        FZ_HIP(hipMemsetAsync(ctx->d_aggacc, 0, ctx->aggacc_tiles * tile_words * sizeof(unsigned long long), ctx->stream), "aggregation scratch clear");
        ctx->agg_dirty = 0;
    }
    *acc = ctx->d_aggacc;
    return FZ_OK;
}

extern "C" {

const char *fz_version(void) { return "fusion_hip 0.1.0 (gfx950)"; }
const char *fz_last_error(void) { return g_err; }

int fz_device_count(int *out_count) {
    FZ_REQUIRE(out_count, "out_count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *out_count = 0;
        return fz_set_error(FZ_E_NODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *out_count = n;
    return FZ_OK;
}

static int upload_doubles(const double *h, size_t n, double **d_out) {
    FZ_HIP(hipMalloc((void **)d_out, (n ? n : 1) * sizeof(double)), "table alloc");
    if (n) FZ_HIP(hipMemcpy(*d_out, h, n * sizeof(double), hipMemcpyHostToDevice), "table upload");
    return FZ_OK;
}

// h_fwd / h_inv != NULL: the context's twiddle tables are THESE (fz_ctx_create_tables) instead of the bit-reversed powers of a root
static int ctx_create(int device_id, uint32_t q, int degree, uint32_t root, uint32_t inv_root, const uint32_t *h_fwd, const uint32_t *h_inv,
                      fz_ctx **out) {
    FZ_REQUIRE(out, "out is NULL");
    *out = nullptr;
    // any odd modulus below 2^32: centred residues |x| <= (q - 1) / 2 < 2^31 are int32 whatever q is, and every bound of
    // fz_arith.h is in terms of 2^31-sized operands and twiddles below 2^32 (round 5; rounds 1-4 refused q >= 2^31)
    FZ_REQUIRE(q >= 3 && (q & 1u), "modulus %u must be odd and >= 3", q);
    // root == 0: "ring-only" context (pointwise ops, norm/weight, matvec on rows of `degree` values;
    // no transform tables).  The reference lets polynomial objects exist for parameter tuples that
    // admit no NTT (e.g. root_order 1), and their + - * norm weight still work.
    const bool tables = h_fwd != nullptr;
    const bool ring_only = !tables && (root == 0);
    if (ring_only) {
        FZ_REQUIRE(degree >= 1 && degree <= (1 << 20), "degree %d out of range", degree);
    } else {
        FZ_REQUIRE(degree >= 2 && (degree & (degree - 1)) == 0, "degree %d must be a power of two >= 2", degree);
        if (degree > kFzMaxDegree) return fz_set_error(FZ_E_UNSUPPORTED, "degree %d > %d not supported by the NTT kernels", degree, kFzMaxDegree);
        if (tables) {
            FZ_REQUIRE(h_inv, "both tables are required");
        } else {
            FZ_REQUIRE(((uint64_t)q - 1) % (2u * (uint64_t)degree) == 0, "2*degree=%d does not divide q-1", 2 * degree);
            FZ_REQUIRE(root > 0 && root < q && inv_root > 0 && inv_root < q, "root / inv_root must be in (0, q)");
            // primitive 2*degree-th root (order a power of two): root^degree == -1
            FZ_REQUIRE(powmod_u64(root, (uint64_t)degree, q) == (uint64_t)q - 1,
                       "root %u is not a primitive %d-th root of unity mod %u", root, 2 * degree, q);
            FZ_REQUIRE(((uint64_t)root * inv_root) % q == 1, "root * inv_root != 1 mod q");
        }
    }

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fz_set_error(FZ_E_NODEVICE, "no HIP device available");
    FZ_REQUIRE(device_id >= 0 && device_id < ndev, "device_id %d out of range (0..%d)", device_id, ndev - 1);
    FZ_HIP(hipSetDevice(device_id), "hipSetDevice");

    fz_ctx *c = new (std::nothrow) fz_ctx();
    if (!c) return fz_set_error(FZ_E_HIP, "out of host memory");
    memset(c, 0, sizeof(*c));
    c->device = device_id;
    hipDeviceProp_t prop;
    int rc = fz_check_hip(hipGetDeviceProperties(&prop, device_id), "hipGetDeviceProperties");
    if (rc != FZ_OK) { delete c; return rc; }
    c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    c->q = q; c->root = root; c->inv_root = inv_root;
    c->degree = degree;
    c->logd = ring_only ? -1 : 0;
    if (!ring_only) while ((1 << c->logd) < degree) ++c->logd;
    c->mod = fz_make_mod(q);

    double *tw = nullptr, *itw = nullptr, *twB = nullptr, *itwB = nullptr;
    size_t nB = 0;
    const int n = ring_only ? 0 : degree, k = c->logd;
    if (!ring_only) {
        c->h_tw = (uint32_t *)malloc(sizeof(uint32_t) * n);
        c->h_itw = (uint32_t *)malloc(sizeof(uint32_t) * n);
        tw = (double *)malloc(sizeof(double) * n);
        itw = (double *)malloc(sizeof(double) * n);
        for (int i = 0; i < n; ++i) {
            // bit_reverse_copy([pow(root, i, q)])  (algebra/polynomials.py:396-397, :416-417) -- or whatever table the caller
            // hands to cooley_tukey_ntt / gentleman_sande_intt (ntt.py:274-290, :354-372 use it as it is)
            c->h_tw[i] = tables ? h_fwd[i] % q : (uint32_t)powmod_u64(root, bitrev((unsigned)i, k), q);
            c->h_itw[i] = tables ? h_inv[i] % q : (uint32_t)powmod_u64(inv_root, bitrev((unsigned)i, k), q);
            tw[i] = (double)c->h_tw[i];
            itw[i] = (double)c->h_itw[i];
        }
        const uint64_t n_inv = powmod_u64((uint64_t)n, (uint64_t)q - 2, q);
        for (int i = 0; i < 16; ++i) {
            c->twA.w[i] = (i < n) ? tw[i] : 0.0;
            c->itwA.w[i] = (i < n) ? itw[i] : 0.0;
            c->twA.w2[i] = c->twA.w[i] * c->mod.kq;
            c->itwA.w2[i] = c->itwA.w[i] * c->mod.kq;
        }
        c->twA.n_inv = c->itwA.n_inv = (double)n_inv;
        c->twA.w1_n_inv = 0.0;
        c->itwA.w1_n_inv = (double)(((unsigned __int128)c->h_itw[1] * n_inv) % q);
        c->twA.n_inv2 = c->itwA.n_inv2 = c->itwA.n_inv * c->mod.kq;
        c->twA.w1_n_inv2 = 0.0;
        c->itwA.w1_n_inv2 = c->itwA.w1_n_inv * c->mod.kq;

        // per-lane tables of the contiguous pass ([NE][L]); see fz_ntt.hip / tools/ntt_layout_model.py
        if (k >= 5 && k <= 8) {
            const int L = n / 16, SB = k - 4, NE = 16 - (16 >> SB);
            nB = (size_t)NE * L * 2;                       // (w, w * K / q) pairs
            twB = (double *)malloc(sizeof(double) * nB);
            itwB = (double *)malloc(sizeof(double) * nB);
            for (int ls = 0; ls < SB; ++ls) {
                {   // forward: distance 2^(SB-1-ls), ng groups per lane
                    const int t = 1 << (SB - 1 - ls), ng = 16 / (2 * t);
                    const int ebase = (16 >> SB) * ((1 << ls) - 1);
                    for (int g = 0; g < ng; ++g)
                        for (int b = 0; b < L; ++b) {
                            const double w = tw[(16 << ls) + b * ng + g];
                            twB[((size_t)(ebase + g) * L + b) * 2] = w;
                            twB[((size_t)(ebase + g) * L + b) * 2 + 1] = w * c->mod.kq;
                        }
                }
                {   // inverse: distance 2^ls
                    const int ng = 8 >> ls, ebase = 16 - (16 >> ls);
                    for (int g = 0; g < ng; ++g)
                        for (int b = 0; b < L; ++b) {
                            const double w = itw[(n >> (ls + 1)) + b * ng + g];
                            itwB[((size_t)(ebase + g) * L + b) * 2] = w;
                            itwB[((size_t)(ebase + g) * L + b) * 2 + 1] = w * c->mod.kq;
                        }
                }
            }
        }
    }

    rc = fz_check_hip(hipEventCreate(&c->ev0), "event create");
    if (rc == FZ_OK) rc = fz_check_hip(hipEventCreate(&c->ev1), "event create");
    if (rc == FZ_OK) rc = upload_doubles(tw, n, &c->d_tw);
    if (rc == FZ_OK) rc = upload_doubles(itw, n, &c->d_itw);
    if (rc == FZ_OK && !ring_only) {
        double *pairs = (double *)malloc(sizeof(double) * 4 * (size_t)n);
        for (int i = 0; i < n; ++i) {
            pairs[2 * i] = tw[i];
            pairs[2 * i + 1] = tw[i] * c->mod.kq;
            pairs[2 * n + 2 * i] = itw[i];
            pairs[2 * n + 2 * i + 1] = itw[i] * c->mod.kq;
        }
        rc = upload_doubles(pairs, 2 * (size_t)n, &c->d_tw2);
        if (rc == FZ_OK) rc = upload_doubles(pairs + 2 * n, 2 * (size_t)n, &c->d_itw2);
        free(pairs);
    }
    {
        // every benchmarking / test knob is read HERE, once: no entry point consults the environment afterwards (DESIGN.md
        // section 10 lists them; round 4 removed the knobs of closed experiments together with their instantiations)
        auto knob = [](const char *name) { const char *v = getenv(name); return v ? atoi(v) : 0; };
        c->force_kernel = knob("FZ_NTT_KERNEL");
        c->knob_ntt_rows = knob("FZ_NTT_ROWS");
        // measured crossover, inputs NOT cache-resident, both schedules on one box: degree 256 -- the radix-4 kernels lead up to
        // 2^14 rows (4.23 / 5.58 / 8.47 us at 2^12 .. 2^14 against 4.97 / 6.53 / 9.10 for the 16-per-lane kernel), the 16-per-lane
        // kernel from 24 576 rows (6 x 4096: 13.7 us against 14.2; 2^15: 14.6 against 15.6) -- round 5's kernel, whose start-up
        // overlaps the first chunk with the twiddle table (profiles/r05_ntt_crossover.txt; rounds 3-4: from 2^16);
        // degree 64 -- radix-4 up to 2^18 rows (round 2's measurement)
        c->small_batch_rows = degree == 256 ? (3 << 13) : (1 << 19);
        c->knob_agg_direct = knob("FZ_AGG_DIRECT");
        c->knob_shake_full = knob("FZ_SHAKE_FORM");
        c->knob_verify_ordered = knob("FZ_VERIFY_ORDERED");
        // The fence-free cross-workgroup combine of verify_fused (relaxed agent-scope atomics on the library's own
        // coarse-grained scratch, ordered by data dependence: csrc/fz_ntt.hip) is an argument about THIS chip's memory-side
        // atomics; anything that does not report gfx950 gets the acquire/release instantiation.
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) c->knob_verify_ordered = 1;
        c->knob_unfused = knob("FZ_UNFUSED");
        c->knob_polymul_form = knob("FZ_POLYMUL_FORM");
        c->knob_no_imad = knob("FZ_NO_IMAD");
        c->knob_matvec_slices = knob("FZ_MATVEC_SLICES");
        c->knob_verify_cent = knob("FZ_VERIFY_CENT");
        // fz_malloc's block pool: FZ_POOL_MB megabytes at most over all contexts of the process (default 4096, 0 = every fz_free is a hipFree)
        c->pool_cap = (size_t)(getenv("FZ_POOL_MB") ? (knob("FZ_POOL_MB") < 0 ? 0 : knob("FZ_POOL_MB")) : 4096) << 20;
    }
    if (rc == FZ_OK) rc = upload_doubles(twB, nB, &c->d_twB);
    if (rc == FZ_OK) rc = upload_doubles(itwB, nB, &c->d_itwB);
    if (rc == FZ_OK && nB) {
        static_assert(sizeof(FzTwA) == 36 * sizeof(double), "FzTwA is 36 doubles");
        const FzTwA both[2] = {c->twA, c->itwA};
        rc = upload_doubles(reinterpret_cast<const double *>(both), 72, &c->d_twAB);
    }
    if (rc == FZ_OK) rc = fz_check_hip(hipMalloc((void **)&c->d_verdict, 64 * sizeof(int)), "verdict alloc");
    c->verdict_cap = 64;
    if (rc == FZ_OK && !ring_only) rc = fz_ntt_query_grid(c);
    free(tw); free(itw); free(twB); free(itwB);
    if (rc != FZ_OK) { fz_ctx_destroy(c); return rc; }
    fz_registry_add(c);
    *out = c;
    return FZ_OK;
}

int fz_ctx_create(int device_id, uint32_t q, int degree, uint32_t root, uint32_t inv_root, fz_ctx **out) {
    return ctx_create(device_id, q, degree, root, inv_root, nullptr, nullptr, out);
}

// cooley_tukey_ntt / gentleman_sande_intt take the twiddle table as an ARGUMENT and use whatever they are handed
// (algebra/ntt.py:274-290, :354-372: `s = bit_rev_root_powers[m + i]`): a context whose tables are the caller's own lists --
// not necessarily the powers of one root -- runs the same butterfly network on them.  Entries are reduced mod q.
int fz_ctx_create_tables(int device_id, uint32_t q, int degree, const uint32_t *h_fwd, const uint32_t *h_inv, fz_ctx **out) {
    FZ_REQUIRE(h_fwd && h_inv, "both tables are required (pass the same one twice when only one direction is used)");
    return ctx_create(device_id, q, degree, 0, 0, h_fwd, h_inv, out);
}

int fz_ctx_destroy(fz_ctx *ctx) {
    if (!ctx) return FZ_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->d_tw) (void)hipFree(ctx->d_tw);
    if (ctx->d_itw) (void)hipFree(ctx->d_itw);
    if (ctx->d_tw2) (void)hipFree(ctx->d_tw2);
    if (ctx->d_itw2) (void)hipFree(ctx->d_itw2);
    if (ctx->d_twB) (void)hipFree(ctx->d_twB);
    if (ctx->d_itwB) (void)hipFree(ctx->d_itwB);
    if (ctx->d_twAB) (void)hipFree(ctx->d_twAB);
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    if (ctx->d_scratch2) (void)hipFree(ctx->d_scratch2);
    if (ctx->d_verdict) (void)hipFree(ctx->d_verdict);
    if (ctx->prof_ev) {
        for (int i = 0; i < 2 * ctx->prof_cap; ++i) (void)hipEventDestroy(ctx->prof_ev[i]);
        free(ctx->prof_ev);
        free(ctx->prof_kind);
    }
    if (ctx->d_vpart) (void)hipFree(ctx->d_vpart);
    if (ctx->d_vstate) (void)hipFree(ctx->d_vstate);
    if (ctx->d_aggacc) (void)hipFree(ctx->d_aggacc);
    for (int i = 0; i < ctx->n_retired; ++i) (void)hipFree(ctx->retired[i]);
    free(ctx->retired);
    fz_registry_remove(ctx);
    {
        std::lock_guard<std::mutex> g(ctx->pool_mu);
        pool_release_locked(ctx, 0);
        for (int i = 0; i < ctx->n_live; ++i)
            if (ctx->live_blocks[i].ev) (void)hipEventDestroy(ctx->live_blocks[i].ev);
    }
    free(ctx->pool_blocks);
    free(ctx->live_blocks);          // (blocks the caller never freed stay the caller's)
    if (ctx->d_chal_tab) (void)hipFree(ctx->d_chal_tab);
    for (auto &st : ctx->chal_stage) {
        if (st.ev) { if (st.busy) (void)hipEventSynchronize(st.ev); (void)hipEventDestroy(st.ev); }
        if (st.h) (void)hipHostFree(st.h);
    }
    if (ctx->d_diag) (void)hipFree(ctx->d_diag);
    if (ctx->diag_stream) (void)hipStreamDestroy(ctx->diag_stream);
    if (ctx->d_mt_init) (void)hipFree(ctx->d_mt_init);
    if (ctx->d_stamp) (void)hipFree(ctx->d_stamp);
    free(ctx->stamp_first);
    free(ctx->stamp_count);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    free(ctx->h_tw);
    free(ctx->h_itw);
    delete ctx;
    return FZ_OK;
}

int fz_ctx_set_stream(fz_ctx *ctx, void *hip_stream) {
    FZ_REQUIRE(ctx, "ctx is NULL");
    if (fz_capturing(ctx)) return fz_set_error(FZ_E_BADARG, "the stream cannot change during graph capture");
    if (ctx->stream != (hipStream_t)hip_stream) {
        // the accumulator words of the one-pass aggregation / fused verification, the scratch areas and the blocks of the
        // pool belong to the context, not to a stream: work still in flight on the old stream must not share them with
        // work on the new one.  Unconditional on every change (a context with only pooled or live blocks used to skip it).
        FZ_DEV(ctx);
        FZ_HIP(hipStreamSynchronize(ctx->stream), "stream change: synchronise the old stream");
    }
    ctx->stream = (hipStream_t)hip_stream;
    return FZ_OK;
}

int fz_ctx_synchronize(fz_ctx *ctx) {
    FZ_REQUIRE(ctx, "ctx is NULL");
    FZ_DEV(ctx);
    if (fz_capturing(ctx)) return fz_set_error(FZ_E_BADARG, "synchronisation is not allowed during graph capture");
    FZ_HIP(hipStreamSynchronize(ctx->stream), "stream synchronize");
    return FZ_OK;
}

int fz_stream_create(fz_ctx *ctx, void **out_stream) {
    FZ_REQUIRE(ctx && out_stream, "NULL argument");
    FZ_DEV(ctx);
    hipStream_t s = nullptr;
    FZ_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "stream create");
    *out_stream = (void *)s;
    return FZ_OK;
}

int fz_stream_create_priority(fz_ctx *ctx, int high, void **out_stream) {
    FZ_REQUIRE(ctx && out_stream, "NULL argument");
    FZ_DEV(ctx);
    int least = 0, greatest = 0;                     // numerically LOWER = higher priority
    FZ_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest), "stream priority range");
    hipStream_t s = nullptr;
    FZ_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, high ? greatest : least), "stream create");
    *out_stream = (void *)s;
    return FZ_OK;
}

int fz_stream_destroy(fz_ctx *ctx, void *hip_stream) {
    FZ_REQUIRE(ctx, "ctx is NULL");
    FZ_DEV(ctx);
    if (!hip_stream) return FZ_OK;
    if (ctx->stream == (hipStream_t)hip_stream) return fz_set_error(FZ_E_BADARG, "the stream is still attached to this context");
    FZ_HIP(hipStreamDestroy((hipStream_t)hip_stream), "stream destroy");
    return FZ_OK;
}

// ---- graph capture: a launch-bound sequence of device-pointer calls recorded once, replayed with one call ------
int fz_graph_begin(fz_ctx *ctx) {
    FZ_REQUIRE(ctx, "ctx is NULL");
    FZ_DEV(ctx);
    if (fz_capturing(ctx)) return fz_set_error(FZ_E_BADARG, "a capture is already open on this context (or its stream has joined another context's)");
    if (ctx->stream == nullptr)
        return fz_set_error(FZ_E_BADARG, "graph capture needs a non-default stream (fz_ctx_set_stream)");
    if (ctx->prof_on) return fz_set_error(FZ_E_BADARG, "per-dispatch profiling is on: events cannot be captured");
    FZ_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    FZ_HIP(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed), "begin capture");
    ctx->capturing = 1;
    return FZ_OK;
}

int fz_graph_end(fz_ctx *ctx, fz_graph **out_graph) {
    FZ_REQUIRE(ctx && out_graph, "NULL argument");
    FZ_DEV(ctx);
    if (!ctx->capturing) return fz_set_error(FZ_E_BADARG, "no capture is open on this context");
    ctx->capturing = 0;
    hipGraph_t g = nullptr;
    FZ_HIP(hipStreamEndCapture(ctx->stream, &g), "end capture");
    if (!g) return fz_set_error(FZ_E_HIP, "the capture produced no graph (a captured call failed)");
    hipGraphExec_t ex = nullptr;
    hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGraphDestroy(g);
        return fz_check_hip(e, "graph instantiate");
    }
    fz_graph *G = new fz_graph;
    ctx->graphs_captured++;
    G->graph = g;
    G->exec = ex;
    G->device = ctx->device;
    *out_graph = G;
    return FZ_OK;
}

int fz_graph_launch(fz_ctx *ctx, fz_graph *graph) {
    FZ_REQUIRE(ctx && graph, "NULL argument");
    FZ_DEV(ctx);
    if (fz_capturing(ctx)) return fz_set_error(FZ_E_BADARG, "a graph cannot be launched into its own capture");
    if (graph->device != ctx->device) return fz_set_error(FZ_E_BADARG, "graph was captured on device %d", graph->device);
    FZ_HIP(hipGraphLaunch(graph->exec, ctx->stream), "graph launch");
    return FZ_OK;
}

int fz_graph_destroy(fz_graph *graph) {
    if (!graph) return FZ_OK;
    (void)hipGraphExecDestroy(graph->exec);
    (void)hipGraphDestroy(graph->graph);
    delete graph;
    return FZ_OK;
}

// ---- events: ordering between the streams of two contexts ------------------------------------------------------------
struct fz_event {
    hipEvent_t ev;
    int device;
};

int fz_event_create(fz_ctx *ctx, fz_event **out) {
    FZ_REQUIRE(ctx && out, "NULL argument");
    *out = nullptr;
    FZ_DEV(ctx);
    fz_event *e = new (std::nothrow) fz_event();
    if (!e) return fz_set_error(FZ_E_HIP, "out of host memory");
    e->device = ctx->device;
    hipError_t rc = hipEventCreateWithFlags(&e->ev, hipEventDisableTiming);
    if (rc != hipSuccess) { delete e; return fz_check_hip(rc, "event create"); }
    *out = e;
    return FZ_OK;
}

int fz_event_record(fz_ctx *ctx, fz_event *ev) {
    FZ_REQUIRE(ctx && ev, "NULL argument");
    if (ev->device != ctx->device) return fz_set_error(FZ_E_BADARG, "event was created on device %d", ev->device);
    FZ_DEV(ctx);
    FZ_HIP(hipEventRecord(ev->ev, ctx->stream), "event record");
    return FZ_OK;
}

int fz_event_wait(fz_ctx *ctx, fz_event *ev) {
    FZ_REQUIRE(ctx && ev, "NULL argument");
    if (ev->device != ctx->device) return fz_set_error(FZ_E_BADARG, "event was created on device %d", ev->device);
    FZ_DEV(ctx);
    FZ_HIP(hipStreamWaitEvent(ctx->stream, ev->ev, 0), "stream wait event");
    return FZ_OK;
}

int fz_event_destroy(fz_event *ev) {
    if (!ev) return FZ_OK;
    (void)hipSetDevice(ev->device);
    hipError_t rc = hipEventDestroy(ev->ev);
    delete ev;
    return fz_check_hip(rc, "event destroy");
}

int fz_ctx_twiddles(fz_ctx *ctx, uint32_t *h_fwd, uint32_t *h_inv) {
    FZ_REQUIRE(ctx, "ctx is NULL");
    if (ctx->logd < 0) return fz_set_error(FZ_E_UNSUPPORTED, "ring-only context has no transform tables");
    if (h_fwd) memcpy(h_fwd, ctx->h_tw, sizeof(uint32_t) * ctx->degree);
    if (h_inv) memcpy(h_inv, ctx->h_itw, sizeof(uint32_t) * ctx->degree);
    return FZ_OK;
}

// Blocks of kPoolMin bytes or more that come back through fz_free are kept and handed out again by fz_malloc for requests
// they fit without wasting more than a quarter: hipFree of a large block takes ~180 us and synchronises the whole device
// (measured: 1 MiB 1 us, 16 MiB - 1 GiB 178-190 us; hipMalloc 10-12 us), which is most of what a 1024-key keygen_batch spent
// outside its kernels.
// Safety of reuse: fz_free records an event on the context's stream and the stream that takes the block out of the pool
// waits for it, so the block's previous users (queued on the context's stream at fz_free time -- the documented requirement
// of fz_free, include/fusion_hip.h) finish before its next ones start, whatever fz_ctx_set_stream did in between (which
// also drains the old stream on every change).  The arrays are guarded by pool_mu.
// Budget: ONE process-wide cap (FZ_POOL_MB, default 4096) over the pools of all contexts -- sixteen private contexts
// (tools/probes/concurrent_batches.py) share it instead of stranding 4 GiB each; fz_pool_trim gives a context's blocks back
// (Context.close calls it); a failed hipMalloc flushes the pools of EVERY context on the device before it retries.
static const size_t kPoolMin = 256 << 10;
static std::mutex g_ctx_mu;                       // registry of live contexts (fz_ctx_create / fz_ctx_destroy)
static fz_ctx *g_ctxs[256];
static int g_nctx = 0;
static std::atomic<size_t> g_pool_bytes{0};       // bytes idle in all pools of the process

static void fz_registry_add(fz_ctx *c) {
    std::lock_guard<std::mutex> g(g_ctx_mu);
    if (g_nctx < 256) g_ctxs[g_nctx++] = c;
}

static void fz_registry_remove(fz_ctx *c) {
    std::lock_guard<std::mutex> g(g_ctx_mu);
    for (int i = 0; i < g_nctx; ++i)
        if (g_ctxs[i] == c) { g_ctxs[i] = g_ctxs[--g_nctx]; break; }
}

static bool grow(fz_ctx::FzBlock *&arr, int &cap, int need) {
    if (need <= cap) return true;
    const int ncap = cap ? 2 * cap : 64;
    fz_ctx::FzBlock *n = (fz_ctx::FzBlock *)realloc(arr, (size_t)ncap * sizeof(fz_ctx::FzBlock));
    if (!n) return false;
    arr = n;
    cap = ncap;
    return true;
}

// pool_mu held: hand pooled blocks back to the runtime, oldest first, until at most `keep` bytes stay
static void pool_release_locked(fz_ctx *ctx, size_t keep) {
    int k = 0;
    while (k < ctx->n_pool && ctx->pool_bytes > keep) {
        fz_ctx::FzBlock &b = ctx->pool_blocks[k++];
        (void)hipFree(b.p);                          // synchronises the device: whatever still used the block has finished
        if (b.ev) (void)hipEventDestroy(b.ev);
        ctx->pool_bytes -= b.bytes;
        g_pool_bytes -= b.bytes;
    }
    for (int i = k; i < ctx->n_pool; ++i) ctx->pool_blocks[i - k] = ctx->pool_blocks[i];
    ctx->n_pool -= k;
}

// a hipMalloc failed: idle blocks of ANY context on this device may be what stands in the way
static void pool_flush_device(int device) {
    std::lock_guard<std::mutex> g(g_ctx_mu);
    for (int i = 0; i < g_nctx; ++i) {
        fz_ctx *c = g_ctxs[i];
        if (c->device != device) continue;
        std::lock_guard<std::mutex> gp(c->pool_mu);
        pool_release_locked(c, 0);
    }
}

int fz_pool_trim(fz_ctx *ctx, size_t keep_bytes) {
    FZ_REQUIRE(ctx, "ctx is NULL");
    FZ_DEV(ctx);
    if (fz_capturing(ctx)) return fz_set_error(FZ_E_BADARG, "the pool cannot be trimmed during graph capture (hipFree synchronises)");
    std::lock_guard<std::mutex> g(ctx->pool_mu);
    pool_release_locked(ctx, keep_bytes);
    return FZ_OK;
}

int fz_malloc(fz_ctx *ctx, size_t bytes, void **d_out) {
    FZ_REQUIRE(ctx && d_out, "NULL argument");
    FZ_DEV(ctx);
    if (bytes == 0) bytes = 1;
    void *p = nullptr;
    hipEvent_t ev = nullptr;
    if (bytes >= kPoolMin && !fz_capturing(ctx)) {       // (a pooled block's event was recorded outside the capture: not waitable inside one)
        std::lock_guard<std::mutex> g(ctx->pool_mu);
        int best = -1;
        for (int i = 0; i < ctx->n_pool; ++i) {
            const size_t b = ctx->pool_blocks[i].bytes;
            if (b >= bytes && b - bytes <= bytes / 4 && (best < 0 || b < ctx->pool_blocks[best].bytes)) best = i;
        }
        if (best >= 0) {
            p = ctx->pool_blocks[best].p;
            bytes = ctx->pool_blocks[best].bytes;
            ev = ctx->pool_blocks[best].ev;
            ctx->pool_bytes -= bytes;
            g_pool_bytes -= bytes;
            for (int k = best + 1; k < ctx->n_pool; ++k) ctx->pool_blocks[k - 1] = ctx->pool_blocks[k];    // keeps age order
            --ctx->n_pool;
        }
    }
    if (p && ev) {
        // the next users of the block run after its previous ones (a no-op when both are on one stream)
        hipError_t e = hipStreamWaitEvent(ctx->stream, ev, 0);
        if (e != hipSuccess) { (void)hipGetLastError(); (void)hipEventSynchronize(ev); }
    }
    if (!p) {
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {                        // out of memory: give every idle block on this device back and try once more
            (void)hipGetLastError();
            pool_flush_device(ctx->device);
            e = hipMalloc(&p, bytes);
        }
        FZ_HIP(e, "hipMalloc");
    }
    if (bytes >= kPoolMin && ctx->pool_cap) {
        std::lock_guard<std::mutex> g(ctx->pool_mu);
        if (!grow(ctx->live_blocks, ctx->cap_live, ctx->n_live + 1)) {
            (void)hipFree(p);
            if (ev) (void)hipEventDestroy(ev);
            return fz_set_error(FZ_E_HIP, "out of host memory");
        }
        ctx->live_blocks[ctx->n_live++] = {p, bytes, ev};
    } else if (ev) {
        (void)hipEventDestroy(ev);
    }
    *d_out = p;
    return FZ_OK;
}

int fz_free(fz_ctx *ctx, void *d_ptr) {
    FZ_REQUIRE(ctx, "ctx is NULL");
    FZ_DEV(ctx);
    if (!d_ptr) return FZ_OK;
    hipEvent_t stale = nullptr;
    {
        std::lock_guard<std::mutex> g(ctx->pool_mu);
        for (int i = ctx->n_live - 1; i >= 0; --i) {
            if (ctx->live_blocks[i].p != d_ptr) continue;
            fz_ctx::FzBlock b = ctx->live_blocks[i];
            ctx->live_blocks[i] = ctx->live_blocks[--ctx->n_live];
            stale = b.ev;
            if (fz_capturing(ctx) || b.bytes > ctx->pool_cap || !grow(ctx->pool_blocks, ctx->cap_pool, ctx->n_pool + 1)) break;
            // room under the process-wide cap: this context's oldest blocks go first; if other contexts hold the rest, do not pool
            if (g_pool_bytes + b.bytes > ctx->pool_cap) {
                const size_t over = g_pool_bytes + b.bytes - ctx->pool_cap;
                pool_release_locked(ctx, ctx->pool_bytes > over ? ctx->pool_bytes - over : 0);
            }
            if (g_pool_bytes + b.bytes > ctx->pool_cap) break;
            if (!b.ev && hipEventCreateWithFlags(&b.ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); b.ev = nullptr; break; }
            if (hipEventRecord(b.ev, ctx->stream) != hipSuccess) { (void)hipGetLastError(); break; }
            ctx->pool_blocks[ctx->n_pool++] = b;
            ctx->pool_bytes += b.bytes;
            g_pool_bytes += b.bytes;
            return FZ_OK;
        }
    }
    if (stale) (void)hipEventDestroy(stale);
    FZ_HIP(hipFree(d_ptr), "hipFree");
    return FZ_OK;
}

int fz_memcpy_h2d(fz_ctx *ctx, void *d_dst, const void *h_src, size_t bytes) {
    FZ_REQUIRE(ctx && (bytes == 0 || (d_dst && h_src)), "NULL argument");
    FZ_DEV(ctx);
    if (bytes) FZ_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream), "memcpy h2d");
    return FZ_OK;
}

int fz_memcpy_d2h(fz_ctx *ctx, void *h_dst, const void *d_src, size_t bytes) {
    FZ_REQUIRE(ctx && (bytes == 0 || (h_dst && d_src)), "NULL argument");
    FZ_DEV(ctx);
    if (fz_capturing(ctx)) return fz_set_error(FZ_E_BADARG, "a synchronous device-to-host copy cannot be captured");
    if (bytes) FZ_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream), "memcpy d2h");
    FZ_HIP(hipStreamSynchronize(ctx->stream), "memcpy d2h sync");
    return FZ_OK;
}

int fz_timer_start(fz_ctx *ctx) {
    FZ_REQUIRE(ctx, "ctx is NULL");
    FZ_DEV(ctx);
    FZ_HIP(hipEventRecord(ctx->ev0, ctx->stream), "event record");
    return FZ_OK;
}

int fz_timer_stop_ms(fz_ctx *ctx, float *out_ms) {
    FZ_REQUIRE(ctx && out_ms, "NULL argument");
    FZ_DEV(ctx);
    if (fz_capturing(ctx)) return fz_set_error(FZ_E_BADARG, "the timer cannot be read during graph capture");
    FZ_HIP(hipEventRecord(ctx->ev1, ctx->stream), "event record");
    FZ_HIP(hipEventSynchronize(ctx->ev1), "event synchronize");
    FZ_HIP(hipEventElapsedTime(out_ms, ctx->ev0, ctx->ev1), "event elapsed");
    return FZ_OK;
}

int fz_profile_begin(fz_ctx *ctx, int max_launches, int sample_every) {
    FZ_REQUIRE(ctx && max_launches > 0 && max_launches <= (1 << 20) && sample_every >= 1, "bad argument");
    FZ_DEV(ctx);
    if (fz_capturing(ctx)) return fz_set_error(FZ_E_BADARG, "per-dispatch profiling cannot start during graph capture");
    if (max_launches > ctx->prof_cap) {
        hipEvent_t *ev = (hipEvent_t *)realloc(ctx->prof_ev, sizeof(hipEvent_t) * 2 * (size_t)max_launches);
        unsigned char *kind = (unsigned char *)realloc(ctx->prof_kind, (size_t)max_launches);
        if (!ev || !kind) return fz_set_error(FZ_E_HIP, "out of host memory");
        ctx->prof_ev = ev;
        ctx->prof_kind = kind;
        for (int i = 2 * ctx->prof_cap; i < 2 * max_launches; ++i) FZ_HIP(hipEventCreate(&ctx->prof_ev[i]), "event create");
        ctx->prof_cap = max_launches;
    }
    ctx->prof_n = 0;
    ctx->prof_every = sample_every;
    ctx->prof_seen[0] = ctx->prof_seen[1] = 0;
    ctx->prof_on = 1;
    return FZ_OK;
}

int fz_profile_end(fz_ctx *ctx, double *fwd_avg_us, int *fwd_count, double *inv_avg_us, int *inv_count) {
    FZ_REQUIRE(ctx && fwd_avg_us && fwd_count && inv_avg_us && inv_count, "NULL argument");
    FZ_DEV(ctx);
    ctx->prof_on = 0;
    FZ_HIP(hipStreamSynchronize(ctx->stream), "profile sync");
    double sum[3] = {0, 0, 0};                  // kind 2 (multi-job launches) is reported by fz_profile_end_samples only
    int cnt[3] = {0, 0, 0};
    for (int i = 0; i < ctx->prof_n; ++i) {
        float ms = 0;
        FZ_HIP(hipEventElapsedTime(&ms, ctx->prof_ev[2 * i], ctx->prof_ev[2 * i + 1]), "event elapsed");
        sum[ctx->prof_kind[i]] += ms * 1e3;
        cnt[ctx->prof_kind[i]]++;
    }
    *fwd_avg_us = cnt[0] ? sum[0] / cnt[0] : 0.0;
    *inv_avg_us = cnt[1] ? sum[1] / cnt[1] : 0.0;
    *fwd_count = cnt[0];
    *inv_count = cnt[1];
    ctx->prof_n = 0;
    return FZ_OK;
}

int fz_profile_end_samples(fz_ctx *ctx, double *us, int *kind, int cap, int *n) {
    FZ_REQUIRE(ctx && us && kind && n && cap >= 0, "bad argument");
    FZ_DEV(ctx);
    ctx->prof_on = 0;
    FZ_HIP(hipStreamSynchronize(ctx->stream), "profile sync");
    int k = 0;
    for (int i = 0; i < ctx->prof_n && k < cap; ++i, ++k) {
        float ms = 0;
        FZ_HIP(hipEventElapsedTime(&ms, ctx->prof_ev[2 * i], ctx->prof_ev[2 * i + 1]), "event elapsed");
        us[k] = ms * 1e3;
        kind[k] = ctx->prof_kind[i];
    }
    *n = k;
    ctx->prof_n = 0;
    return FZ_OK;
}

// ---- transforms ----------------------------------------------------------------------------
int fz_ntt_forward(fz_ctx *ctx, const int32_t *d_in, int32_t *d_out, size_t batch) {
    FZ_REQUIRE(ctx && (batch == 0 || (d_in && d_out)), "NULL argument");
    FZ_DEV(ctx);
    return fz_launch_ntt(ctx, d_in, d_out, batch, false);
}

int fz_ntt_inverse(fz_ctx *ctx, const int32_t *d_in, int32_t *d_out, size_t batch) {
    FZ_REQUIRE(ctx && (batch == 0 || (d_in && d_out)), "NULL argument");
    FZ_DEV(ctx);
    return fz_launch_ntt(ctx, d_in, d_out, batch, true);
}

static int ntt_host(fz_ctx *ctx, int32_t *h_data, size_t batch, bool inverse) {
    FZ_REQUIRE(ctx && (batch == 0 || h_data), "NULL argument");
    FZ_DEV(ctx);
    if (batch == 0) return FZ_OK;
    const size_t bytes = batch * (size_t)ctx->degree * sizeof(int32_t);
    void *d = nullptr;
    FZ_TRY(fz_scratch(ctx, bytes, &d));
    FZ_TRY(fz_memcpy_h2d(ctx, d, h_data, bytes));
    FZ_TRY(fz_launch_ntt(ctx, (const int32_t *)d, (int32_t *)d, batch, inverse));
    return fz_memcpy_d2h(ctx, h_data, d, bytes);
}
int fz_ntt_forward_host(fz_ctx *ctx, int32_t *h_data, size_t batch) { return ntt_host(ctx, h_data, batch, false); }
int fz_ntt_inverse_host(fz_ctx *ctx, int32_t *h_data, size_t batch) { return ntt_host(ctx, h_data, batch, true); }

// ---- pointwise -------------------------------------------------------------------------------
int fz_pw_mul(fz_ctx *ctx, const int32_t *a, const int32_t *b, int32_t *out, size_t count) {
    FZ_REQUIRE(ctx && (count == 0 || (a && b && out)), "NULL argument");
    FZ_DEV(ctx);
    return fz_launch_pw(ctx, FZ_OP_MUL, a, b, out, count);
}
int fz_pw_add(fz_ctx *ctx, const int32_t *a, const int32_t *b, int32_t *out, size_t count) {
    FZ_REQUIRE(ctx && (count == 0 || (a && b && out)), "NULL argument");
    FZ_DEV(ctx);
    return fz_launch_pw(ctx, FZ_OP_ADD, a, b, out, count);
}
int fz_pw_sub(fz_ctx *ctx, const int32_t *a, const int32_t *b, int32_t *out, size_t count) {
    FZ_REQUIRE(ctx && (count == 0 || (a && b && out)), "NULL argument");
    FZ_DEV(ctx);
    return fz_launch_pw(ctx, FZ_OP_SUB, a, b, out, count);
}
int fz_pw_neg(fz_ctx *ctx, const int32_t *a, int32_t *out, size_t count) {
    FZ_REQUIRE(ctx && (count == 0 || (a && out)), "NULL argument");
    // the reference's __neg__ is -(x mod q) in [-(q-1), 0] (polynomials.py:155-163): the one value of the path that is NOT centred,
    // and for q >= 2^31 not an int32 either.  fz_pw_sub from a zero row gives the centred negation for every modulus.
    if (ctx->q >= 0x80000000u)
        return fz_set_error(FZ_E_UNSUPPORTED, "-(x mod q) does not fit int32 for q >= 2^31: use fz_pw_sub(0, x), the centred negation");
    FZ_DEV(ctx);
    return fz_launch_pw(ctx, FZ_OP_NEG, a, a, out, count);
}
int fz_pw_mulacc(fz_ctx *ctx, int32_t *acc, const int32_t *a, const int32_t *b, size_t count) {
    FZ_REQUIRE(ctx && (count == 0 || (acc && a && b)), "NULL argument");
    FZ_DEV(ctx);
    return fz_launch_pw(ctx, FZ_OP_MULACC, a, b, acc, count);
}
int fz_pw_mul_bcast(fz_ctx *ctx, const int32_t *a, const int32_t *s, int32_t *out, size_t rows) {
    FZ_REQUIRE(ctx && (rows == 0 || (a && s && out)), "NULL argument");
    FZ_DEV(ctx);
    return fz_launch_pw_bcast(ctx, a, s, out, rows);
}

int fz_pw_binary_host(fz_ctx *ctx, int op, const int32_t *h_a, const int32_t *h_b, int32_t *h_out, size_t count) {
    FZ_REQUIRE(ctx && op >= FZ_OP_MUL && op <= FZ_OP_NEG, "bad op %d", op);
    FZ_DEV(ctx);
    FZ_REQUIRE(count == 0 || (h_a && h_out && (op == FZ_OP_NEG || h_b)), "NULL argument");
    if (op == FZ_OP_NEG && ctx->q >= 0x80000000u)       // see fz_pw_neg
        return fz_set_error(FZ_E_UNSUPPORTED, "-(x mod q) does not fit int32 for q >= 2^31: use fz_pw_sub(0, x), the centred negation");
    if (count == 0) return FZ_OK;
    const size_t seg = (count * sizeof(int32_t) + 255) & ~(size_t)255;
    void *d = nullptr;
    FZ_TRY(fz_scratch(ctx, 3 * seg, &d));
    int32_t *da = (int32_t *)d, *db = (int32_t *)((char *)d + seg), *dout = (int32_t *)((char *)d + 2 * seg);
    FZ_TRY(fz_memcpy_h2d(ctx, da, h_a, count * sizeof(int32_t)));
    if (op != FZ_OP_NEG) FZ_TRY(fz_memcpy_h2d(ctx, db, h_b, count * sizeof(int32_t)));
    FZ_TRY(fz_launch_pw(ctx, op, da, op == FZ_OP_NEG ? da : db, dout, count));
    return fz_memcpy_d2h(ctx, h_out, dout, count * sizeof(int32_t));
}

// ---- synthetic batches ----------------------------------------------------------------------------
int fz_fill_synthetic(fz_ctx *ctx, int32_t *d_out, size_t count, uint64_t seed) {
    FZ_REQUIRE(ctx && (count == 0 || d_out), "NULL argument");
    FZ_DEV(ctx);
    return fz_launch_fill_synthetic(ctx, d_out, count, (unsigned long long)seed);
}

// ---- negacyclic product -------------------------------------------------------------------------
int fz_poly_mul(fz_ctx *ctx, const int32_t *d_f, const int32_t *d_g, int32_t *d_out, size_t batch) {
    FZ_REQUIRE(ctx && (batch == 0 || (d_f && d_g && d_out)), "NULL argument");
    FZ_DEV(ctx);
    if (ctx->logd < 0) return fz_set_error(FZ_E_UNSUPPORTED, "ring-only context (created with root 0) has no transforms");
    if (batch == 0) return FZ_OK;
    if ((((uintptr_t)d_f | (uintptr_t)d_g | (uintptr_t)d_out) & 3) != 0)
        return fz_set_error(FZ_E_BADARG, "buffers must be 4-byte aligned");
    if (!ctx->knob_unfused && (ctx->logd == 6 || ctx->logd == 8 || fz_polymul16_ok(ctx, d_f, d_g, d_out, batch)))
        return fz_launch_polymul_fused(ctx, d_f, d_g, d_out, batch);
    // generic degrees: NTT(f), NTT(g) into scratch, product in place, inverse into out
    const size_t n = batch * (size_t)ctx->degree, seg = (n * sizeof(int32_t) + 255) & ~(size_t)255;
    void *d = nullptr;
    FZ_TRY(fz_scratch(ctx, 2 * seg, &d));
    int32_t *fh = (int32_t *)d, *gh = (int32_t *)((char *)d + seg);
    FZ_TRY(fz_launch_ntt(ctx, d_f, fh, batch, false));
    FZ_TRY(fz_launch_ntt(ctx, d_g, gh, batch, false));
    FZ_TRY(fz_launch_pw(ctx, FZ_OP_MUL, fh, gh, fh, n));
    return fz_launch_ntt(ctx, fh, d_out, batch, true);
}

int fz_poly_mul_host(fz_ctx *ctx, const int32_t *h_f, const int32_t *h_g, int32_t *h_out, size_t batch) {
    FZ_REQUIRE(ctx && (batch == 0 || (h_f && h_g && h_out)), "NULL argument");
    FZ_DEV(ctx);
    if (ctx->logd < 0) return fz_set_error(FZ_E_UNSUPPORTED, "ring-only context (created with root 0) has no transforms");
    if (batch == 0) return FZ_OK;
    const size_t bytes = batch * (size_t)ctx->degree * sizeof(int32_t), seg = (bytes + 255) & ~(size_t)255;
    void *d = nullptr;
    FZ_TRY(fz_scratch2(ctx, 2 * seg, &d));            // scratch2: the generic path of fz_poly_mul owns the first scratch
    int32_t *df = (int32_t *)d, *dg = (int32_t *)((char *)d + seg);
    FZ_TRY(fz_memcpy_h2d(ctx, df, h_f, bytes));
    FZ_TRY(fz_memcpy_h2d(ctx, dg, h_g, bytes));
    FZ_TRY(fz_poly_mul(ctx, df, dg, df, batch));
    return fz_memcpy_d2h(ctx, h_out, df, bytes);
}

// ---- matrix-vector ------------------------------------------------------------------------------
int fz_matvec(fz_ctx *ctx, const int32_t *A, const int32_t *S, int32_t *out, size_t batch, int l) {
    FZ_REQUIRE(ctx && l >= 1 && (batch == 0 || (A && S && out)), "bad argument");
    FZ_DEV(ctx);
    return fz_launch_matvec(ctx, A, S, out, batch, l);
}

int fz_matvec_host(fz_ctx *ctx, const int32_t *h_A, const int32_t *h_S, int32_t *h_out, size_t batch, int l) {
    FZ_REQUIRE(ctx && l >= 1 && (batch == 0 || (h_A && h_S && h_out)), "bad argument");
    FZ_DEV(ctx);
    if (batch == 0) return FZ_OK;
    const size_t row = (size_t)ctx->degree * sizeof(int32_t);
    const size_t bA = (size_t)l * row, bS = batch * (size_t)l * row, bO = batch * row;
    const size_t oS = (bA + 255) & ~(size_t)255, oO = oS + ((bS + 255) & ~(size_t)255);
    void *d = nullptr;
    FZ_TRY(fz_scratch(ctx, oO + bO, &d));
    char *base = (char *)d;
    FZ_TRY(fz_memcpy_h2d(ctx, base, h_A, bA));
    FZ_TRY(fz_memcpy_h2d(ctx, base + oS, h_S, bS));
    FZ_TRY(fz_launch_matvec(ctx, (const int32_t *)base, (const int32_t *)(base + oS), (int32_t *)(base + oO), batch, l));
    return fz_memcpy_d2h(ctx, h_out, base + oO, bO);
}

// ---- fused scheme cores ---------------------------------------------------------------------------
int fz_keygen_core(fz_ctx *ctx, const int32_t *d_A, const int32_t *d_coef, int32_t *d_sk_hat, int32_t *d_vk,
                   size_t batch, int l) {
    FZ_REQUIRE(ctx && l >= 1 && (batch == 0 || (d_A && d_coef && d_sk_hat && d_vk)), "bad argument");
    FZ_DEV(ctx);
    // sk_hat = NTT(every secret row); vk_{L,R} = A . sk_hat_{L,R}   (fusion/fusion.py:363-370)
    if (batch == 0) return FZ_OK;
    if ((ctx->logd == 6 || ctx->logd == 8) && batch * 2 <= 0x7fffffffu && !ctx->knob_unfused &&
        ((((uintptr_t)d_A | (uintptr_t)d_coef | (uintptr_t)d_sk_hat) & 15) == 0))
        return fz_launch_keygen_fused(ctx, d_A, d_coef, d_sk_hat, d_vk, batch * 2, l);   // one launch, sk_hat not re-read
    FZ_TRY(fz_launch_ntt(ctx, d_coef, d_sk_hat, batch * 2 * (size_t)l, false));
    return fz_launch_matvec(ctx, d_A, d_sk_hat, d_vk, batch * 2, l);
}

int fz_keygen_core_bcast(fz_ctx *ctx, const int32_t *d_A, const int32_t *d_coef, int32_t *d_sk_hat, int32_t *d_vk,
                         size_t batch, int l) {
    FZ_REQUIRE(ctx && l >= 1 && (batch == 0 || (d_A && d_coef && d_sk_hat && d_vk)), "bad argument");
    FZ_DEV(ctx);
    if (batch == 0) return FZ_OK;
    if ((ctx->logd == 6 || ctx->logd == 8) && batch * 2 <= 0x7fffffffu && !ctx->knob_unfused &&
        ((((uintptr_t)d_A | (uintptr_t)d_coef | (uintptr_t)d_sk_hat) & 15) == 0))
        return fz_launch_keygen_fused(ctx, d_A, d_coef, d_sk_hat, d_vk, batch * 2, l, true);
    // generic degrees: expand the rows in sk_hat, transform in place, then the products
    FZ_TRY(fz_launch_bcast_rows(ctx, d_coef, d_sk_hat, batch * 2, l));
    FZ_TRY(fz_launch_ntt(ctx, d_sk_hat, d_sk_hat, batch * 2 * (size_t)l, false));
    return fz_launch_matvec(ctx, d_A, d_sk_hat, d_vk, batch * 2, l);
}

int fz_sign_core(fz_ctx *ctx, const int32_t *d_sk_hat, const int32_t *d_c_hat, int32_t *d_sig, size_t batch, int l) {
    FZ_REQUIRE(ctx && l >= 1 && (batch == 0 || (d_sk_hat && d_c_hat && d_sig)), "bad argument");
    FZ_DEV(ctx);
    return fz_launch_sign(ctx, d_sk_hat, d_c_hat, d_sig, batch, l);
}

int fz_aggregate_partial_batch(fz_ctx *ctx, const int32_t *d_sig, const int32_t *d_alpha_hat, int64_t *d_partial,
                               size_t partial_stride, size_t groups, size_t N, int l) {
    FZ_REQUIRE(ctx && l >= 1 && d_partial && (N == 0 || groups == 0 || (d_sig && d_alpha_hat)), "bad argument");
    FZ_DEV(ctx);
    FZ_REQUIRE(N < ((size_t)1 << 21), "N=%zu too large for exact int64/fp64 accumulation (< 2^21)", N);
    FZ_REQUIRE(groups <= 65535 && (groups <= 1 || partial_stride >= (size_t)l * ctx->degree), "bad groups / stride");
    return fz_launch_aggregate(ctx, d_sig, d_alpha_hat, d_partial, partial_stride, nullptr, groups, N, l);
}

int fz_aggregate_target_partial_batch(fz_ctx *ctx, const int32_t *d_sig, const int32_t *d_alpha_hat, const int32_t *d_vkL,
                                      const int32_t *d_vkR, const int32_t *d_c_hat, int64_t *d_partial, size_t partial_stride,
                                      int64_t *d_target_partial, size_t target_stride, size_t groups, size_t N, int l) {
    FZ_REQUIRE(ctx && l >= 1 && d_partial && d_target_partial, "bad argument");
    FZ_DEV(ctx);
    FZ_REQUIRE(N == 0 || groups == 0 || (d_sig && d_alpha_hat && d_vkL && d_vkR && d_c_hat), "NULL argument");
    FZ_REQUIRE(N < ((size_t)1 << 21), "N=%zu too large for exact int64/fp64 accumulation (< 2^21)", N);
    FZ_REQUIRE(groups <= 65535 && (groups <= 1 || (partial_stride >= (size_t)l * ctx->degree && target_stride >= (size_t)ctx->degree)),
               "bad groups / strides");
    FZ_REQUIRE((((uintptr_t)d_vkL | (uintptr_t)d_vkR | (uintptr_t)d_c_hat | (uintptr_t)d_alpha_hat | (uintptr_t)d_sig) & 15) == 0,
               "inputs must be 16-byte aligned");
    return fz_launch_aggregate(ctx, d_sig, d_alpha_hat, d_partial, partial_stride, nullptr, groups, N, l, d_vkL, d_vkR, d_c_hat,
                               d_target_partial, target_stride);
}

int fz_sign_aggregate_target_partial_batch(fz_ctx *ctx, const int32_t *d_sk_hat, const int32_t *d_c_hat, const int32_t *d_alpha_hat,
                                           const int32_t *d_vkL, const int32_t *d_vkR, int32_t *d_sig, int64_t *d_partial,
                                           size_t partial_stride, int64_t *d_target_partial, size_t target_stride, size_t groups,
                                           size_t N, int l) {
    FZ_REQUIRE(ctx && l >= 1 && d_partial, "bad argument");
    FZ_DEV(ctx);
    FZ_REQUIRE((d_vkL != nullptr) == (d_vkR != nullptr) && (d_vkL != nullptr) == (d_target_partial != nullptr),
               "verification keys and the target's partials come together or not at all");
    FZ_REQUIRE(N == 0 || groups == 0 || (d_sk_hat && d_c_hat && d_alpha_hat && d_sig), "NULL argument");
    FZ_REQUIRE(N < ((size_t)1 << 21), "N=%zu too large for exact int64/fp64 accumulation (< 2^21)", N);
    FZ_REQUIRE(groups <= 65535 && (groups <= 1 || (partial_stride >= (size_t)l * ctx->degree && (!d_vkL || target_stride >= (size_t)ctx->degree))),
               "bad groups / strides");
    if (N == 0 || groups == 0) return FZ_OK;
    const int d = ctx->degree;
    const uintptr_t align = (uintptr_t)d_sk_hat | (uintptr_t)d_c_hat | (uintptr_t)d_alpha_hat | (uintptr_t)d_vkL | (uintptr_t)d_vkR | (uintptr_t)d_sig;
    if (!ctx->knob_unfused && d % 4 == 0 && (d & (d - 1)) == 0 && d <= 256 && (align & 15) == 0)
        // one launch: sigma is written as it is computed and aggregated from registers (aggregate_onepass<.., SIGN>)
        return fz_launch_aggregate(ctx, nullptr, d_alpha_hat, d_partial, partial_stride, nullptr, groups, N, l, d_vkL, d_vkR, d_c_hat,
                                   d_target_partial, target_stride, nullptr, d_sk_hat, d_sig);
    // any other degree or alignment (and FZ_UNFUSED=1): the two launches this call stands for
    FZ_TRY(fz_launch_sign(ctx, d_sk_hat, d_c_hat, d_sig, groups * N, l));
    return fz_launch_aggregate(ctx, d_sig, d_alpha_hat, d_partial, partial_stride, nullptr, groups, N, l, d_vkL, d_vkR, d_c_hat,
                               d_target_partial, target_stride);
}

int fz_aggregate_partial(fz_ctx *ctx, const int32_t *d_sig, const int32_t *d_alpha_hat, int64_t *d_partial,
                         size_t N, int l) {
    return fz_aggregate_partial_batch(ctx, d_sig, d_alpha_hat, d_partial, 0, 1, N, l);
}

int fz_runtime_info(fz_ctx *ctx, int *out_build, int *out_runtime, char *out_arch, size_t arch_cap) {
    FZ_REQUIRE(ctx, "ctx is NULL");
    if (out_build) *out_build = HIP_VERSION;
    if (out_runtime) {
        int v = 0;
        FZ_HIP(hipRuntimeGetVersion(&v), "hipRuntimeGetVersion");
        *out_runtime = v;
    }
    if (out_arch && arch_cap) {
        hipDeviceProp_t prop;
        FZ_HIP(hipGetDeviceProperties(&prop, ctx->device), "hipGetDeviceProperties");
        snprintf(out_arch, arch_cap, "%s", prop.gcnArchName);
    }
    return FZ_OK;
}

// many aggregates of different sizes: chunks of kFzRaggedMax groups per launch
static int aggregate_ragged(fz_ctx *ctx, const int32_t *d_sig, const int32_t *d_alpha_hat, const int32_t *d_vkL, const int32_t *d_vkR,
                            const int32_t *d_c_hat, const size_t *h_offsets, size_t groups, int l, int32_t *d_out32,
                            int64_t *d_partial, size_t partial_stride, int64_t *d_target_partial, size_t target_stride) {
    const size_t per = (size_t)l * ctx->degree;
    for (size_t g = 0; g < groups; ++g) {
        FZ_REQUIRE(h_offsets[g] <= h_offsets[g + 1], "offsets must not decrease");
        FZ_REQUIRE(h_offsets[g + 1] - h_offsets[g] < ((size_t)1 << 21), "aggregate %zu: too many signers for exact accumulation (< 2^21)", g);
    }
    FZ_REQUIRE(h_offsets[groups] < 0xffffffffull, "too many signers in one call");
    for (size_t g0 = 0; g0 < groups; g0 += (size_t)kFzRaggedMax) {
        const size_t n = std::min<size_t>(kFzRaggedMax, groups - g0);
        size_t nmax = 0;
        for (size_t g = g0; g < g0 + n; ++g) nmax = std::max(nmax, h_offsets[g + 1] - h_offsets[g]);
        FZ_TRY(fz_launch_aggregate(ctx, d_sig, d_alpha_hat, d_partial ? d_partial + g0 * partial_stride : nullptr, partial_stride,
                                   d_out32 ? d_out32 + g0 * per : nullptr, n, nmax, l, d_vkL, d_vkR, d_c_hat,
                                   d_target_partial ? d_target_partial + g0 * target_stride : nullptr, target_stride, h_offsets + g0));
    }
    return FZ_OK;
}

int fz_aggregate_core_ragged(fz_ctx *ctx, const int32_t *d_sig, const int32_t *d_alpha_hat, const size_t *h_offsets, size_t groups,
                             int l, int32_t *d_out) {
    FZ_REQUIRE(ctx && l >= 1 && h_offsets && (groups == 0 || (d_sig && d_alpha_hat && d_out)), "bad argument");
    FZ_DEV(ctx);
    if (groups == 0) return FZ_OK;
    return aggregate_ragged(ctx, d_sig, d_alpha_hat, nullptr, nullptr, nullptr, h_offsets, groups, l, d_out, nullptr, 0, nullptr, 0);
}

int fz_aggregate_target_partial_ragged(fz_ctx *ctx, const int32_t *d_sig, const int32_t *d_alpha_hat, const int32_t *d_vkL,
                                       const int32_t *d_vkR, const int32_t *d_c_hat, const size_t *h_offsets, size_t groups, int l,
                                       int64_t *d_partial, size_t partial_stride, int64_t *d_target_partial, size_t target_stride) {
    FZ_REQUIRE(ctx && l >= 1 && h_offsets && d_target_partial && d_alpha_hat && d_vkL && d_vkR && d_c_hat, "bad argument");
    FZ_REQUIRE((d_sig == nullptr) == (d_partial == nullptr), "d_sig and d_partial go together (both NULL: verification targets only)");
    FZ_DEV(ctx);
    if (groups == 0) return FZ_OK;
    FZ_REQUIRE(groups == 1 || ((!d_partial || partial_stride >= (size_t)l * ctx->degree) && target_stride >= (size_t)ctx->degree),
               "bad strides");
    return aggregate_ragged(ctx, d_sig, d_alpha_hat, d_vkL, d_vkR, d_c_hat, h_offsets, groups, l, nullptr, d_partial, partial_stride,
                            d_target_partial, target_stride);
}

int fz_reduce_i64(fz_ctx *ctx, const int64_t *d_in, int32_t *d_out, size_t count) {
    FZ_REQUIRE(ctx && (count == 0 || (d_in && d_out)), "NULL argument");
    FZ_DEV(ctx);
    return fz_launch_reduce_i64(ctx, d_in, d_out, count);
}

int fz_aggregate_core(fz_ctx *ctx, const int32_t *d_sig, const int32_t *d_alpha_hat, int32_t *d_out, size_t N, int l) {
    FZ_REQUIRE(ctx && l >= 1 && N >= 1 && d_sig && d_alpha_hat && d_out, "bad argument");
    FZ_DEV(ctx);
    FZ_REQUIRE(N < ((size_t)1 << 21), "N=%zu too large for exact fp64 accumulation (< 2^21)", N);
    return fz_launch_aggregate(ctx, d_sig, d_alpha_hat, nullptr, 0, d_out, 1, N, l);
}

int fz_target_partial_batch(fz_ctx *ctx, const int32_t *d_vkL, const int32_t *d_vkR, const int32_t *d_c_hat,
                            const int32_t *d_alpha_hat, int64_t *d_partial, size_t partial_stride, size_t groups,
                            size_t N) {
    FZ_REQUIRE(ctx && d_partial && (N == 0 || groups == 0 || (d_vkL && d_vkR && d_c_hat && d_alpha_hat)), "bad argument");
    FZ_DEV(ctx);
    FZ_REQUIRE(N < ((size_t)1 << 21), "N=%zu too large (< 2^21)", N);
    FZ_REQUIRE(groups <= 65535 && (groups <= 1 || partial_stride >= (size_t)ctx->degree), "bad groups / stride");
    return fz_launch_target_partial(ctx, d_vkL, d_vkR, d_c_hat, d_alpha_hat, d_partial, partial_stride, groups, N);
}

int fz_target_partial(fz_ctx *ctx, const int32_t *d_vkL, const int32_t *d_vkR, const int32_t *d_c_hat,
                      const int32_t *d_alpha_hat, int64_t *d_partial, size_t N) {
    return fz_target_partial_batch(ctx, d_vkL, d_vkR, d_c_hat, d_alpha_hat, d_partial, 0, 1, N);
}

int fz_norm_weight(fz_ctx *ctx, const int32_t *d_coef, size_t batch, int64_t *d_max_abs, int32_t *d_weight) {
    FZ_REQUIRE(ctx && (batch == 0 || (d_coef && d_max_abs && d_weight)), "NULL argument");
    FZ_DEV(ctx);
    return fz_launch_norm_weight(ctx, d_coef, batch, d_max_abs, d_weight);
}

int fz_norm_weight_host(fz_ctx *ctx, const int32_t *h_coef, size_t batch, int64_t *h_max_abs, int32_t *h_weight) {
    FZ_REQUIRE(ctx && (batch == 0 || (h_coef && h_max_abs && h_weight)), "NULL argument");
    FZ_DEV(ctx);
    if (batch == 0) return FZ_OK;
    const size_t bC = batch * (size_t)ctx->degree * sizeof(int32_t);
    const size_t oM = (bC + 255) & ~(size_t)255, oW = oM + ((batch * sizeof(int64_t) + 255) & ~(size_t)255);
    void *d = nullptr;
    FZ_TRY(fz_scratch(ctx, oW + batch * sizeof(int32_t), &d));
    char *base = (char *)d;
    FZ_TRY(fz_memcpy_h2d(ctx, base, h_coef, bC));
    FZ_TRY(fz_launch_norm_weight(ctx, (const int32_t *)base, batch, (int64_t *)(base + oM), (int32_t *)(base + oW)));
    FZ_TRY(fz_memcpy_d2h(ctx, h_max_abs, base + oM, batch * sizeof(int64_t)));
    return fz_memcpy_d2h(ctx, h_weight, base + oW, batch * sizeof(int32_t));
}

int fz_verify_with_target_batch(fz_ctx *ctx, const int32_t *d_A, const int32_t *d_sig, const int32_t *d_target,
                                size_t groups, int l, int64_t beta_vf, int64_t omega_vf, int *h_verdicts) {
    FZ_REQUIRE(ctx && l >= 1 && groups >= 1 && groups <= 65535 && d_A && d_sig && d_target && h_verdicts, "bad argument");
    FZ_DEV(ctx);
    const size_t row = (size_t)ctx->degree * sizeof(int32_t), rows = groups * (size_t)l;
    // scratch: observed [G][D] i32 | coef [G][l][D] i32 | max_abs [G][l] i64 | weight [G][l] i32
    const size_t oC = (groups * row + 255) & ~(size_t)255;
    const size_t oM = oC + ((rows * row + 255) & ~(size_t)255);
    const size_t oW = oM + ((rows * sizeof(int64_t) + 255) & ~(size_t)255);
    void *d = nullptr;
    FZ_TRY(fz_scratch(ctx, oW + rows * sizeof(int32_t), &d));
    if (groups > ctx->verdict_cap) {
        FZ_HIP(hipStreamSynchronize(ctx->stream), "verdict sync");
        FZ_TRY(fz_retire(ctx, ctx->d_verdict, "verdict free"));
        ctx->d_verdict = nullptr;
        ctx->verdict_cap = 0;
        FZ_HIP(hipMalloc((void **)&ctx->d_verdict, groups * sizeof(int)), "verdict alloc");
        ctx->verdict_cap = groups;
    }
    if ((ctx->logd == 6 || ctx->logd == 8) && !ctx->knob_unfused) {
        // one launch: sigma read once (matvec + inverse transforms + norm/weight + verdict fused)
        FZ_TRY(fz_launch_verify_fused(ctx, d_A, d_sig, d_target, groups, l, beta_vf, omega_vf, ctx->d_verdict));
        return fz_memcpy_d2h(ctx, h_verdicts, ctx->d_verdict, groups * sizeof(int));
    }
    char *base = (char *)d;
    int32_t *observed = (int32_t *)base, *coef = (int32_t *)(base + oC);
    int64_t *mx = (int64_t *)(base + oM);
    int32_t *wt = (int32_t *)(base + oW);
    FZ_TRY(fz_launch_matvec(ctx, d_A, d_sig, observed, groups, l));               // fusion.py:715-717
    FZ_TRY(fz_launch_ntt(ctx, d_sig, coef, rows, true));                          // fusion.py:690-692
    FZ_TRY(fz_launch_norm_weight(ctx, coef, rows, mx, wt));                       // fusion.py:722-727
    FZ_TRY(fz_launch_verdict(ctx, d_target, observed, mx, wt, groups, l, beta_vf, omega_vf, ctx->d_verdict));
    return fz_memcpy_d2h(ctx, h_verdicts, ctx->d_verdict, groups * sizeof(int));
}

int fz_verify_with_target_batch_async(fz_ctx *ctx, const int32_t *d_A, const int32_t *d_sig, const int32_t *d_target,
                                      size_t groups, int l, int64_t beta_vf, int64_t omega_vf, int *d_verdicts) {
    FZ_REQUIRE(ctx && l >= 1 && groups >= 1 && groups <= 65535 && d_A && d_sig && d_target && d_verdicts, "bad argument");
    FZ_DEV(ctx);
    if (ctx->logd != 6 && ctx->logd != 8)
        return fz_set_error(FZ_E_UNSUPPORTED, "asynchronous verification needs the fused kernel (degree 64 or 256)");
    return fz_launch_verify_fused(ctx, d_A, d_sig, d_target, groups, l, beta_vf, omega_vf, d_verdicts);
}

int fz_verify_partials_batch_async(fz_ctx *ctx, const int32_t *d_A, const int64_t *d_partial, size_t partial_stride,
                                   const int64_t *d_target_partial, size_t target_stride, size_t groups, int l,
                                   int64_t beta_vf, int64_t omega_vf, int *d_verdicts) {
    FZ_REQUIRE(ctx && l >= 1 && groups >= 1 && groups <= 65535 && d_A && d_partial && d_target_partial && d_verdicts, "bad argument");
    FZ_DEV(ctx);
    FZ_REQUIRE(groups == 1 || (partial_stride >= (size_t)l * ctx->degree && target_stride >= (size_t)ctx->degree), "bad strides");
    FZ_REQUIRE((((uintptr_t)d_partial | (uintptr_t)d_target_partial | (uintptr_t)d_A) & 15) == 0 && (partial_stride & 1) == 0,
               "partials must be 16-byte aligned");
    if (ctx->logd != 6 && ctx->logd != 8)
        return fz_set_error(FZ_E_UNSUPPORTED, "verification from int64 partials needs the fused kernel (degree 64 or 256)");
    return fz_launch_verify_fused_i64(ctx, d_A, d_partial, partial_stride, d_target_partial, target_stride, groups, l, beta_vf,
                                      omega_vf, d_verdicts);
}

int fz_verify_with_target(fz_ctx *ctx, const int32_t *d_A, const int32_t *d_sig, const int32_t *d_target, int l,
                          int64_t beta_vf, int64_t omega_vf, int *h_verdict) {
    return fz_verify_with_target_batch(ctx, d_A, d_sig, d_target, 1, l, beta_vf, omega_vf, h_verdict);
}

int fz_verify_core(fz_ctx *ctx, const int32_t *d_A, const int32_t *d_sig, const int32_t *d_vkL, const int32_t *d_vkR,
                   const int32_t *d_c_hat, const int32_t *d_alpha_hat, size_t N, int l,
                   int64_t beta_vf, int64_t omega_vf, int *h_verdict) {
    FZ_REQUIRE(ctx && l >= 1 && N >= 1 && d_A && d_sig && d_vkL && d_vkR && d_c_hat && d_alpha_hat && h_verdict,
               "bad argument");
    FZ_DEV(ctx);
    // target lives behind the fz_verify_with_target scratch layout: allocate both up front
    const size_t row = (size_t)ctx->degree * sizeof(int32_t);
    const size_t inner = ((row + 255) & ~(size_t)255) + (((size_t)l * row + 255) & ~(size_t)255) +
                         (((size_t)l * sizeof(int64_t) + 255) & ~(size_t)255) + (((size_t)l * sizeof(int32_t) + 255) & ~(size_t)255);
    const size_t oP = inner, oT = oP + (((size_t)ctx->degree * sizeof(int64_t) + 255) & ~(size_t)255);
    void *d = nullptr;
    FZ_TRY(fz_scratch(ctx, oT + row, &d));
    char *base = (char *)d;
    int64_t *partial = (int64_t *)(base + oP);
    int32_t *target = (int32_t *)(base + oT);
    FZ_TRY(fz_target_partial(ctx, d_vkL, d_vkR, d_c_hat, d_alpha_hat, partial, N));   // fusion.py:706-714
    FZ_TRY(fz_launch_reduce_i64(ctx, partial, target, (size_t)ctx->degree));
    return fz_verify_with_target(ctx, d_A, d_sig, target, l, beta_vf, omega_vf, h_verdict);
}


// ---- many independent transforms in one dispatch -------------------------------------------------------
int fz_ntt_multi(fz_ctx *ctx, const fz_ntt_job *h_jobs, size_t n_jobs) {
    FZ_REQUIRE(ctx && (n_jobs == 0 || h_jobs), "NULL argument");
    FZ_DEV(ctx);
    if (ctx->logd < 0) return fz_set_error(FZ_E_UNSUPPORTED, "ring-only context (created with root 0) has no transforms");
    for (size_t j = 0; j < n_jobs; ++j) {
        const fz_ntt_job &jb = h_jobs[j];
        FZ_REQUIRE(jb.rows == 0 || (jb.d_in && jb.d_out), "job %zu: NULL buffer", j);
        FZ_REQUIRE(jb.rows < ((size_t)1 << 31), "job %zu: too many rows", j);
        if ((((uintptr_t)jb.d_in | (uintptr_t)jb.d_out) & 15) != 0 && ctx->logd >= 2)
            return fz_set_error(FZ_E_BADARG, "job %zu: transform buffers must be 16-byte aligned", j);
    }
    if (ctx->logd != 6 && ctx->logd != 8) {              // other degrees: one launch per job (same results)
        for (size_t j = 0; j < n_jobs; ++j)
            FZ_TRY(fz_launch_ntt(ctx, h_jobs[j].d_in, h_jobs[j].d_out, h_jobs[j].rows, h_jobs[j].inverse != 0));
        return FZ_OK;
    }
    const unsigned ppw = 64u / (unsigned)(ctx->degree / 4);
    FzMultiJobs J;
    memset(&J, 0, sizeof(J));
    unsigned long long total = 0;
    for (size_t j = 0; j < n_jobs; ++j) {
        const fz_ntt_job &jb = h_jobs[j];
        if (jb.rows == 0) continue;
        const unsigned long long tasks = (jb.rows + ppw - 1) / ppw;
        if (J.n == kFzMultiMax || total + tasks > 0x7fffffffull) {      // table full: flush
            FZ_TRY(fz_launch_ntt_multi(ctx, J));
            memset(&J, 0, sizeof(J));
            total = 0;
        }
        total += tasks;
        J.in[J.n] = jb.d_in;
        J.out[J.n] = jb.d_out;
        J.rows[J.n] = (unsigned)jb.rows | (jb.inverse ? 0x80000000u : 0u);
        ++J.n;
    }
    return fz_launch_ntt_multi(ctx, J);
}

// the next staging slot, at least `bytes` long and no longer read by any launch (fz_internal.h)
static int challenge_stage(fz_ctx *ctx, size_t bytes, fz_ctx::FzStage **out) {
    fz_ctx::FzStage &st = ctx->chal_stage[ctx->chal_stage_next];
    ctx->chal_stage_next ^= 1;
    if (!st.ev) FZ_HIP(hipEventCreateWithFlags(&st.ev, hipEventDisableTiming), "staging event");
    if (st.busy) { FZ_HIP(hipEventSynchronize(st.ev), "staging slot still in use"); st.busy = 0; }
    if (st.bytes < bytes) {
        if (st.h) { FZ_HIP(hipHostFree(st.h), "staging free"); st.h = nullptr; st.bytes = 0; }
        const size_t want = std::max<size_t>((bytes + 65535) & ~(size_t)65535, 256 << 10);
        FZ_HIP(hipHostMalloc((void **)&st.h, want, hipHostMallocDefault), "staging alloc");
        st.bytes = want;
    }
    *out = &st;
    return FZ_OK;
}

// ---- the challenge pipeline on the device (device half) ---------------------------------------------
// the pre-hashed messages come either as h_prehash [N][32] (computed by the caller, fz_hash_messages) or are computed here,
// on the device, from the messages themselves (h_msgs back to back, h_msg_off [N + 1]); h_prehash_out (optional, with
// messages only) receives them
static int challenge_dev(fz_ctx *ctx, const fz_scheme_params *P, const int32_t *d_vk, const uint8_t *h_prehash, const char *h_msgs,
                         const size_t *h_msg_off, uint8_t *h_prehash_out, size_t N, int32_t *d_out, bool transform) {
    FZ_REQUIRE(ctx && P && (N == 0 || (d_vk && (h_prehash || h_msg_off) && d_out)), "NULL argument");
    if (!h_prehash && N) {
        FZ_REQUIRE(h_msg_off[N] == h_msg_off[0] || h_msgs, "NULL argument");
        for (size_t i = 0; i < N; ++i) FZ_REQUIRE(h_msg_off[i] <= h_msg_off[i + 1], "message offsets must not decrease");
    }
    FZ_DEV(ctx);
    if (!fz_host_params_ok(P)) return fz_set_error(FZ_E_BADARG, "bad scheme parameters");
    if (P->degree != ctx->degree || P->modulus != (int64_t)ctx->q)
        return fz_set_error(FZ_E_BADARG, "scheme parameters (degree %d) do not belong to this context (degree %d)", P->degree, ctx->degree);
    if (fz_capturing(ctx)) return fz_set_error(FZ_E_BADARG, "the challenge pipeline uploads the pre-hashed messages: not during graph capture");
    const long long bound = std::max<long long>(1, std::min<long long>((long long)P->modulus / 2, P->beta_ch));
    if (bound != 1 || P->degree > 256 || P->degree < 4 || (P->degree & (P->degree - 1)) || P->omega_ch > P->degree ||
        (((uintptr_t)d_vk | (uintptr_t)d_out) & 15))
        return fz_set_error(FZ_E_UNSUPPORTED, "device challenge pipeline: ternary challenges (norm bound 1), degree 4..256 and "
                                              "16-byte aligned rows only; use fz_challenge_coefficients (host)");
    if (transform && ctx->logd < 0) return fz_set_error(FZ_E_UNSUPPORTED, "ring-only context (created with root 0) has no transforms");
    if (N == 0) return FZ_OK;
    int sb = 0, cb = 0, ib = 0;
    const size_t needed = fz_host_challenge_needed_bytes(P, &sb, &cb, &ib);
    const int out_blocks = (int)((needed + 135) / 136);
    // longest possible text: fixed pieces + 2*degree values of up to 11 characters + separators + 78 digits
    char t0[384], t1[384], t2[16];
    int n0, n1, n2;
    fz_host_vk_text_parts(P, t0, &n0, t1, &n1, t2, &n2, 384);
    if (n0 < 0) return fz_set_error(FZ_E_UNSUPPORTED, "verification-key text pieces do not fit");
    const size_t max_len = (size_t)n0 + n1 + n2 + (size_t)2 * P->degree * 13 + 78;
    size_t blocks = max_len / 136 + 1;
    blocks += blocks & 1;                                   // even: rows stay 16-byte aligned
    const size_t text_stride = blocks * 136;
    if (!ctx->d_chal_tab || ctx->chal_tab_ib != ib || ctx->chal_tab_degree != P->degree) {
        std::vector<uint32_t> tab((size_t)(P->degree + 1) * 16);
        fz_challenge_weight_table(ib, P->degree, tab.data());
        if (ctx->d_chal_tab) { FZ_HIP(hipStreamSynchronize(ctx->stream), "table sync"); FZ_TRY(fz_retire(ctx, ctx->d_chal_tab, "table free")); ctx->d_chal_tab = nullptr; }
        FZ_HIP(hipMalloc((void **)&ctx->d_chal_tab, tab.size() * 4), "table alloc");
        FZ_HIP(hipMemcpy(ctx->d_chal_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice), "table upload");
        ctx->chal_tab_ib = ib;
        ctx->chal_tab_degree = P->degree;
    }
    // Which form.  Up to kWaveFormMax signers per call every signer gets a WAVE (fz_launch_challenge_wave: the chain of ~108
    // permutations at 24 instructions + 4 gathers per round, text and stream in LDS); beyond, the three-kernel pipeline with 32
    // or 64 signers per wave has the higher throughput (profiles/r06_challenge_pipeline.txt).
    constexpr size_t kWaveFormMax = 4608;                    // 4096 signers: 0.67 ms against 0.77; 5120: 0.85 against 0.79
    const bool wave_form = fz_challenge_wave_ok(P) && (ctx->knob_shake_full == 3 || (ctx->knob_shake_full == 0 && N <= kWaveFormMax));
    if (wave_form) {
        // One launch, nothing uploaded: the kernel reads the messages (or digests) from pinned host memory and leaves the
        // digests there.  The call returns without synchronising unless the caller wants the digests.
        const size_t msg_bytes = h_prehash ? 0 : h_msg_off[N] - h_msg_off[0];
        const size_t o_pre = 0, o_off = N * 32, o_msg = o_off + (N + 1) * 8;
        fz_ctx::FzStage *st = nullptr;
        FZ_TRY(challenge_stage(ctx, o_msg + msg_bytes + 64, &st));
        if (h_prehash) {
            memcpy(st->h + o_pre, h_prehash, N * 32);
        } else {
            static_assert(sizeof(size_t) == sizeof(unsigned long long), "offsets travel as 64-bit words");
            memcpy(st->h + o_off, h_msg_off, (N + 1) * 8);
            if (msg_bytes) memcpy(st->h + o_msg, h_msgs + h_msg_off[0], msg_bytes);
        }
        int rc = fz_launch_challenge_wave(ctx, P, d_vk, h_prehash ? st->h + o_pre : nullptr, st->h + o_msg, (const unsigned long long *)(st->h + o_off),
                                          (!h_prehash && h_prehash_out) ? st->h + o_pre : nullptr, N, text_stride, out_blocks, ctx->d_chal_tab, d_out);
        if (rc == FZ_OK && transform) rc = fz_launch_ntt(ctx, d_out, d_out, N, false);
        FZ_HIP(hipEventRecord(st->ev, ctx->stream), "staging event record");
        st->busy = 1;
        if (rc != FZ_OK) return rc;
        if (!h_prehash && h_prehash_out) {
            FZ_HIP(hipStreamSynchronize(ctx->stream), "digest sync");
            memcpy(h_prehash_out, st->h + o_pre, N * 32);
        }
        return FZ_OK;
    }
    // signers per pass: 65536 = two waves of 32 signers on each of the chip's 1024 SIMDs (a second wave per SIMD fills the
    // issue slots one wave alone leaves empty); bounds the scratch at ~16 KB per signer
    const size_t chunk = 65536;
    for (size_t base = 0; base < N; base += chunk) {
        const size_t n = std::min(chunk, N - base);
        const size_t xstride = (n + 63) & ~(size_t)63;
        const size_t o_pre = 0, o_nb = (n * 32 + 255) & ~(size_t)255, o_text = o_nb + ((n * 4 + 255) & ~(size_t)255);
        const size_t o_xof = o_text + ((n * text_stride + 255) & ~(size_t)255);
        const size_t o_off = o_xof + ((((size_t)out_blocks * 34 + 1) * xstride * 4 + 255) & ~(size_t)255);    // + one spare word row (decoder)
        const size_t msg_bytes = h_prehash ? 0 : h_msg_off[base + n] - h_msg_off[base];
        const size_t o_dec = o_off + (((n + 1) * 8 + 255) & ~(size_t)255);              // [n][16] words: the integers in base 10^9
        const size_t o_msg = o_dec + n * 64;
        const size_t total = h_prehash ? o_off : o_msg + ((msg_bytes + 255) & ~(size_t)255);
        void *scr = nullptr;
        FZ_TRY(fz_scratch(ctx, total, &scr));
        uint8_t *sp = (uint8_t *)scr;
        if (h_prehash) {
            FZ_HIP(hipMemcpyAsync(sp + o_pre, h_prehash + 32 * base, n * 32, hipMemcpyHostToDevice, ctx->stream), "upload of the pre-hashed messages");
        } else {
            static_assert(sizeof(size_t) == sizeof(unsigned long long), "offsets travel as 64-bit words");
            FZ_HIP(hipMemcpyAsync(sp + o_off, h_msg_off + base, (n + 1) * 8, hipMemcpyHostToDevice, ctx->stream), "upload of the message offsets");
            if (msg_bytes)
                FZ_HIP(hipMemcpyAsync(sp + o_msg, h_msgs + h_msg_off[base], msg_bytes, hipMemcpyHostToDevice, ctx->stream), "upload of the messages");
            FZ_TRY(fz_launch_prehash(ctx, P, sp + o_msg, (const unsigned long long *)(sp + o_off), n, sp + o_pre, (uint32_t *)(sp + o_dec)));
            if (h_prehash_out)
                FZ_HIP(hipMemcpyAsync(h_prehash_out + 32 * base, sp + o_pre, n * 32, hipMemcpyDeviceToHost, ctx->stream), "download of the pre-hashed messages");
        }
        FZ_HIP(hipStreamSynchronize(ctx->stream), "upload sync");      // the caller's buffers have been consumed when this returns
        FZ_TRY(fz_launch_challenge(ctx, P, d_vk + base * 2 * (size_t)P->degree, sp + o_pre,
                                   h_prehash ? nullptr : (const uint32_t *)(sp + o_dec), n, sp + o_text, text_stride,
                                   (int *)(sp + o_nb), (uint32_t *)(sp + o_xof), xstride, out_blocks, ctx->d_chal_tab,
                                   d_out + base * (size_t)P->degree));
    }
    if (transform) return fz_launch_ntt(ctx, d_out, d_out, N, false);
    return FZ_OK;
}

int fz_challenge_coefficients_dev(fz_ctx *ctx, const fz_scheme_params *P, const int32_t *d_vk, const uint8_t *h_prehash,
                                  size_t N, int32_t *d_coefs) {
    FZ_REQUIRE(h_prehash || N == 0, "NULL argument");
    return challenge_dev(ctx, P, d_vk, h_prehash, nullptr, nullptr, nullptr, N, d_coefs, false);
}

int fz_challenge_hat_dev(fz_ctx *ctx, const fz_scheme_params *P, const int32_t *d_vk, const uint8_t *h_prehash, size_t N,
                         int32_t *d_c_hat) {
    FZ_REQUIRE(h_prehash || N == 0, "NULL argument");
    return challenge_dev(ctx, P, d_vk, h_prehash, nullptr, nullptr, nullptr, N, d_c_hat, true);
}

int fz_challenge_hat_msgs_dev(fz_ctx *ctx, const fz_scheme_params *P, const int32_t *d_vk, const char *h_msgs, const size_t *h_msg_off,
                              size_t N, int32_t *d_c_hat, uint8_t *h_prehash_out) {
    FZ_REQUIRE(h_msg_off || N == 0, "NULL argument");
    return challenge_dev(ctx, P, d_vk, nullptr, h_msgs, h_msg_off, h_prehash_out, N, d_c_hat, true);
}

// ---- the reference's seeded secret-key sampler on the device (fz_sample.hip) --------------------------------------------
int fz_sample_secret_polys_dev(fz_ctx *ctx, const uint64_t *h_seeds, size_t N, int64_t modulus, int degree, int64_t norm_bound,
                               int64_t weight_bound, int32_t *d_out) {
    FZ_REQUIRE(ctx && (N == 0 || (h_seeds && d_out)), "NULL argument");
    FZ_REQUIRE(degree >= 1 && modulus >= 2, "bad degree / modulus");
    FZ_DEV(ctx);
    if (fz_capturing(ctx)) return fz_set_error(FZ_E_BADARG, "the sampler uploads the seeds: not during graph capture");
    const int64_t bound = std::max<int64_t>(0, std::min<int64_t>(modulus / 2, norm_bound));
    if (bound < 1 || bound >= (1ll << 32)) return fz_set_error(FZ_E_BADARG, "empty range for randrange()");
    if (weight_bound < degree)
        return fz_set_error(FZ_E_UNSUPPORTED, "device sampler: weight bound = degree only (no shuffle); use fz_sample_secret_polys");
    // the right half of a key is seeded with seed + 1: at seed = 2^64 - 1 that is 2^64, a THREE-word key for CPython's
    // init_by_array, not the wrapped 0 -- outside what this entry (and the host clone) takes
    for (size_t i = 0; i < N; ++i)
        if (h_seeds[i] == UINT64_MAX)
            return fz_set_error(FZ_E_UNSUPPORTED, "seed %zu is 2^64 - 1: seed + 1 needs a wider key than this sampler takes", i);
    if (N == 0) return FZ_OK;
    if (!ctx->d_mt_init) {
        uint32_t tab[624];
        fz_mt_init_table(tab);
        FZ_HIP(hipMalloc((void **)&ctx->d_mt_init, sizeof(tab)), "sampler table alloc");
        FZ_HIP(hipMemcpy(ctx->d_mt_init, tab, sizeof(tab), hipMemcpyHostToDevice), "sampler table upload");
    }
    int kbits = 0;
    for (uint64_t t = (uint64_t)bound; t; t >>= 1) ++kbits;            // bound.bit_length()
    // Up to 4096 keys (8192 generators: a seeding wave on every other CU): two kernels -- seeding on a lane per generator,
    // everything after it on a wave per generator, the states (20 MiB at most) through scratch: 54 + 29 us per 1024 keys
    // against 190 in one kernel.  Beyond, the one lane-per-polynomial kernel already has a wave on every CU and the two
    // forms take the same time (16 384 keys: 0.47 ms one kernel, 0.56 ms in four chunks of two).
    const bool two_kernels = bound < (1ll << 31) && N <= 4096;
    const size_t state_bytes = two_kernels ? N * 2 * 624 * sizeof(uint32_t) : 0;
    void *scr = nullptr;
    FZ_TRY(fz_scratch(ctx, state_bytes + 256, &scr));
    uint32_t *d_state = two_kernels ? (uint32_t *)scr : nullptr;
    // the seeds and the "ran out of output" flag live in the pinned staging slot the challenge pipeline uses: the kernels read the
    // seeds in place and store the flag there, so the call is two launches and one synchronisation -- no memset, no pageable upload,
    // no download (0.098 -> 0.078 ms per 1024 keys: most of what was left beside the two kernels' 57 us)
    fz_ctx::FzStage *st = nullptr;
    FZ_TRY(challenge_stage(ctx, 64 + N * 8, &st));
    volatile int *h_fail = reinterpret_cast<volatile int *>(st->h);
    *h_fail = 0;
    memcpy(st->h + 64, h_seeds, N * 8);
    int rc = fz_launch_mt_sample(ctx, reinterpret_cast<const unsigned long long *>(st->h + 64), N, degree, (uint32_t)bound, kbits,
                                 ctx->d_mt_init, d_out, reinterpret_cast<int *>(st->h), d_state);
    FZ_HIP(hipEventRecord(st->ev, ctx->stream), "staging event record");
    st->busy = 1;
    if (rc != FZ_OK) return rc;
    FZ_HIP(hipStreamSynchronize(ctx->stream), "sampler sync");
    const int fail = *h_fail;
    if (fail) return fz_set_error(FZ_E_UNSUPPORTED, "device sampler ran out of generator output for a seed; use fz_sample_secret_polys");
    return FZ_OK;
}

// ---- launch-floor diagnostics ------------------------------------------------------------------------------
int fz_diag_empty_launch(fz_ctx *ctx) {
    FZ_REQUIRE(ctx, "ctx is NULL");
    FZ_DEV(ctx);
    return fz_launch_diag(ctx, 0, nullptr, nullptr, 0);
}

int fz_diag_copy(fz_ctx *ctx, const void *d_src, void *d_dst, size_t bytes) {
    FZ_REQUIRE(ctx && (bytes == 0 || (d_src && d_dst)), "NULL argument");
    FZ_REQUIRE((((uintptr_t)d_src | (uintptr_t)d_dst) & 15) == 0 && bytes % 16 == 0, "16-byte aligned buffers and size");
    FZ_DEV(ctx);
    return fz_launch_diag(ctx, 1, d_src, d_dst, bytes);
}

int fz_diag_ntt_schedule(fz_ctx *ctx, size_t rows, int *family) {
    FZ_REQUIRE(ctx && family, "NULL argument");
    *family = 0;
    if (ctx->logd < 5 || ctx->logd > 8) return FZ_OK;
    const bool radix4_exists = ctx->logd == 6 || ctx->logd == 8;           // (degrees 32 and 128 only have the 16-per-lane kernels)
    if (!radix4_exists || ctx->force_kernel == 16) *family = 16;
    else if (ctx->force_kernel == 4) *family = 4;
    else *family = rows >= (size_t)ctx->small_batch_rows ? 16 : 4;
    return FZ_OK;
}

int fz_diag_shader_clock(fz_ctx *ctx, unsigned microseconds, double *out_mhz) {
    FZ_REQUIRE(ctx && out_mhz, "NULL argument");
    FZ_REQUIRE(microseconds >= 1 && microseconds <= 1000000, "between 1 us and 1 s");
    FZ_DEV(ctx);
    // its own stream: the probe runs BESIDE whatever the caller queued on the context's stream (that is the point).  Stream
    // and result word live as long as the context: hipMalloc / hipFree here would synchronise the device with that work.
    if (!ctx->diag_stream) {
        FZ_HIP(hipStreamCreateWithFlags(&ctx->diag_stream, hipStreamNonBlocking), "diag stream");
        FZ_HIP(hipMalloc((void **)&ctx->d_diag, 2 * sizeof(unsigned long long)), "diag alloc");
    }
    unsigned long long h[2] = {0, 0};
    FZ_TRY(fz_launch_diag_clock(ctx->diag_stream, (unsigned long long)microseconds * 100ull, ctx->d_diag));
    FZ_HIP(hipMemcpyAsync(h, ctx->d_diag, sizeof(h), hipMemcpyDeviceToHost, ctx->diag_stream), "diag read");
    FZ_HIP(hipStreamSynchronize(ctx->diag_stream), "diag sync");
    *out_mhz = h[1] ? 100.0 * (double)h[0] / (double)h[1] : 0.0;
    return FZ_OK;
}

int fz_diag_delay(fz_ctx *ctx, unsigned microseconds) {
    FZ_REQUIRE(ctx, "ctx is NULL");
    FZ_REQUIRE(microseconds >= 1 && microseconds <= 100000, "between 1 us and 100 ms");
    FZ_DEV(ctx);
    return fz_launch_diag_clock(ctx->stream, (unsigned long long)microseconds * 100ull, nullptr);
}

// ---- device-side launch timestamps of the multi-job transform (diagnostics: include/fusion_hip_diag.h) -----------------
// rocprofv3 --kernel-trace serialises the dispatches of all streams, and HIP events are host-visible markers between
// dispatches: neither shows WHEN launches of different streams ran relative to each other.  While stamps are on, every
// workgroup of every fz_ntt_multi launch of this context stores the 100 MHz reference counter (s_memrealtime: one counter
// for the whole chip) at entry and -- after its stores have been acknowledged -- at exit; launch k of the recording is the
// interval [min entry, max exit] over its workgroups.  Slots are assigned when a launch is ISSUED (or captured: the slot is
// part of the recorded kernel arguments), so a captured graph is replayed ONCE between fz_diag_stamps_reset and
// fz_diag_stamps_read.
int fz_diag_stamps_begin(fz_ctx *ctx, size_t max_launches, size_t max_workgroups) {
    FZ_REQUIRE(ctx && max_launches >= 1 && max_launches <= (1u << 20) && max_workgroups >= 1 && max_workgroups <= ((size_t)1 << 26), "bad argument");
    FZ_DEV(ctx);
    if (fz_capturing(ctx)) return fz_set_error(FZ_E_BADARG, "stamps cannot be set up during graph capture");
    // a recording that is still on is switched OFF first: if anything below fails, no launch takes a slot in a buffer that is gone
    ctx->stamp_on = 0;
    ctx->stamp_n = 0;
    ctx->stamp_used = 0;
    ctx->stamp_launch_cap = 0;
    ctx->stamp_wg_cap = 0;
    if (ctx->d_stamp) { void *old = ctx->d_stamp; ctx->d_stamp = nullptr; FZ_TRY(fz_retire(ctx, old, "stamp buffer")); }
    free(ctx->stamp_first); free(ctx->stamp_count);
    ctx->stamp_first = (size_t *)malloc(sizeof(size_t) * max_launches);
    ctx->stamp_count = (unsigned *)malloc(sizeof(unsigned) * max_launches);
    if (!ctx->stamp_first || !ctx->stamp_count) {
        free(ctx->stamp_first); free(ctx->stamp_count);
        ctx->stamp_first = nullptr; ctx->stamp_count = nullptr;
        return fz_set_error(FZ_E_HIP, "out of host memory");
    }
    FZ_HIP(hipMalloc((void **)&ctx->d_stamp, 16 * max_workgroups), "stamp buffer");
    FZ_HIP(hipMemsetAsync(ctx->d_stamp, 0, 16 * max_workgroups, ctx->stream), "stamp reset");
    ctx->stamp_launch_cap = (int)max_launches;
    ctx->stamp_wg_cap = max_workgroups;
    ctx->stamp_n = 0;
    ctx->stamp_used = 0;
    ctx->stamp_on = 1;
    return FZ_OK;
}

// stop assigning slots (launches issued from now on carry no stamps); the recording stays readable
int fz_diag_stamps_stop(fz_ctx *ctx) {
    FZ_REQUIRE(ctx, "ctx is NULL");
    ctx->stamp_on = 0;
    return FZ_OK;
}

// zero every slot (asynchronous on the context's stream): before the ONE replay of a captured recording that is to be read
int fz_diag_stamps_reset(fz_ctx *ctx) {
    FZ_REQUIRE(ctx && ctx->d_stamp, "no stamp buffer");
    FZ_DEV(ctx);
    FZ_HIP(hipMemsetAsync(ctx->d_stamp, 0, 16 * ctx->stamp_used, ctx->stream), "stamp reset");
    return FZ_OK;
}

// synchronises the context's stream; per recorded launch k < *n: h_start[k] / h_end[k] = min entry / max exit over its
// workgroups (ticks of the 100 MHz counter; 0 / 0 when no workgroup of the launch has run since the reset), h_workgroups[k]
// (optional) = how many of its workgroups stamped, h_last_start[k] (optional) = the latest entry (when the dispatcher had
// handed out the launch's last workgroup)
int fz_diag_stamps_read(fz_ctx *ctx, uint64_t *h_start, uint64_t *h_end, uint64_t *h_last_start, uint32_t *h_workgroups, size_t cap, size_t *n) {
    FZ_REQUIRE(ctx && h_start && h_end && n, "NULL argument");
    FZ_REQUIRE(ctx->d_stamp, "no stamp buffer");
    FZ_DEV(ctx);
    if (fz_capturing(ctx)) return fz_set_error(FZ_E_BADARG, "stamps cannot be read during graph capture");
    FZ_HIP(hipStreamSynchronize(ctx->stream), "stamp sync");
    unsigned long long *h = (unsigned long long *)malloc(16 * (ctx->stamp_used ? ctx->stamp_used : 1));      // (no C++ exception crosses the C ABI)
    if (!h) return fz_set_error(FZ_E_HIP, "out of host memory");
    if (ctx->stamp_used) {
        const int rc = fz_check_hip(hipMemcpy(h, ctx->d_stamp, 16 * ctx->stamp_used, hipMemcpyDeviceToHost), "stamp read");
        if (rc != FZ_OK) { free(h); return rc; }
    }
    size_t k = 0;
    for (; k < (size_t)ctx->stamp_n && k < cap; ++k) {
        unsigned long long lo = ~0ull, hi = 0, last = 0;
        unsigned seen = 0;
        for (size_t w = ctx->stamp_first[k]; w < ctx->stamp_first[k] + ctx->stamp_count[k]; ++w) {
            const unsigned long long a = h[2 * w], b = h[2 * w + 1];
            if (!a && !b) continue;
            ++seen;
            lo = std::min(lo, a); hi = std::max(hi, b); last = std::max(last, a);
        }
        h_start[k] = seen ? lo : 0;
        h_end[k] = hi;
        if (h_last_start) h_last_start[k] = last;
        if (h_workgroups) h_workgroups[k] = seen;
    }
    *n = k;
    free(h);
    return FZ_OK;
}

// ---- the one exchange step of the path, in the C ABI: RCCL all-reduce of the int64 partial sums ---------------------
// RCCL is bound lazily (dlopen): the library loads and every other entry point works on a machine without it.
// WHICH copy is bound is an ownership question (VERDICT / ADVICE r04: an abort at process exit): a torch wheel ships its
// own librccl.so under the soname librccl.so.1 but asks for it by the unversioned file name, so a process that bound
// /opt/rocm's copy by soname first and imported torch afterwards carried TWO RCCLs of different ROCm releases -- and with
// RTLD_GLOBAL (rounds 2-4) the first copy's nccl* symbols interposed on the second's callers.  The rule now:
//   1. a copy already mapped under the soname serves (RTLD_NOLOAD): whoever loaded it owns it, we share it;
//   2. otherwise the copy that ships BESIDE THE HIP RUNTIME THIS PROCESS RUNS ON (dladdr of a HIP entry point: torch/lib when
//      fusion_hip mapped the wheel's runtime first, /opt/rocm/lib otherwise), opened by path -- a later `import torch` then finds
//      the same file (same device / inode) already mapped instead of adding a second copy;
//   3. otherwise the soname on the default search path.
// Always RTLD_LOCAL (nothing of RCCL enters the global scope), never dlclose'd (RCCL's own teardown runs at exit, after ours).
namespace {
struct RcclApi {
    void *handle;
    int (*GetUniqueId)(void *);
    int (*CommInitRank)(void **, int, fz_unique_id, int);
    int (*CommDestroy)(void *);
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t);
    const char *(*GetErrorString)(int);
    int (*CommCount)(void *, int *);
    int (*GetVersion)(int *);
    int (*Broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t);
    int (*ReduceScatter)(const void *, void *, size_t, int, int, void *, hipStream_t);
    char path[512];      // the file the symbols came from
    char how[64];        // which of the three rules found it
};
RcclApi g_rccl = {};
std::mutex g_rccl_mu;                       // binding and the registry of live communicators
std::vector<fz_comm *> g_live_comms;

void *rccl_open(char *how, size_t how_cap) {
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    if (h) { snprintf(how, how_cap, "already mapped (shared)"); return h; }
    Dl_info di;
    if (dladdr((const void *)&hipGetDeviceCount, &di) && di.dli_fname) {
        std::string dir(di.dli_fname);
        const size_t cut = dir.rfind('/');
        if (cut != std::string::npos) {
            dir.resize(cut + 1);
            for (const char *n : {"librccl.so.1", "librccl.so"}) {
                h = dlopen((dir + n).c_str(), RTLD_NOW | RTLD_LOCAL);
                if (h) { snprintf(how, how_cap, "beside the HIP runtime"); return h; }
            }
        }
    }
    for (const char *n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (h) { snprintf(how, how_cap, "default search path"); return h; }
    }
    return nullptr;
}

int rccl_bind() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.handle) return FZ_OK;
    RcclApi a = {};
    void *h = nullptr;
    try {
        h = rccl_open(a.how, sizeof a.how);
    } catch (const std::bad_alloc &) {
        return fz_set_error(FZ_E_HIP, "out of host memory");
    }
    if (!h) return fz_set_error(FZ_E_RCCL, "RCCL not found (librccl.so.1): %s", dlerror());
    a.GetUniqueId = (int (*)(void *))dlsym(h, "ncclGetUniqueId");
    a.CommInitRank = (int (*)(void **, int, fz_unique_id, int))dlsym(h, "ncclCommInitRank");
    a.CommDestroy = (int (*)(void *))dlsym(h, "ncclCommDestroy");
    a.AllReduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(h, "ncclAllReduce");
    a.GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
    a.CommCount = (int (*)(void *, int *))dlsym(h, "ncclCommCount");
    a.GetVersion = (int (*)(int *))dlsym(h, "ncclGetVersion");
    a.Broadcast = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(h, "ncclBroadcast");
    a.ReduceScatter = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(h, "ncclReduceScatter");
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce || !a.GetErrorString)
        return fz_set_error(FZ_E_RCCL, "librccl.so.1 lacks an expected symbol");      // (the handle stays open: never dlclose)
    Dl_info di;
    snprintf(a.path, sizeof a.path, "%s", dladdr((const void *)a.AllReduce, &di) && di.dli_fname ? di.dli_fname : "?");
    a.handle = h;
    g_rccl = a;
    return FZ_OK;
}

int rccl_check(int rc, const char *what) {
    if (rc == 0) return FZ_OK;
    return fz_set_error(FZ_E_RCCL, "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error");
}
}  // namespace

struct fz_comm {
    void *comm;          // ncclComm_t
    int nranks, rank, device;
};

int fz_comm_unique_id(fz_unique_id *out_id) {
    FZ_REQUIRE(out_id, "out_id is NULL");
    FZ_TRY(rccl_bind());
    return rccl_check(g_rccl.GetUniqueId(out_id), "ncclGetUniqueId");
}

int fz_comm_create(fz_ctx *ctx, int nranks, int rank, const fz_unique_id *id, fz_comm **out) {
    FZ_REQUIRE(ctx && id && out && nranks >= 1 && rank >= 0 && rank < nranks, "bad argument");
    *out = nullptr;
    FZ_DEV(ctx);                                          // the communicator binds to the CURRENT device
    FZ_TRY(rccl_bind());
    void *c = nullptr;
    FZ_TRY(rccl_check(g_rccl.CommInitRank(&c, nranks, *id, rank), "ncclCommInitRank"));
    fz_comm *C = new (std::nothrow) fz_comm();
    if (!C) { g_rccl.CommDestroy(c); return fz_set_error(FZ_E_HIP, "out of host memory"); }
    C->comm = c; C->nranks = nranks; C->rank = rank; C->device = ctx->device;
    try {
        std::lock_guard<std::mutex> lk(g_rccl_mu);
        g_live_comms.push_back(C);
    } catch (const std::bad_alloc &) {
        g_rccl.CommDestroy(c);
        delete C;
        return fz_set_error(FZ_E_HIP, "out of host memory");
    }
    *out = C;
    return FZ_OK;
}

// Idempotent: a handle that is not (or no longer) in the registry of live communicators is left alone -- a second destroy
// of the same handle returns FZ_OK without touching RCCL or the freed wrapper.  Communicators still alive when the
// process ends are NOT destroyed by this library (no static destructor, no atexit hook: ncclCommDestroy talks to the
// device, and by then the HIP runtime's own teardown may have begun); RCCL's teardown owns them.
int fz_comm_destroy(fz_comm *comm) {
    if (!comm) return FZ_OK;
    {
        std::lock_guard<std::mutex> lk(g_rccl_mu);
        auto it = std::find(g_live_comms.begin(), g_live_comms.end(), comm);
        if (it == g_live_comms.end()) return FZ_OK;
        g_live_comms.erase(it);
    }
    int rc = FZ_OK;
    if (comm->comm && g_rccl.CommDestroy) {
        (void)hipSetDevice(comm->device);
        (void)hipDeviceSynchronize();                    // nothing of ours may still be using the communicator's streams
        rc = rccl_check(g_rccl.CommDestroy(comm->comm), "ncclCommDestroy");
    }
    comm->comm = nullptr;
    delete comm;
    return rc;
}

// which RCCL serves this process: the file the symbols were bound from, the rule that found it ("already mapped (shared)" |
// "beside the HIP runtime" | "default search path"), and how many DIFFERENT files named librccl* the process has mapped
// (more than one = two copies, the state the binding rule exists to avoid)
int fz_rccl_library(char *out_path, size_t path_cap, char *out_how, size_t how_cap, int *out_copies_mapped) {
    FZ_TRY(rccl_bind());
    if (out_path && path_cap) snprintf(out_path, path_cap, "%s", g_rccl.path);
    if (out_how && how_cap) snprintf(out_how, how_cap, "%s", g_rccl.how);
    if (out_copies_mapped) try {
        std::vector<std::string> seen;
        if (FILE *f = fopen("/proc/self/maps", "r")) {
            char line[1024];
            while (fgets(line, sizeof line, f)) {
                const char *p = strchr(line, '/');
                if (!p || !strstr(p, "librccl")) continue;
                std::string path(p);
                while (!path.empty() && (path.back() == '\n' || path.back() == ' ')) path.pop_back();
                if (std::find(seen.begin(), seen.end(), path) == seen.end()) seen.push_back(path);
            }
            fclose(f);
        }
        *out_copies_mapped = (int)seen.size();
    } catch (const std::bad_alloc &) {
        return fz_set_error(FZ_E_HIP, "out of host memory");
    }
    return FZ_OK;
}

int fz_comm_info(fz_comm *comm, int *out_nranks, int *out_rank) {
    FZ_REQUIRE(comm, "comm is NULL");
    int n = comm->nranks;
    if (g_rccl.CommCount) FZ_TRY(rccl_check(g_rccl.CommCount(comm->comm, &n), "ncclCommCount"));   // what RCCL itself reports
    if (out_nranks) *out_nranks = n;
    if (out_rank) *out_rank = comm->rank;
    return FZ_OK;
}

int fz_rccl_version(int *out_version) {
    FZ_REQUIRE(out_version, "out_version is NULL");
    *out_version = 0;
    FZ_TRY(rccl_bind());
    if (!g_rccl.GetVersion) return fz_set_error(FZ_E_RCCL, "librccl.so.1 lacks ncclGetVersion");
    return rccl_check(g_rccl.GetVersion(out_version), "ncclGetVersion");
}

int fz_broadcast_i32(fz_ctx *ctx, fz_comm *comm, int32_t *d_buf, size_t count, int root) {
    FZ_REQUIRE(ctx && comm && (count == 0 || d_buf), "NULL argument");
    FZ_REQUIRE(root >= 0 && root < comm->nranks, "root outside the communicator");
    if (comm->device != ctx->device) return fz_set_error(FZ_E_BADARG, "communicator was created on device %d", comm->device);
    FZ_DEV(ctx);
    if (count == 0) return FZ_OK;
    if (!g_rccl.Broadcast) return fz_set_error(FZ_E_RCCL, "librccl.so.1 lacks ncclBroadcast");
    // in place, ncclInt32 (= 2), on the context's stream like the all-reduce
    return rccl_check(g_rccl.Broadcast(d_buf, d_buf, count, 2, root, comm->comm, ctx->stream), "ncclBroadcast");
}

int fz_reduce_scatter_i64(fz_ctx *ctx, fz_comm *comm, int64_t *d_buf, size_t count_per_rank) {
    FZ_REQUIRE(ctx && comm && (count_per_rank == 0 || d_buf), "NULL argument");
    if (comm->device != ctx->device) return fz_set_error(FZ_E_BADARG, "communicator was created on device %d", comm->device);
    FZ_DEV(ctx);
    if (count_per_rank == 0) return FZ_OK;
    if (!g_rccl.ReduceScatter) return fz_set_error(FZ_E_RCCL, "librccl.so.1 lacks ncclReduceScatter");
    // in place (NCCL's convention: the receive buffer is this rank's block of the send buffer), ncclInt64 (= 4), ncclSum (= 0)
    return rccl_check(g_rccl.ReduceScatter(d_buf, d_buf + (size_t)comm->rank * count_per_rank, count_per_rank, 4, 0, comm->comm, ctx->stream),
                      "ncclReduceScatter");
}

int fz_allreduce_i64(fz_ctx *ctx, fz_comm *comm, int64_t *d_buf, size_t count) {
    FZ_REQUIRE(ctx && comm && (count == 0 || d_buf), "NULL argument");
    if (comm->device != ctx->device) return fz_set_error(FZ_E_BADARG, "communicator was created on device %d", comm->device);
    FZ_DEV(ctx);
    if (count == 0) return FZ_OK;
    // in place, ncclInt64 (= 4), ncclSum (= 0), on the context's stream: ordered with the kernels around it and
    // capturable by fz_graph_* like them
    return rccl_check(g_rccl.AllReduce(d_buf, d_buf, count, 4, 0, comm->comm, ctx->stream), "ncclAllReduce");
}

}  // extern "C"
