// fz_queue.hip -- asynchronous batch queue below the host language (C ABI section "asynchronous batch queue").
//
// The reference is called once per key and once per signature (fusion/fusion.py:338-373 keygen, :534-557 sign); a caller of
// this library batches them, but at BASELINE's size -- 1024 keys + 1024 signatures per call -- one call is a latency chain
// (MT19937 seeding, ~108 Keccak permutations per signer on a few dozen waves) that leaves most of the chip idle: 0.93 M
// pairs/s from one host thread, and only 3.8-4.3 M/s with 8-16 Python threads, each with a context of its own, and
// GPU_MAX_HW_QUEUES=16 (profiles/r03_concurrent_batches.txt).  The queue closes that gap below Python:
//
//   * submit() copies a call's inputs (seeds, message bytes), hands back a ticket and returns -- microseconds;
//   * W worker threads, each owning a context, a HIP stream and its scratch, take EVERYTHING that is pending (up to
//     max_rows keys) as ONE batch: every row of the path is independent, so n calls of 1024 rows are one launch sequence
//     of n * 1024 rows -- the 16 384-row rate (6.8 M sign/s) instead of sixteen latency chains; results are cut back
//     into the calls' rows, bit-identical to separate calls (tests/test_gpu_queue.py);
//   * while one worker's batch sits in a latency-bound kernel another worker's batch runs beside it (W streams:
//     W <= 4 needs no more than the HIP runtime's default hardware queues);
//   * verification keys go back to the callers' (pinned: fz_pinned_alloc) buffers with asynchronous copies; secret keys
//     and signatures stay in device memory, owned by the batch and released when every call of it has been released.
//
// One context still serves one host thread: a worker's context is touched by that worker alone (buffers of released
// batches are handed back to the worker to free).
//
// Round 5: aggregate() + verify() and verify() calls are queued the same way (fusion.py:655-677, :680-728; the reference is
// called once per aggregate).  An aggregate of a few dozen signers is a launch at the dispatch floor (8 signers 3.5 us,
// 64 signers ~5 us, a single verification 4.6 us) behind milliseconds of host hashing; the worker takes every pending call
// of the kind as ONE batch: the per-signer challenge pipeline of all signers in one launch sequence, hash_ag's serial sponge
// of every aggregate on its own host thread, then ONE ragged launch for all partial sums (fz_aggregate_target_partial_ragged)
// and ONE for all verifications -- aggregate_many / verify_many of BatchScheme, below Python and without blocking the caller.
#include "fz_internal.h"
#include "../../include/fusion_hip.h"

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace {

struct Batch {                         // device results of one coalesced batch, shared by its calls
    int worker;
    int32_t *d_sk, *d_vk, *d_sig;      // fz_malloc blocks of the worker's context (nullptr: worker scratch, not kept)
    std::atomic<int> refs;
    // events recorded on CONSUMERS' streams behind their last use of the rows (fz_queue_release_after); the worker's stream
    // waits for every one of them before the blocks go back to the pool, whose reuse is ordered on the worker's stream only
    // (guarded by the queue's mutex until the batch is garbage, then the worker's alone)
    std::vector<hipEvent_t> after;
};

enum { KIND_KEYGEN_SIGN = 0, KIND_AGGREGATE_VERIFY = 1, KIND_VERIFY = 2 };

struct Job {
    uint64_t ticket;
    size_t n;
    int flags;
    std::vector<uint64_t> seeds;
    std::string msgs;
    std::vector<size_t> off;           // n + 1 offsets into msgs
    int32_t *h_vk_out;                 // [n][2][degree] or nullptr
    // aggregate / verify calls
    int kind = KIND_KEYGEN_SIGN;
    std::vector<int32_t> vk;           // [n][2][degree]: public keys, copied at submit (2 KiB per signer)
    const int32_t *rows = nullptr;     // KIND_AGGREGATE_VERIFY: the signatures [n][rank][degree]; KIND_VERIFY: the aggregate [rank][degree]
                                       // (caller-owned, valid until the call has finished; host unless FZ_QUEUE_ROWS_ON_DEVICE)
    int32_t *h_agg_out = nullptr;      // [rank][degree] or nullptr
    int *h_verdict_out = nullptr;      // FZ_VERDICT_* or nullptr
};

struct Done {
    int status;
    std::string error;
    Batch *batch;
    size_t row0, n;
    bool keep_sk;                      // this call asked for its secret keys (another call of the batch may have: the batch then holds them)
};

struct Worker {
    std::thread th;
    fz_ctx *ctx = nullptr;
    void *stream = nullptr;
    int32_t *d_A = nullptr;
    // scratch that never leaves the worker: secret polynomials, challenges, and -- for batches nobody keeps -- keys and signatures
    int32_t *d_coef = nullptr, *d_c = nullptr, *d_sk = nullptr, *d_vk = nullptr, *d_sig = nullptr;
    size_t cap_coef = 0, cap_c = 0, cap_sk = 0, cap_vk = 0, cap_sig = 0;      // in rows
    std::vector<Batch *> garbage;      // released batches, freed by the worker itself (guarded by the queue's mutex)
    // aggregate / verify batches: everything is scratch (results go to the callers' host buffers)
    int32_t *a_sig = nullptr, *a_vk = nullptr, *a_L = nullptr, *a_R = nullptr, *a_c = nullptr, *a_al = nullptr, *a_agg = nullptr,
            *a_tgt = nullptr, *a_verd = nullptr;
    int32_t *a_psum = nullptr, *a_tsum = nullptr;      // int64 partial sums (kept as int32_t* for grow(): row_bytes says the size)
    size_t c_sig = 0, c_vk = 0, c_L = 0, c_R = 0, c_c = 0, c_al = 0, c_agg = 0, c_tgt = 0, c_verd = 0, c_psum = 0, c_tsum = 0;
};

}  // namespace

struct fz_queue {
    int device, l, degree;
    fz_scheme_params P;
    int64_t beta_sk, omega_sk;
    size_t max_rows;
    std::vector<int32_t> A;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::deque<Job> pending;
    std::unordered_map<uint64_t, Done> done;        // finished calls whose results (or error) somebody may still ask for
    std::unordered_set<uint64_t> live;              // submitted, not finished
    int first_error = FZ_OK;                        // of a FZ_QUEUE_DISCARD call since the last drain
    std::string first_error_text;
    uint64_t next_ticket = 1;
    size_t inflight = 0;               // submitted and not yet completed
    bool stopping = false;
    int init_rc = FZ_OK;
    std::string init_err;
    std::vector<Worker> workers;
    uint64_t st_jobs = 0, st_batches = 0, st_rows = 0;
    // aggregate / verify calls (fz_queue_enable_aggregate)
    bool agg_enabled = false;
    int64_t beta_vf = 0, omega_vf = 0;
    size_t capacity = 0;
    int host_threads = 1;
};

namespace {

const char *last_error() { return fz_last_error(); }

int grow(fz_ctx *ctx, int32_t **p, size_t *cap_rows, size_t rows, size_t row_bytes) {
    if (rows <= *cap_rows) return FZ_OK;
    if (*p) { int rc = fz_free(ctx, *p); *p = nullptr; *cap_rows = 0; if (rc != FZ_OK) return rc; }
    void *q = nullptr;
    int rc = fz_malloc(ctx, rows * row_bytes, &q);
    if (rc != FZ_OK) return rc;
    *p = (int32_t *)q;
    *cap_rows = rows;
    return FZ_OK;
}

void free_batch(fz_queue *Q, Worker &w, Batch *b) {
    (void)Q;
    for (hipEvent_t ev : b->after) {                // the consumers' kernels first: fz_free orders reuse behind THIS stream
        (void)hipStreamWaitEvent((hipStream_t)w.stream, ev, 0);
        (void)hipEventDestroy(ev);                  // (released by the runtime once the wait above has been satisfied)
    }
    b->after.clear();
    if (b->d_sk) (void)fz_free(w.ctx, b->d_sk);
    if (b->d_vk) (void)fz_free(w.ctx, b->d_vk);
    if (b->d_sig) (void)fz_free(w.ctx, b->d_sig);
    delete b;
}

// one coalesced batch: jobs[0..] rows back to back
int run_batch(fz_queue *Q, Worker &w, std::vector<Job> &jobs, Batch **out_batch, std::vector<size_t> &row0) {
    const int d = Q->degree, l = Q->l;
    size_t N = 0, msg_bytes = 0;
    bool keep = false, keep_sk = false;
    row0.reserve(jobs.size());
    for (auto &j : jobs) {
        row0.push_back(N);
        N += j.n;
        msg_bytes += j.msgs.size();
        if (!(j.flags & FZ_QUEUE_DISCARD)) keep = true;
        if ((j.flags & FZ_QUEUE_KEEP_SK) && !(j.flags & FZ_QUEUE_DISCARD)) keep_sk = true;
    }
    std::vector<uint64_t> seeds;
    seeds.reserve(N);
    std::string msgs;
    msgs.reserve(msg_bytes);
    std::vector<size_t> off;
    off.reserve(N + 1);
    off.push_back(0);
    for (auto &j : jobs) {
        seeds.insert(seeds.end(), j.seeds.begin(), j.seeds.end());
        const size_t base = msgs.size();
        msgs += j.msgs;
        for (size_t i = 1; i <= j.n; ++i) off.push_back(base + j.off[i]);
    }
    const size_t poly = (size_t)d * 4;
    int rc;
    if ((rc = grow(w.ctx, &w.d_coef, &w.cap_coef, N, 2 * poly)) != FZ_OK) return rc;
    if ((rc = grow(w.ctx, &w.d_c, &w.cap_c, N, poly)) != FZ_OK) return rc;
    Batch *b = nullptr;
    int32_t *d_sk, *d_vk, *d_sig;
    if (keep) {
        b = new Batch();
        b->worker = (int)(&w - Q->workers.data());
        b->d_sk = b->d_vk = b->d_sig = nullptr;
        b->refs = 0;
        void *p = nullptr;
        if ((rc = fz_malloc(w.ctx, N * 2 * poly, &p)) != FZ_OK) { free_batch(Q, w, b); return rc; }
        b->d_vk = (int32_t *)p;
        if ((rc = fz_malloc(w.ctx, N * (size_t)l * poly, &p)) != FZ_OK) { free_batch(Q, w, b); return rc; }
        b->d_sig = (int32_t *)p;
        if (keep_sk) {
            if ((rc = fz_malloc(w.ctx, N * 2 * (size_t)l * poly, &p)) != FZ_OK) { free_batch(Q, w, b); return rc; }
            b->d_sk = (int32_t *)p;
        }
    }
    if (!b || !b->d_sk) {
        if ((rc = grow(w.ctx, &w.d_sk, &w.cap_sk, N, 2 * (size_t)l * poly)) != FZ_OK) { if (b) free_batch(Q, w, b); return rc; }
    }
    if (!b) {
        if ((rc = grow(w.ctx, &w.d_vk, &w.cap_vk, N, 2 * poly)) != FZ_OK) return rc;
        if ((rc = grow(w.ctx, &w.d_sig, &w.cap_sig, N, (size_t)l * poly)) != FZ_OK) return rc;
    }
    d_sk = (b && b->d_sk) ? b->d_sk : w.d_sk;
    d_vk = b ? b->d_vk : w.d_vk;
    d_sig = b ? b->d_sig : w.d_sig;
    // the path itself, exactly what BatchScheme.keygen_batch + sign_batch issue for one call (fusion.py:338-373, :534-557)
    rc = fz_sample_secret_polys_dev(w.ctx, seeds.data(), N, Q->P.modulus, d, Q->beta_sk, Q->omega_sk, w.d_coef);
    if (rc == FZ_OK) rc = fz_keygen_core_bcast(w.ctx, w.d_A, w.d_coef, d_sk, d_vk, N, l);
    if (rc == FZ_OK) rc = fz_challenge_hat_msgs_dev(w.ctx, &Q->P, d_vk, msgs.data(), off.data(), N, w.d_c, nullptr);
    if (rc == FZ_OK) rc = fz_sign_core(w.ctx, d_sk, w.d_c, d_sig, N, l);
    if (rc == FZ_OK) {
        hipStream_t st = (hipStream_t)w.stream;
        for (size_t k = 0; k < jobs.size() && rc == FZ_OK; ++k)
            if (jobs[k].h_vk_out)
                rc = fz_check_hip(hipMemcpyAsync(jobs[k].h_vk_out, d_vk + row0[k] * 2 * (size_t)d, jobs[k].n * 2 * poly,
                                                 hipMemcpyDeviceToHost, st), "queue: verification keys to the host");
        if (rc == FZ_OK) rc = fz_ctx_synchronize(w.ctx);
    } else {
        (void)fz_ctx_synchronize(w.ctx);
    }
    if (rc != FZ_OK) {
        if (b) free_batch(Q, w, b);
        return rc;
    }
    *out_batch = b;
    return FZ_OK;
}

// one coalesced batch of aggregate+verify (kind 1) or verify (kind 2) calls: G aggregates, their signers back to back.
// The flow of BatchScheme._hash_ag_many + aggregate_target_partial_ragged + verify_partials (fusion_hip/scheme.py), in C.
int run_agg_batch_steps(fz_queue *Q, Worker &w, std::vector<Job> &jobs, int kind, std::vector<int> &verdicts);
// Every exit of the steps below that reports a failure may leave copies FROM THE CALLERS' ROWS in flight on the worker's stream
// (they are enqueued first): the stream is drained before the failure is reported, so that a caller who frees or recycles its
// rows on an error never races with them.
int run_agg_batch(fz_queue *Q, Worker &w, std::vector<Job> &jobs, int kind, std::vector<int> &verdicts) {
    const int rc = run_agg_batch_steps(Q, w, jobs, kind, verdicts);
    if (rc != FZ_OK) (void)fz_ctx_synchronize(w.ctx);
    return rc;
}
int run_agg_batch_steps(fz_queue *Q, Worker &w, std::vector<Job> &jobs, int kind, std::vector<int> &verdicts) {
    const size_t d = (size_t)Q->degree, l = (size_t)Q->l, poly = d * 4, G = jobs.size();
    std::vector<size_t> offs(G + 1, 0);
    size_t msg_bytes = 0;
    for (size_t g = 0; g < G; ++g) { offs[g + 1] = offs[g] + jobs[g].n; msg_bytes += jobs[g].msgs.size(); }
    const size_t N = offs[G];
    verdicts.assign(G, FZ_VERDICT_OK);
    // host staging: keys as one [N][2][d] array and as left / right rows, messages back to back
    std::vector<int32_t> vk(N * 2 * d), L(N * d), R(N * d), c_hat(N * d), alpha(N * d);
    std::vector<uint8_t> pre(N * 32);
    std::string msgs;
    msgs.reserve(msg_bytes);
    std::vector<size_t> off;
    off.reserve(N + 1);
    off.push_back(0);
    for (size_t g = 0; g < G; ++g) {
        memcpy(vk.data() + offs[g] * 2 * d, jobs[g].vk.data(), jobs[g].n * 2 * poly);
        const size_t base = msgs.size();
        msgs += jobs[g].msgs;
        for (size_t i = 1; i <= jobs[g].n; ++i) off.push_back(base + jobs[g].off[i]);
    }
    for (size_t i = 0; i < N; ++i) {
        memcpy(L.data() + i * d, vk.data() + (2 * i) * d, poly);
        memcpy(R.data() + i * d, vk.data() + (2 * i + 1) * d, poly);
    }
    int rc;
    if ((rc = grow(w.ctx, &w.a_vk, &w.c_vk, N, 2 * poly)) != FZ_OK) return rc;
    if ((rc = grow(w.ctx, &w.a_L, &w.c_L, N, poly)) != FZ_OK) return rc;
    if ((rc = grow(w.ctx, &w.a_R, &w.c_R, N, poly)) != FZ_OK) return rc;
    if ((rc = grow(w.ctx, &w.a_c, &w.c_c, N, poly)) != FZ_OK) return rc;
    if ((rc = grow(w.ctx, &w.a_al, &w.c_al, N, poly)) != FZ_OK) return rc;
    if ((rc = grow(w.ctx, &w.a_agg, &w.c_agg, G, l * poly)) != FZ_OK) return rc;
    if ((rc = grow(w.ctx, &w.a_tgt, &w.c_tgt, G, poly)) != FZ_OK) return rc;
    if ((rc = grow(w.ctx, &w.a_verd, &w.c_verd, G, sizeof(int))) != FZ_OK) return rc;
    if ((rc = grow(w.ctx, &w.a_tsum, &w.c_tsum, G, d * 8)) != FZ_OK) return rc;
    if (kind == KIND_AGGREGATE_VERIFY) {
        if ((rc = grow(w.ctx, &w.a_sig, &w.c_sig, N, l * poly)) != FZ_OK) return rc;
        if ((rc = grow(w.ctx, &w.a_psum, &w.c_psum, G, l * d * 8)) != FZ_OK) return rc;
    }
    hipStream_t st = (hipStream_t)w.stream;
    // the signatures start their way to the device first: the largest transfer, and nothing below needs it before the sums
    if (kind == KIND_AGGREGATE_VERIFY)
        for (size_t g = 0; g < G; ++g) {
            const hipMemcpyKind dir = (jobs[g].flags & FZ_QUEUE_ROWS_ON_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
            if ((rc = fz_check_hip(hipMemcpyAsync(w.a_sig + offs[g] * l * d, jobs[g].rows, jobs[g].n * l * poly, dir, st), "queue: signatures")) != FZ_OK) return rc;
        }
    else
        for (size_t g = 0; g < G; ++g) {
            const hipMemcpyKind dir = (jobs[g].flags & FZ_QUEUE_ROWS_ON_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
            if ((rc = fz_check_hip(hipMemcpyAsync(w.a_agg + g * l * d, jobs[g].rows, l * poly, dir, st), "queue: aggregates")) != FZ_OK) return rc;
        }
    // hash_ch of every signer (fusion.py:511-531) on the device; the pre-hashes and c_hat come back for hash_ag's text
    if ((rc = fz_memcpy_h2d(w.ctx, w.a_vk, vk.data(), N * 2 * poly)) != FZ_OK) return rc;
    if ((rc = fz_challenge_hat_msgs_dev(w.ctx, &Q->P, w.a_vk, msgs.data(), off.data(), N, w.a_c, pre.data())) != FZ_OK) return rc;
    if ((rc = fz_memcpy_d2h(w.ctx, c_hat.data(), w.a_c, N * poly)) != FZ_OK) return rc;
    // hash_ag (fusion.py:632-652) per aggregate: sort by str(vk) (:661-663, :693), ONE serial SHAKE-256, decode; the rows go back
    // to the callers' order (the sums do not care).  Independent aggregates on independent host threads.
    {
        std::atomic<size_t> next(0);
        std::atomic<int> first_rc(FZ_OK);
        auto work = [&]() {
            for (;;) {
                const size_t g = next.fetch_add(1);
                if (g >= G) return;
                try {
                    const size_t n = jobs[g].n, o = offs[g];
                    std::vector<size_t> order(n);
                    int r = fz_sort_by_vk_string(&Q->P, L.data() + o * d, R.data() + o * d, n, order.data(), 1);
                    if (r == FZ_OK) {
                        std::vector<int32_t> sL(n * d), sR(n * d), sC(n * d), sA(n * d);
                        std::vector<uint8_t> sP(n * 32);
                        for (size_t i = 0; i < n; ++i) {
                            const size_t k = o + order[i];
                            memcpy(sL.data() + i * d, L.data() + k * d, poly);
                            memcpy(sR.data() + i * d, R.data() + k * d, poly);
                            memcpy(sC.data() + i * d, c_hat.data() + k * d, poly);
                            memcpy(sP.data() + i * 32, pre.data() + k * 32, 32);
                        }
                        r = fz_aggregation_coefficients(&Q->P, sL.data(), sR.data(), sP.data(), sC.data(), n, sA.data(), 1);
                        if (r == FZ_OK)
                            for (size_t i = 0; i < n; ++i) memcpy(alpha.data() + (o + order[i]) * d, sA.data() + i * d, poly);
                    }
                    if (r != FZ_OK) { int e = FZ_OK; first_rc.compare_exchange_strong(e, r); }
                } catch (const std::exception &) {          // (a pool thread: nothing may leave it)
                    int e = FZ_OK;
                    first_rc.compare_exchange_strong(e, FZ_E_HIP);
                }
            }
        };
        const size_t T = std::min<size_t>(G, (size_t)std::max(1, Q->host_threads));
        std::vector<std::thread> pool;
        try {
            for (size_t t = 1; t < T; ++t) pool.emplace_back(work);
        } catch (const std::exception &) {}                 // fewer threads than asked for: the ones that started (and this one) do the work
        work();
        for (auto &th : pool) th.join();
        if (first_rc.load() != FZ_OK) return fz_set_error(first_rc.load(), "queue: hash_ag failed for an aggregate of the batch");
    }
    if ((rc = fz_memcpy_h2d(w.ctx, w.a_al, alpha.data(), N * poly)) != FZ_OK) return rc;
    if ((rc = fz_ntt_forward(w.ctx, w.a_al, w.a_al, N)) != FZ_OK) return rc;
    if ((rc = fz_memcpy_h2d(w.ctx, w.a_L, L.data(), N * poly)) != FZ_OK) return rc;
    if ((rc = fz_memcpy_h2d(w.ctx, w.a_R, R.data(), N * poly)) != FZ_OK) return rc;
    int64_t *tsum = (int64_t *)w.a_tsum, *psum = (int64_t *)w.a_psum;
    if (kind == KIND_AGGREGATE_VERIFY) {
        // ONE launch: exact int64 partial sums of every aggregate (fusion.py:670-676) and of every verification target (:706-714)
        rc = fz_aggregate_target_partial_ragged(w.ctx, w.a_sig, w.a_al, w.a_L, w.a_R, w.a_c, offs.data(), G, (int)l, psum, l * d, tsum, d);
        // ONE launch: every verdict straight from the sums (fusion.py:690-727)
        if (rc == FZ_OK) rc = fz_verify_partials_batch_async(w.ctx, w.d_A, psum, l * d, tsum, d, G, (int)l, Q->beta_vf, Q->omega_vf, (int *)w.a_verd);
        if (rc == FZ_OK) rc = fz_reduce_i64(w.ctx, psum, w.a_agg, G * l * d);
        for (size_t g = 0; g < G && rc == FZ_OK; ++g)
            if (jobs[g].h_agg_out)
                rc = fz_check_hip(hipMemcpyAsync(jobs[g].h_agg_out, w.a_agg + g * l * d, l * poly, hipMemcpyDeviceToHost, st), "queue: aggregate to the host");
    } else {
        rc = fz_aggregate_target_partial_ragged(w.ctx, nullptr, w.a_al, w.a_L, w.a_R, w.a_c, offs.data(), G, (int)l, nullptr, 0, tsum, d);
        if (rc == FZ_OK) rc = fz_reduce_i64(w.ctx, tsum, w.a_tgt, G * d);
        if (rc == FZ_OK) rc = fz_verify_with_target_batch_async(w.ctx, w.d_A, w.a_agg, w.a_tgt, G, (int)l, Q->beta_vf, Q->omega_vf, (int *)w.a_verd);
    }
    if (rc == FZ_OK) rc = fz_memcpy_d2h(w.ctx, verdicts.data(), w.a_verd, G * sizeof(int));       // (synchronises the stream)
    else (void)fz_ctx_synchronize(w.ctx);
    if (rc != FZ_OK) return rc;
    for (size_t g = 0; g < G; ++g) {
        if (jobs[g].n > Q->capacity) verdicts[g] = FZ_VERDICT_TOO_MANY_KEYS;          // fusion.py:686-687: checked before anything else
        if (jobs[g].h_verdict_out) *jobs[g].h_verdict_out = verdicts[g];
    }
    return FZ_OK;
}

void worker_main(fz_queue *Q, int index) {
    Worker &w = Q->workers[index];
    (void)hipSetDevice(Q->device);                  // a new thread starts on device 0; every fz_* call below selects the context's device again
    std::unique_lock<std::mutex> lk(Q->mu);
    for (;;) {
        Q->cv_work.wait(lk, [&] { return Q->stopping || !Q->pending.empty() || !w.garbage.empty(); });
        if (!w.garbage.empty()) {
            std::vector<Batch *> g;
            g.swap(w.garbage);
            lk.unlock();
            for (Batch *b : g) free_batch(Q, w, b);
            lk.lock();
            continue;
        }
        if (Q->pending.empty()) {
            if (Q->stopping) break;
            continue;
        }
        // everything that is pending, up to max_rows keys: one batch
        std::vector<Job> jobs;
        size_t rows = 0;
        const int kind = Q->pending.front().kind;       // calls are taken in order: a batch is a run of calls of ONE kind
        while (!Q->pending.empty() && Q->pending.front().kind == kind && (jobs.empty() || rows + Q->pending.front().n <= Q->max_rows)) {
            rows += Q->pending.front().n;
            jobs.push_back(std::move(Q->pending.front()));
            Q->pending.pop_front();
        }
        if (!Q->pending.empty()) Q->cv_work.notify_one();          // what is left (another kind, or over max_rows) is another worker's
        lk.unlock();
        Batch *b = nullptr;
        std::vector<size_t> row0;
        std::vector<int> verdicts;
        int rc;
        try {
            rc = kind == KIND_KEYGEN_SIGN ? run_batch(Q, w, jobs, &b, row0) : run_agg_batch(Q, w, jobs, kind, verdicts);
        } catch (const std::exception &) {          // the coalesced host copies (bad_alloc) or a thread that would not start: an error of these calls, not std::terminate
            (void)fz_ctx_synchronize(w.ctx);
            b = nullptr;
            rc = fz_set_error(FZ_E_HIP, "queue worker: out of host memory while coalescing %zu calls", jobs.size());
        }
        // Completion bookkeeping: nothing here may leave the thread as an exception (std::terminate in a library thread --
        // ADVICE r04).  A record that cannot be stored (out of host memory) is reported through fz_queue_drain's first error
        // and the call's ticket then reads as finished-with-nothing.
        std::string err;
        try {
            if (rc != FZ_OK) err = last_error();
        } catch (const std::bad_alloc &) {}
        lk.lock();
        int kept = 0;
        for (size_t k = 0; k < jobs.size(); ++k) {
            Q->live.erase(jobs[k].ticket);
            if (jobs[k].flags & FZ_QUEUE_DISCARD) {                 // fire and forget: only a failure is remembered (for fz_queue_drain)
                if (rc != FZ_OK && Q->first_error == FZ_OK) {
                    Q->first_error = rc;
                    try { Q->first_error_text = err; } catch (const std::bad_alloc &) {}
                }
                continue;
            }
            try {
                Done dn;
                dn.status = rc;
                dn.error = err;
                dn.batch = (rc == FZ_OK && b) ? b : nullptr;
                dn.row0 = row0.size() > k ? row0[k] : 0;
                dn.n = jobs[k].n;
                dn.keep_sk = (jobs[k].flags & FZ_QUEUE_KEEP_SK) != 0;
                Q->done.emplace(jobs[k].ticket, std::move(dn));
                if (rc == FZ_OK && b) ++kept;
            } catch (const std::bad_alloc &) {
                if (Q->first_error == FZ_OK) {
                    Q->first_error = FZ_E_HIP;
                    try { Q->first_error_text = "out of host memory while recording a call's result"; } catch (const std::bad_alloc &) {}
                }
            }
        }
        if (b) {
            if (kept) b->refs = kept;
            else {
                try { w.garbage.push_back(b); }
                catch (const std::bad_alloc &) { lk.unlock(); free_batch(Q, w, b); lk.lock(); }
            }
        }
        Q->inflight -= jobs.size();
        Q->st_jobs += jobs.size();
        Q->st_batches += 1;
        Q->st_rows += rows;
        Q->cv_done.notify_all();
    }
    lk.unlock();
    for (int32_t *p : {w.d_coef, w.d_c, w.d_sk, w.d_vk, w.d_sig, w.d_A, w.a_sig, w.a_vk, w.a_L, w.a_R, w.a_c, w.a_al, w.a_agg, w.a_tgt,
                       w.a_verd, w.a_psum, w.a_tsum})
        if (p) (void)fz_free(w.ctx, p);
    if (w.ctx) {
        (void)fz_ctx_set_stream(w.ctx, nullptr);
        if (w.stream) (void)fz_stream_destroy(w.ctx, w.stream);
        (void)fz_ctx_destroy(w.ctx);
    }
}

int worker_init(fz_queue *Q, Worker &w) {
    int rc = fz_ctx_create(Q->device, (uint32_t)Q->P.modulus, Q->degree, (uint32_t)Q->P.root, (uint32_t)Q->P.inv_root, &w.ctx);
    if (rc != FZ_OK) return rc;
    if ((rc = fz_stream_create(w.ctx, &w.stream)) != FZ_OK) return rc;
    if ((rc = fz_ctx_set_stream(w.ctx, w.stream)) != FZ_OK) return rc;
    void *p = nullptr;
    if ((rc = fz_malloc(w.ctx, Q->A.size() * 4, &p)) != FZ_OK) return rc;
    w.d_A = (int32_t *)p;
    if ((rc = fz_memcpy_h2d(w.ctx, w.d_A, Q->A.data(), Q->A.size() * 4)) != FZ_OK) return rc;
    return fz_ctx_synchronize(w.ctx);
}

}  // namespace

extern "C" {

int fz_pinned_alloc(size_t bytes, void **h_out) {
    if (!h_out) return fz_set_error(FZ_E_BADARG, "h_out is NULL");
    *h_out = nullptr;
    return fz_check_hip(hipHostMalloc(h_out, bytes ? bytes : 1, hipHostMallocDefault), "pinned host allocation");
}

int fz_pinned_free(void *h_ptr) {
    if (!h_ptr) return FZ_OK;
    return fz_check_hip(hipHostFree(h_ptr), "pinned host free");
}

int fz_queue_create(int device, const fz_scheme_params *P, int rank, int64_t beta_sk, int64_t omega_sk, const int32_t *h_A,
                    int workers, size_t max_rows, fz_queue **out) {
    if (!P || !h_A || !out) return fz_set_error(FZ_E_BADARG, "NULL argument");
    *out = nullptr;
    if (rank < 1 || workers < 1 || workers > 16 || max_rows < 1)
        return fz_set_error(FZ_E_BADARG, "rank >= 1, 1 <= workers <= 16, max_rows >= 1");
    if (P->modulus <= 0 || P->modulus >= (1ll << 32) || P->degree < 1) return fz_set_error(FZ_E_BADARG, "bad parameter set");
    fz_queue *Q = new (std::nothrow) fz_queue();
    if (!Q) return fz_set_error(FZ_E_HIP, "out of host memory");
    Q->device = device;
    Q->l = rank;
    Q->degree = P->degree;
    Q->P = *P;
    Q->beta_sk = beta_sk;
    Q->omega_sk = omega_sk;
    Q->max_rows = max_rows;
    Q->A.assign(h_A, h_A + (size_t)rank * P->degree);
    Q->workers.resize((size_t)workers);
    // contexts are created HERE, on the caller's thread (errors come back as this call's error), then handed to the threads
    for (auto &w : Q->workers) {
        const int rc = worker_init(Q, w);
        if (rc != FZ_OK) {
            for (auto &v : Q->workers)
                if (v.ctx) {
                    if (v.d_A) (void)fz_free(v.ctx, v.d_A);
                    (void)fz_ctx_set_stream(v.ctx, nullptr);
                    if (v.stream) (void)fz_stream_destroy(v.ctx, v.stream);
                    (void)fz_ctx_destroy(v.ctx);
                }
            delete Q;
            return rc;
        }
    }
    for (int i = 0; i < workers; ++i) Q->workers[(size_t)i].th = std::thread(worker_main, Q, i);
    *out = Q;
    return FZ_OK;
}

int fz_queue_destroy(fz_queue *Q) {
    if (!Q) return FZ_OK;
    {
        std::unique_lock<std::mutex> lk(Q->mu);
        // what was submitted is finished first; results nobody released go back to their workers
        Q->cv_done.wait(lk, [&] { return Q->inflight == 0; });
        for (auto &kv : Q->done)
            if (kv.second.batch && --kv.second.batch->refs == 0) Q->workers[(size_t)kv.second.batch->worker].garbage.push_back(kv.second.batch);
        Q->done.clear();
        Q->stopping = true;
        Q->cv_work.notify_all();
    }
    for (auto &w : Q->workers)
        if (w.th.joinable()) w.th.join();
    delete Q;
    return FZ_OK;
}

int fz_queue_submit_keygen_sign(fz_queue *Q, const uint64_t *h_seeds, size_t n, const char *h_msgs, const size_t *h_msg_off,
                                int32_t *h_vk_out, int flags, uint64_t *out_ticket) {
    if (!Q || !out_ticket || (n && (!h_seeds || !h_msg_off))) return fz_set_error(FZ_E_BADARG, "NULL argument");
    if (n == 0 || n > Q->max_rows) return fz_set_error(FZ_E_BADARG, "between 1 and max_rows (%zu) keys per call", Q->max_rows);
    if (h_msg_off[0] != 0) return fz_set_error(FZ_E_BADARG, "h_msg_off[0] must be 0");
    for (size_t i = 0; i < n; ++i) {
        if (h_msg_off[i + 1] < h_msg_off[i]) return fz_set_error(FZ_E_BADARG, "message offsets must not decrease");
        if (h_seeds[i] == ~0ull) return fz_set_error(FZ_E_UNSUPPORTED, "seed 2^64 - 1: seed + 1 wraps (use the Python sampler)");
    }
    if (h_msg_off[n] && !h_msgs) return fz_set_error(FZ_E_BADARG, "h_msgs is NULL");
    try {                                           // no C++ exception crosses the C ABI
        Job j;
        j.n = n;
        j.flags = flags;
        j.seeds.assign(h_seeds, h_seeds + n);
        j.msgs.assign(h_msgs ? h_msgs : "", h_msg_off[n]);
        j.off.assign(h_msg_off, h_msg_off + n + 1);
        j.h_vk_out = h_vk_out;
        std::lock_guard<std::mutex> lk(Q->mu);
        if (Q->stopping) return fz_set_error(FZ_E_BADARG, "the queue is shutting down");
        Q->live.reserve(Q->live.size() + 1);        // whatever can throw, before the queue's state changes
        Q->pending.push_back(std::move(j));
        Job &q = Q->pending.back();
        q.ticket = Q->next_ticket++;
        *out_ticket = q.ticket;
        Q->live.insert(q.ticket);
        Q->inflight += 1;
    } catch (const std::bad_alloc &) {
        return fz_set_error(FZ_E_HIP, "out of host memory while copying the call's inputs");
    }
    Q->cv_work.notify_one();
    return FZ_OK;
}

int fz_queue_enable_aggregate(fz_queue *Q, int64_t beta_vf, int64_t omega_vf, size_t capacity, int host_threads) {
    if (!Q) return fz_set_error(FZ_E_BADARG, "queue is NULL");
    if (beta_vf < 0 || omega_vf < 0 || capacity < 1 || host_threads < 1 || host_threads > 256)
        return fz_set_error(FZ_E_BADARG, "bounds >= 0, capacity >= 1, 1 <= host_threads <= 256");
    std::lock_guard<std::mutex> lk(Q->mu);
    Q->beta_vf = beta_vf;
    Q->omega_vf = omega_vf;
    Q->capacity = capacity;
    Q->host_threads = host_threads;
    Q->agg_enabled = true;
    return FZ_OK;
}

static int submit_agg(fz_queue *Q, int kind, const int32_t *h_vk, const char *h_msgs, const size_t *h_msg_off, size_t n, const int32_t *rows,
                      int32_t *h_agg_out, int *h_verdict_out, int flags, uint64_t *out_ticket) {
    if (!Q || !out_ticket || !h_vk || !h_msg_off || !rows) return fz_set_error(FZ_E_BADARG, "NULL argument");
    if (!Q->agg_enabled) return fz_set_error(FZ_E_BADARG, "fz_queue_enable_aggregate has not been called");
    if (n == 0 || n > Q->max_rows) return fz_set_error(FZ_E_BADARG, "between 1 and max_rows (%zu) signers per call", Q->max_rows);
    if (n >= ((size_t)1 << 21)) return fz_set_error(FZ_E_UNSUPPORTED, "too many signers for exact accumulation (< 2^21)");
    if (h_msg_off[0] != 0) return fz_set_error(FZ_E_BADARG, "h_msg_off[0] must be 0");
    for (size_t i = 0; i < n; ++i)
        if (h_msg_off[i + 1] < h_msg_off[i]) return fz_set_error(FZ_E_BADARG, "message offsets must not decrease");
    if (h_msg_off[n] && !h_msgs) return fz_set_error(FZ_E_BADARG, "h_msgs is NULL");
    if (flags & ~FZ_QUEUE_ROWS_ON_DEVICE) return fz_set_error(FZ_E_BADARG, "only FZ_QUEUE_ROWS_ON_DEVICE applies to aggregate / verify calls");
    try {
        Job j;
        j.kind = kind;
        j.n = n;
        j.flags = flags;
        j.h_vk_out = nullptr;
        j.vk.assign(h_vk, h_vk + n * 2 * (size_t)Q->degree);
        j.msgs.assign(h_msgs ? h_msgs : "", h_msg_off[n]);
        j.off.assign(h_msg_off, h_msg_off + n + 1);
        j.rows = rows;
        j.h_agg_out = h_agg_out;
        j.h_verdict_out = h_verdict_out;
        std::lock_guard<std::mutex> lk(Q->mu);
        if (Q->stopping) return fz_set_error(FZ_E_BADARG, "the queue is shutting down");
        Q->live.reserve(Q->live.size() + 1);
        Q->pending.push_back(std::move(j));
        Job &q = Q->pending.back();
        q.ticket = Q->next_ticket++;
        *out_ticket = q.ticket;
        Q->live.insert(q.ticket);
        Q->inflight += 1;
    } catch (const std::bad_alloc &) {
        return fz_set_error(FZ_E_HIP, "out of host memory while copying the call's inputs");
    }
    Q->cv_work.notify_one();
    return FZ_OK;
}

int fz_queue_submit_aggregate_verify(fz_queue *Q, const int32_t *h_vk, const char *h_msgs, const size_t *h_msg_off, size_t n,
                                     const int32_t *sig, int32_t *h_agg_out, int *h_verdict_out, int flags, uint64_t *out_ticket) {
    return submit_agg(Q, KIND_AGGREGATE_VERIFY, h_vk, h_msgs, h_msg_off, n, sig, h_agg_out, h_verdict_out, flags, out_ticket);
}

int fz_queue_submit_verify(fz_queue *Q, const int32_t *h_vk, const char *h_msgs, const size_t *h_msg_off, size_t n,
                           const int32_t *aggregate, int *h_verdict_out, int flags, uint64_t *out_ticket) {
    if (!h_verdict_out) return fz_set_error(FZ_E_BADARG, "h_verdict_out is NULL");
    return submit_agg(Q, KIND_VERIFY, h_vk, h_msgs, h_msg_off, n, aggregate, nullptr, h_verdict_out, flags, out_ticket);
}

int fz_queue_wait(fz_queue *Q, uint64_t ticket, fz_queue_result *out) {
    if (!Q) return fz_set_error(FZ_E_BADARG, "queue is NULL");
    std::unique_lock<std::mutex> lk(Q->mu);
    if (ticket == 0 || ticket >= Q->next_ticket) return fz_set_error(FZ_E_BADARG, "unknown ticket");
    Q->cv_done.wait(lk, [&] { return Q->live.count(ticket) == 0; });
    auto it = Q->done.find(ticket);
    if (it == Q->done.end()) {                               // a FZ_QUEUE_DISCARD call, or released before: finished, nothing to hand out
        if (out) memset(out, 0, sizeof(*out));
        return FZ_OK;
    }
    const Done &dn = it->second;
    if (out) {
        memset(out, 0, sizeof(*out));
        out->status = dn.status;
        out->n = dn.n;
        if (dn.batch) {
            const size_t d = (size_t)Q->degree, l = (size_t)Q->l;
            out->d_vk = dn.batch->d_vk + dn.row0 * 2 * d;
            out->d_sig = dn.batch->d_sig + dn.row0 * l * d;
            out->d_sk_hat = (dn.keep_sk && dn.batch->d_sk) ? dn.batch->d_sk + dn.row0 * 2 * l * d : nullptr;
        }
    }
    if (dn.status != FZ_OK) return fz_set_error(dn.status, "queued batch failed: %s", dn.error.c_str());
    return FZ_OK;
}

// consumer != NULL: the rows may still be in use by work ALREADY QUEUED on the consumer's stream -- an event recorded there now
// is what the owning worker's stream waits for before the blocks return to the pool (the pool orders reuse on the
// worker's stream only, so a release right behind an asynchronous kernel on the rows let the next batch overwrite them)
static int queue_release(fz_queue *Q, uint64_t ticket, fz_ctx *consumer) {
    if (!Q) return fz_set_error(FZ_E_BADARG, "queue is NULL");
    hipEvent_t ev = nullptr;
    if (consumer) {
        if (hipSetDevice(consumer->device) != hipSuccess) return fz_set_error(FZ_E_HIP, "cannot select the consumer's device");
        int rc = fz_check_hip(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "queue: release event");
        if (rc == FZ_OK) rc = fz_check_hip(hipEventRecord(ev, consumer->stream), "queue: release event record");
        if (rc != FZ_OK) { if (ev) (void)hipEventDestroy(ev); return rc; }
    }
    std::unique_lock<std::mutex> lk(Q->mu);
    if (ticket == 0 || ticket >= Q->next_ticket) { if (ev) (void)hipEventDestroy(ev); return fz_set_error(FZ_E_BADARG, "unknown ticket"); }
    Q->cv_done.wait(lk, [&] { return Q->live.count(ticket) == 0; });
    auto it = Q->done.find(ticket);
    if (it == Q->done.end()) { if (ev) (void)hipEventDestroy(ev); return FZ_OK; }      // released before: idempotent
    Batch *b = it->second.batch;
    Q->done.erase(it);
    int rc = FZ_OK;
    if (b) {
        if (ev) {
            try { b->after.push_back(ev); ev = nullptr; }
            catch (const std::bad_alloc &) {}
        }
        if (ev) {                                           // could not be attached: wait for the consumer here instead
            lk.unlock();
            rc = fz_check_hip(hipEventSynchronize(ev), "queue: release event wait");
            (void)hipEventDestroy(ev);
            ev = nullptr;
            lk.lock();
        }
        if (--b->refs == 0) {
            try { Q->workers[(size_t)b->worker].garbage.push_back(b); }
            catch (const std::bad_alloc &) { ++b->refs; return fz_set_error(FZ_E_HIP, "out of host memory while releasing a call"); }
            Q->cv_work.notify_all();
        }
    }
    if (ev) (void)hipEventDestroy(ev);
    return rc;
}

int fz_queue_release(fz_queue *Q, uint64_t ticket) { return queue_release(Q, ticket, nullptr); }

int fz_queue_release_after(fz_queue *Q, uint64_t ticket, fz_ctx *consumer) {
    if (!consumer) return fz_set_error(FZ_E_BADARG, "consumer context is NULL (fz_queue_release is the form without one)");
    return queue_release(Q, ticket, consumer);
}

int fz_queue_drain(fz_queue *Q) {
    if (!Q) return fz_set_error(FZ_E_BADARG, "queue is NULL");
    std::unique_lock<std::mutex> lk(Q->mu);
    Q->cv_done.wait(lk, [&] { return Q->inflight == 0; });
    if (Q->first_error != FZ_OK) {
        const int rc = Q->first_error;
        const std::string text = Q->first_error_text;
        Q->first_error = FZ_OK;
        Q->first_error_text.clear();
        return fz_set_error(rc, "a discarded call failed: %s", text.c_str());
    }
    return FZ_OK;
}

int fz_queue_stats(fz_queue *Q, uint64_t *out_calls, uint64_t *out_batches, uint64_t *out_rows) {
    if (!Q) return fz_set_error(FZ_E_BADARG, "queue is NULL");
    std::lock_guard<std::mutex> lk(Q->mu);
    if (out_calls) *out_calls = Q->st_jobs;
    if (out_batches) *out_batches = Q->st_batches;
    if (out_rows) *out_rows = Q->st_rows;
    return FZ_OK;
}

}  // extern "C"
