// fz_ntt.hip -- batched negacyclic NTT / INTT kernels for gfx950.
//
// Computes exactly what cooley_tukey_ntt (algebra/ntt.py:216-291) and gentleman_sande_intt
// (algebra/ntt.py:294-377) compute for each row: natural order in / bit-reversed order out
// for the forward transform, the reverse for the inverse (including the n^{-1} scaling),
// every output the centred residue.  The butterflies are the Longa-Naehrig merged-twiddle
// butterflies of the reference; only the schedule differs.
//
// Schedule (degree D = 16*L, 32 <= D <= 256; tools/ntt_layout_model.py is the index model):
//   * L lanes of a wave own one polynomial, 16 coefficients per lane, 64/L polynomials per
//     wave, one wave per workgroup (so the only synchronisation is wave-local).
//   * strided pass: lane r holds x[r + L*k], k = 0..15.  The four stages with butterfly
//     distance >= L pair registers of the same lane, and their twiddles depend only on k:
//     they are wave-uniform and arrive as scalar (SGPR) operands from the kernarg segment.
//   * one transpose through LDS (padded rows: conflict-free ds_write_b64 / ds_read_b128).
//   * contiguous pass: lane b holds x[16b .. 16b+15]; the remaining log2(D)-4 stages are
//     again register-local.  Their twiddles differ per lane: a [NE][L] table staged in LDS.
//   * values are exact integers in fp64 with lazy accumulation (fz_arith.h): one 6-op
//     FMA-Barrett multiply + add + sub per butterfly, one centring per output.
//   * a resident grid of single-wave workgroups strides over the batch.
// D <= 16 uses a thread-per-polynomial kernel (all twiddles uniform).
#include "fz_internal.h"
#include "../../include/fusion_hip.h"
#include "../../include/fusion_hip_diag.h"
#include <hip/hip_ext.h>
#include <type_traits>

namespace {
typedef int fz_v4i __attribute__((ext_vector_type(4)));

template <int LOGD>
struct Geom {
    static constexpr int D = 1 << LOGD;
    static constexpr int L = D / 16;              // lanes per polynomial
    static constexpr int PPW = 64 / L;            // polynomials per wave
    static constexpr int SB = LOGD - 4;           // stages of the contiguous pass
    static constexpr int NE = 16 - (16 >> SB);    // per-lane twiddles of the contiguous pass
    static constexpr int PS = D + 2 * (D / 16);   // doubles per polynomial in LDS (16-B pad per 16)
};

__device__ __forceinline__ int pad16(int j) { return j + 2 * (j >> 4); }

// ------------------------------------------------------------------------------------------
// Global <-> LDS staging shared by both directions.
// A wave-task covers PPW consecutive polynomials = ONE contiguous chunk of 1024 int32 (4 KiB) of
// the batch, whatever the degree.  All global traffic is 16 bytes per lane, 1 KiB contiguous per
// wave instruction (4 instructions per task); the lane <-> coefficient mappings the passes need are
// produced by LDS reads/writes.  int32 staging image: chunk element j at word j + 4*(j>>4)
// (20-word rows: the 16-byte-per-lane accesses at a 64-byte lane stride stay conflict free).
// ------------------------------------------------------------------------------------------
constexpr int kChunk = 1024;                         // int32 per wave-task
constexpr int kStageWords = kChunk + 4 * (kChunk / 16);   // 1280 words = 5 KiB

__device__ __forceinline__ int pad4(int j) { return j + 4 * (j >> 4); }

struct Chunk { int4 v0, v1, v2, v3; };

// issue the task's 4 coalesced 16-byte loads.  `task` is wave-uniform, so "does the whole chunk lie inside the batch" is a
// scalar test: every chunk but a ragged last one takes ONE scalar base and the lane's 32-bit offset (the four loads differ
// in their immediate offsets only); the ragged one clamps each piece to the last valid 16 bytes.
// Streaming loads: the 16-per-lane kernels run on batches far larger than the caches and read every input once
// (+2-4 % at 2^18..2^20 rows, +9 % at 2^16 with cold inputs; the radix-4 kernels, used for small batches whose
// data may well be cache-resident, keep normal loads: streaming ones cost them 3-5 % at 2^12 rows)
__device__ __forceinline__ int4 nt_load4(const int32_t *p) {
    const fz_v4i t = __builtin_nontemporal_load(reinterpret_cast<const fz_v4i *>(p));
    return make_int4(t.x, t.y, t.z, t.w);
}

__device__ __forceinline__ Chunk chunk_load(const int32_t *in, size_t task, size_t total, int lane) {
    Chunk c;
    if ((task + 1) * kChunk <= total) {
        const int32_t *b = in + task * kChunk;
        c.v0 = nt_load4(b + 4 * lane);
        c.v1 = nt_load4(b + 4 * lane + 256);
        c.v2 = nt_load4(b + 4 * lane + 512);
        c.v3 = nt_load4(b + 4 * lane + 768);
    } else {
        const size_t base = task * kChunk + 4 * lane;
        const size_t last = total - 4;
        c.v0 = nt_load4(in + (base < total ? base : last));
        c.v1 = nt_load4(in + (base + 256 < total ? base + 256 : last));
        c.v2 = nt_load4(in + (base + 512 < total ? base + 512 : last));
        c.v3 = nt_load4(in + (base + 768 < total ? base + 768 : last));
    }
    return c;
}

__device__ __forceinline__ void chunk_to_lds(int32_t *stage, int lane, const Chunk &c) {
    *reinterpret_cast<int4 *>(stage + pad4(4 * lane)) = c.v0;
    *reinterpret_cast<int4 *>(stage + pad4(256 + 4 * lane)) = c.v1;
    *reinterpret_cast<int4 *>(stage + pad4(512 + 4 * lane)) = c.v2;
    *reinterpret_cast<int4 *>(stage + pad4(768 + 4 * lane)) = c.v3;
}

// Wave-local synchronisation.  Every LDS exchange in these kernels is between lanes of ONE wave
// (each wave owns a private staging region), and a wave's DS instructions execute in order, so no
// s_barrier is needed: the release/acquire pair makes the compiler wait for the outstanding LDS
// operations (s_waitcnt lgkmcnt(0)) and keeps it from moving LDS accesses across this point.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

constexpr int kWavesPerBlock = 4;

// Streaming (non-temporal) stores for outputs the kernel never reads back.  A normal store allocates the line
// dirty in this XCD's 4 MiB L2; for a transform that writes as much as it reads, half of the L2 then holds data
// nobody will hit, and the dirty lines are written back in bursts (and at the end of the kernel).  Measured on the
// NTT kernels: 2^14..2^18 rows 14-20 % faster (2^18 rows: 66 % -> 77 % of HBM peak), the bench's 2^12 rows 3-5 %.
__device__ __forceinline__ void nt_store4(int32_t *p, const int4 &v) {
    fz_v4i t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<fz_v4i *>(p));
}

// the task's 4 coalesced 16-byte stores (same scalar split as chunk_load: only a ragged last chunk predicates its lanes)
__device__ __forceinline__ void chunk_store(int32_t *out, size_t task, size_t total, int lane, const int4 &o0, const int4 &o1,
                                            const int4 &o2, const int4 &o3) {
    if ((task + 1) * kChunk <= total) {
        int32_t *b = out + task * kChunk;
        nt_store4(b + 4 * lane, o0);
        nt_store4(b + 4 * lane + 256, o1);
        nt_store4(b + 4 * lane + 512, o2);
        nt_store4(b + 4 * lane + 768, o3);
    } else {
        const size_t base = task * kChunk + 4 * lane;
        if (base < total) nt_store4(out + base, o0);
        if (base + 256 < total) nt_store4(out + base + 256, o1);
        if (base + 512 < total) nt_store4(out + base + 512, o2);
        if (base + 768 < total) nt_store4(out + base + 768, o3);
    }
}

// one twiddle multiply: 4-op pseudo-Mersenne form when FAST (operand bound |a| <= 2^38), else 6-op
template <bool FAST>
__device__ __forceinline__ double tw_mul(double a, double w, double w2, const FzMod m) {
    return FAST ? fz_mulmod4(a, w, w2, m) : fz_mulmod(a, w, m);
}

// ------------------------------------------------------------------------------------------
// forward: strided pass -> transpose -> contiguous pass
// ------------------------------------------------------------------------------------------
// doubles of LDS a workgroup of the 16-per-lane kernels needs: a transpose region per wave + the per-lane twiddle table
template <int LOGD> constexpr int lds16_doubles() {
    using G = Geom<LOGD>;
    return kWavesPerBlock * G::PPW * G::PS + 2 * G::NE * G::L;
}

// The two passes of the 16-per-lane forward transform on a lane's registers: in, a[k] = element r + L*k of the lane's polynomial
// (|a| <= 2^31); out, a[k] = element 16 * lane' + k of the transform in the order algebra/ntt.py:271-291 leaves it (lane' = the
// lane's index inside its polynomial), NOT reduced (|a| < 2^(34+SB)).  `row` is the polynomial's transpose buffer in LDS; the
// caller has finished reading whatever the buffer held before (a wave_sync) and may write it again after the return.
template <int LOGD, bool FAST, class TA>
__device__ __forceinline__ void fwd16_passes(double (&a)[16], double *row, const int r, const double2 *s_tw, const TA &twA,
                                             const FzMod &m) {
    using G = Geom<LOGD>;
    constexpr int L = G::L, SB = G::SB;
    // strided pass: a 16-point LN transform over k with table entries 1..15 (|a| < 2^34 throughout)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int tk = 8 >> s;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (k & tk) continue;
            const int e = (1 << s) + (k >> (4 - s));
            const double v = tw_mul<FAST>(a[k + tk], twA.w[e], twA.w2[e], m);
            const double u = a[k];
            a[k] = u + v;
            a[k + tk] = u - v;
        }
    }

    // transpose: element j = r + L*k  ->  lane j/16, register j%16
#pragma unroll
    for (int k = 0; k < 16; ++k) (row + r)[pad16(L * k)] = a[k];       // = row[pad16(r + L * k)]: r < L and L divides 16 (constant offsets)
    wave_sync();
    {
        const double2 *blk = reinterpret_cast<const double2 *>(row + 18 * r);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            double2 t = blk[k];
            a[2 * k] = t.x;
            a[2 * k + 1] = t.y;
        }
    }
    wave_sync();

    // contiguous pass: stages with distance 2^(SB-1) .. 1, per-lane twiddles
#pragma unroll
    for (int ls = 0; ls < SB; ++ls) {
        const int t = 1 << (SB - 1 - ls);
        const int ebase = (16 >> SB) * ((1 << ls) - 1);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (k & t) continue;
            const int g = k >> (SB - ls);
            const double2 w = s_tw[(ebase + g) * L + r];
            const double v = tw_mul<FAST>(a[k + t], w.x, w.y, m);
            const double u = a[k];
            a[k] = u + v;
            a[k + t] = u - v;
        }
    }
}

// ... and of the inverse: in, a[k] = element 16 * lane' + k (|a| <= 2^31); out, a[k] = element r + L*k, scaled by n^-1, NOT
// centred (|a| <= q/2 + q * 2^-13: every output has passed the last stage's multiply).
template <int LOGD, bool FAST, class TA>
__device__ __forceinline__ void inv16_passes(double (&a)[16], double *row, const int r, const double2 *s_tw, const TA &twA,
                                             const FzMod &m) {
    using G = Geom<LOGD>;
    constexpr int L = G::L, SB = G::SB;
    // contiguous pass: GS stages with distance 1, 2, .. 2^(SB-1); operands |u - v| <= 2^(32+ls)
#pragma unroll
    for (int ls = 0; ls < SB; ++ls) {
        const int t = 1 << ls;
        const int ebase = 16 - (16 >> ls);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (k & t) continue;
            const int g = k >> (ls + 1);
            const double2 w = s_tw[(ebase + g) * L + r];
            const double u = a[k], v = a[k + t];
            a[k] = u + v;
            a[k + t] = tw_mul<FAST>(u - v, w.x, w.y, m);
        }
    }

    // After the contiguous pass a[0] (the sum of the lane's 16 inputs, up to 2^(31+SB)) is the one value no multiply
    // has reduced; a[1] <= 2^(29+SB), the rest less.  One fold (2 ops) brings the largest operand of the strided pass
    // down to 2^(29+SB) * 2^4 <= 2^37: the last stage can then use the 4-op multiply (16 x 2 ops saved per lane).
    if (FAST && 31 + SB + 4 > 38) a[0] = fz_fold(a[0], m);
    // transpose back to the strided layout
    {
        double2 *blk = reinterpret_cast<double2 *>(row + 18 * r);
#pragma unroll
        for (int k = 0; k < 8; ++k) blk[k] = make_double2(a[2 * k], a[2 * k + 1]);
    }
    wave_sync();
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = (row + r)[pad16(L * k)];       // = row[pad16(r + L * k)] (see fwd16_passes)
    wave_sync();

    // strided pass: GS stages with distance L, 2L, 4L, 8L; uniform twiddles; n^-1 folded into the last stage.
    // Operands stay below 2^38 (see the fold above), so every stage uses the 4-op multiply when the modulus admits it.
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int tk = 1 << s;
        const int h = 8 >> s;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (k & tk) continue;
            const double u = a[k], v = a[k + tk];
            if (s == 3) {
                a[k] = tw_mul<FAST>(u + v, twA.n_inv, twA.n_inv2, m);
                a[k + tk] = tw_mul<FAST>(u - v, twA.w1_n_inv, twA.w1_n_inv2, m);
            } else {
                const int e = h + (k >> (s + 1));
                a[k] = u + v;
                a[k + tk] = tw_mul<FAST>(u - v, twA.w[e], twA.w2[e], m);
            }
        }
    }
}

// the whole forward kernel as a function of (block index, blocks that share the batch): ntt_fwd16 runs it over the grid,
// ntt_jobs16 over the run of workgroups a job owns
template <int LOGD, bool FAST>
__device__ __forceinline__ void fwd16_run(const int32_t *in, int32_t *out, size_t batch, unsigned block, unsigned nblocks, double *lds,
                                          const double2 *__restrict__ twB, const FzTwA &twA, const FzMod &m) {
    using G = Geom<LOGD>;
    constexpr int D = G::D, L = G::L, PPW = G::PPW, SB = G::SB, NE = G::NE, PS = G::PS;
    constexpr int REGION = PPW * PS;                      // doubles per wave
    static_assert(REGION * 2 >= kStageWords, "staging image must fit in the transpose buffer");
    double2 *s_tw = reinterpret_cast<double2 *>(lds + kWavesPerBlock * REGION);      // (w, w2) pairs, [NE][L]

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;      // the wave index is uniform: say so (scalar address arithmetic)
    const int p = lane / L, r = lane % L;
    const size_t total = batch * D;
    const size_t tasks = (total + kChunk - 1) / kChunk;
    const size_t first = (size_t)block * kWavesPerBlock + wave;
    const size_t stride = (size_t)nblocks * kWavesPerBlock;
    // the wave's first chunk is requested BEFORE the twiddle table is staged: two memory latencies overlapped instead of added (a
    // launch of 2^16 rows is four iterations per wave: a microsecond of start-up is 4 % of it)
    Chunk raw0 = {};
    if (first < tasks) raw0 = chunk_load(in, first, total, lane);
    for (int i = threadIdx.x; i < NE * L; i += 64 * kWavesPerBlock) s_tw[i] = twB[i];
    __syncthreads();                                      // the only workgroup-wide barrier
    double *region = lds + wave * REGION;
    int32_t *stage = reinterpret_cast<int32_t *>(region);
    double *row = region + p * PS;
    if (first >= tasks) return;
    // Software pipeline.  gfx9 has ONE in-order counter (vmcnt) for loads and stores, so a wait for a
    // prefetched load also waits for every store issued before... and, at a loop header, the compiler must
    // assume the worst over all entry paths.  Each iteration therefore (1) issues the NEXT chunk's loads
    // first, (2) computes, (3) moves the finished outputs LDS -> registers, (4) waits for the prefetched
    // chunk and stages it into LDS, and only then (5) issues the global stores: the stores are always the
    // youngest outstanding operations and nothing waits for their completion until a whole iteration later.
    chunk_to_lds(stage, lane, raw0);

    // One iteration; MORE = another chunk of this wave follows (its loads are issued first).  The loop runs the MORE form and
    // the wave's last chunk is peeled off as the other: rounds 1-4 issued the loads unconditionally and re-loaded the CURRENT
    // chunk on a wave's last iteration (a quarter more read requests at the four iterations of a multi-job launch: the PMC pass
    // over round 5's headline read 78.2 MB per launch where 64 MiB are due), and a run-time `if (more)` around loads and staging
    // made the compiler wait for ALL memory operations -- the previous iteration's stores -- at the loop header (two
    // branches on one condition are two paths to its wait-count pass): 3-8 % on the stand-alone kernels.
    auto iteration = [&](const size_t task, auto more_tag) __attribute__((always_inline)) {
        constexpr bool more = decltype(more_tag)::value;
        Chunk raw = {};
        if (more) raw = chunk_load(in, task + stride, total, lane);
        wave_sync();
        double a[16];
        {
            int x[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) x[k] = stage[pad4(p * D + r + L * k)];
#pragma unroll
            for (int k = 0; k < 16; ++k) a[k] = (double)x[k];
        }
        wave_sync();

        fwd16_passes<LOGD, FAST>(a, row, r, s_tw, twA, m);

        // lane holds chunk elements [16*lane, 16*lane + 16): centre, stage, store coalesced
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int4 o;
            o.x = (int)fz_cent(a[4 * k + 0], m);
            o.y = (int)fz_cent(a[4 * k + 1], m);
            o.z = (int)fz_cent(a[4 * k + 2], m);
            o.w = (int)fz_cent(a[4 * k + 3], m);
            *reinterpret_cast<int4 *>(stage + pad4(16 * lane + 4 * k)) = o;
        }
        wave_sync();
        const int4 o0 = *reinterpret_cast<const int4 *>(stage + pad4(4 * lane));
        const int4 o1 = *reinterpret_cast<const int4 *>(stage + pad4(256 + 4 * lane));
        const int4 o2 = *reinterpret_cast<const int4 *>(stage + pad4(512 + 4 * lane));
        const int4 o3 = *reinterpret_cast<const int4 *>(stage + pad4(768 + 4 * lane));
        wave_sync();
        if (more) chunk_to_lds(stage, lane, raw);   // waits for the prefetched loads (no store is younger)
        chunk_store(out, task, total, lane, o0, o1, o2, o3);
    };
    size_t task = first;
    for (; task + stride < tasks; task += stride) iteration(task, std::true_type());
    iteration(task, std::false_type());
}

template <int LOGD, bool FAST>
__global__ __launch_bounds__(64 * kWavesPerBlock) void ntt_fwd16(const int32_t *in, int32_t *out, size_t batch,
                                                                 const double2 *__restrict__ twB, FzTwA twA, FzMod m) {
    __shared__ __attribute__((aligned(16))) double lds[lds16_doubles<LOGD>()];
    fwd16_run<LOGD, FAST>(in, out, batch, blockIdx.x, gridDim.x, lds, twB, twA, m);
}

// ------------------------------------------------------------------------------------------
// inverse: contiguous pass -> transpose -> strided pass (n^{-1} folded into the last stage)
// ------------------------------------------------------------------------------------------
template <int LOGD, bool FAST>
__device__ __forceinline__ void inv16_run(const int32_t *in, int32_t *out, size_t batch, unsigned block, unsigned nblocks, double *lds,
                                          const double2 *__restrict__ itwB, const FzTwA &twA, const FzMod &m) {
    using G = Geom<LOGD>;
    constexpr int D = G::D, L = G::L, PPW = G::PPW, SB = G::SB, NE = G::NE, PS = G::PS;
    constexpr int REGION = PPW * PS;
    double2 *s_tw = reinterpret_cast<double2 *>(lds + kWavesPerBlock * REGION);

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;      // the wave index is uniform: say so (scalar address arithmetic)
    const int p = lane / L, r = lane % L;
    const size_t total = batch * D;
    const size_t tasks = (total + kChunk - 1) / kChunk;
    const size_t first = (size_t)block * kWavesPerBlock + wave;
    const size_t stride = (size_t)nblocks * kWavesPerBlock;
    Chunk raw0 = {};
    if (first < tasks) raw0 = chunk_load(in, first, total, lane);      // before the table: see fwd16_run
    for (int i = threadIdx.x; i < NE * L; i += 64 * kWavesPerBlock) s_tw[i] = itwB[i];
    __syncthreads();
    double *region = lds + wave * REGION;
    int32_t *stage = reinterpret_cast<int32_t *>(region);
    double *row = region + p * PS;
    if (first >= tasks) return;
    chunk_to_lds(stage, lane, raw0);

    auto iteration = [&](const size_t task, auto more_tag) __attribute__((always_inline)) {       // pipeline and peeling: see fwd16_run
        constexpr bool more = decltype(more_tag)::value;
        Chunk raw = {};
        if (more) raw = chunk_load(in, task + stride, total, lane);
        wave_sync();
        double a[16];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int4 t = *reinterpret_cast<const int4 *>(stage + pad4(16 * lane + 4 * k));
            a[4 * k + 0] = (double)t.x;
            a[4 * k + 1] = (double)t.y;
            a[4 * k + 2] = (double)t.z;
            a[4 * k + 3] = (double)t.w;
        }
        wave_sync();

        inv16_passes<LOGD, FAST>(a, row, r, s_tw, twA, m);

#pragma unroll
        for (int k = 0; k < 16; ++k) stage[pad4(p * D + r + L * k)] = (int)fz_cent(a[k], m);
        wave_sync();
        const int4 o0 = *reinterpret_cast<const int4 *>(stage + pad4(4 * lane));
        const int4 o1 = *reinterpret_cast<const int4 *>(stage + pad4(256 + 4 * lane));
        const int4 o2 = *reinterpret_cast<const int4 *>(stage + pad4(512 + 4 * lane));
        const int4 o3 = *reinterpret_cast<const int4 *>(stage + pad4(768 + 4 * lane));
        wave_sync();
        if (more) chunk_to_lds(stage, lane, raw);
        chunk_store(out, task, total, lane, o0, o1, o2, o3);
    };
    size_t task = first;
    for (; task + stride < tasks; task += stride) iteration(task, std::true_type());
    iteration(task, std::false_type());
}

template <int LOGD, bool FAST>
__global__ __launch_bounds__(64 * kWavesPerBlock) void ntt_inv16(const int32_t *in, int32_t *out, size_t batch,
                                                                 const double2 *__restrict__ itwB, FzTwA twA, FzMod m) {
    __shared__ __attribute__((aligned(16))) double lds[lds16_doubles<LOGD>()];
    inv16_run<LOGD, FAST>(in, out, batch, blockIdx.x, gridDim.x, lds, itwB, twA, m);
}

// ------------------------------------------------------------------------------------------
// 4 coefficients per lane ("radix-4 in place"): the low-latency schedule for batches that give the
// 16-per-lane kernels less than a few waves per SIMD (BASELINE's B = 4096 is one wave per SIMD
// there).  D/4 lanes own a polynomial; log4(D) passes of two stages each on 4 registers
// {base + k*s}, s = D/4, D/16, .., 1; between passes the polynomial lives in LDS as doubles at
// XOR-swizzled natural positions (conflict-free ds_read/write_b64 for every pass stride, 2-way on
// the final 16-byte accesses).  Global traffic needs no staging: the first pass reads
// x[m + (D/4)k] (256 B contiguous per wave instruction), the last leaves 4 contiguous outputs per
// lane (16-byte coalesced stores); mirrored for the inverse.  Pass 0 twiddles are wave-uniform
// (SGPR); each later pass uses 3 per-lane twiddles kept in registers across tasks.
// Only even log2(D) (the scheme's degrees 64 and 256).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int swz4(int j) { return j ^ (((j >> 4) & 7) << 2); }


// per-lane twiddles are kept either as (w, w * K/q) pairs or as w alone with the quotient twiddle recomputed at each use
// (one more fp64 multiply per twiddle and pass, half the registers: the fused kernels trade it for occupancy)
__device__ __forceinline__ double tw_w(const double2 &t) { return t.x; }
__device__ __forceinline__ double tw_q(const double2 &t, const FzMod &) { return t.y; }
__device__ __forceinline__ double tw_w(const double &t) { return t; }
__device__ __forceinline__ double tw_q(const double &t, const FzMod &m) {
    double w = t;
    asm volatile("" : "+v"(w));        // opaque: the product must be recomputed where it is used, not hoisted into nine more registers
    return w * m.kq;                   // the same IEEE product the host table holds
}
__device__ __forceinline__ void tw_set(double2 &dst, const double2 &src) { dst = src; }
__device__ __forceinline__ void tw_set(double &dst, const double2 &src) { dst = src.x; }

// the log4(D) in-place passes of the radix-4 forward transform on one lane's 4 values per row group (natural positions
// mm + (D/4)k in, bit-reversed-order positions 4mm..4mm+3 out, NOT yet centred).  NR independent row groups (a wave's 64
// lanes hold 64 / (D/4) polynomials per group) go through the passes in lock step: one wave-local synchronisation per
// pass whatever NR is, twiddles and LDS offsets computed once, and NR independent dependency chains for the fp64 pipeline.
// Row group r of this lane's polynomial lives at region + r * 256 doubles.
template <int LOGD, bool FAST, int NR, typename TW = double2, typename TWA = FzTwA>
__device__ __forceinline__ void fwd4_passes_n(double (&a)[NR][4], double *region, const TW (&twl)[LOGD / 2 - 1][3],
                                              const TWA &twA, const FzMod &m, int mm) {
    constexpr int D = 1 << LOGD, P = LOGD / 2;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int s = D >> (2 * i + 2);
        const int base = (mm / s) * 4 * s + mm % s;
        double wA, wA2, wB0, wB02, wB1, wB12;
        if (i == 0) {
            wA = twA.w[1]; wA2 = twA.w2[1]; wB0 = twA.w[2]; wB02 = twA.w2[2]; wB1 = twA.w[3]; wB12 = twA.w2[3];
        } else {
            wA = tw_w(twl[i - 1][0]); wA2 = tw_q(twl[i - 1][0], m);
            wB0 = tw_w(twl[i - 1][1]); wB02 = tw_q(twl[i - 1][1], m);
            wB1 = tw_w(twl[i - 1][2]); wB12 = tw_q(twl[i - 1][2], m);
            wave_sync();
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const double *reg = region + r * 256;
                if (s == 1) {
                    const double2 lo = *reinterpret_cast<const double2 *>(reg + swz4(base));
                    const double2 hi = *reinterpret_cast<const double2 *>(reg + swz4(base + 2));
                    a[r][0] = lo.x; a[r][1] = lo.y; a[r][2] = hi.x; a[r][3] = hi.y;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) a[r][k] = reg[swz4(base + k * s)];
                }
            }
        }
        // stage 2i: distance 2s, one twiddle; stage 2i+1: distance s, two twiddles
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            double v = tw_mul<FAST>(a[r][2], wA, wA2, m), u = a[r][0];
            a[r][0] = u + v; a[r][2] = u - v;
            v = tw_mul<FAST>(a[r][3], wA, wA2, m); u = a[r][1];
            a[r][1] = u + v; a[r][3] = u - v;
            v = tw_mul<FAST>(a[r][1], wB0, wB02, m); u = a[r][0];
            a[r][0] = u + v; a[r][1] = u - v;
            v = tw_mul<FAST>(a[r][3], wB1, wB12, m); u = a[r][2];
            a[r][2] = u + v; a[r][3] = u - v;
        }
        if (i < P - 1) {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                double *reg = region + r * 256;
#pragma unroll
                for (int k = 0; k < 4; ++k) reg[swz4(base + k * s)] = a[r][k];
            }
        }
    }
}

template <int LOGD, bool FAST>
__device__ __forceinline__ void fwd4_passes(double (&a)[4], double *region, const double2 (&twl)[LOGD / 2 - 1][3],
                                            const FzTwA &twA, const FzMod &m, int mm) {
    fwd4_passes_n<LOGD, FAST, 1>(reinterpret_cast<double (&)[1][4]>(a), region, twl, twA, m, mm);
}

template <int LOGD, typename TW = double2>
__device__ __forceinline__ void fwd4_load_twiddles(TW (&twl)[LOGD / 2 - 1][3], const double2 *__restrict__ tw2, int mm) {
    constexpr int D = 1 << LOGD, P = LOGD / 2;
#pragma unroll
    for (int i = 1; i < P; ++i) {
        const int s = D >> (2 * i + 2), g = mm / s, pw = 1 << (2 * i);
        if constexpr (__is_same(TW, double2)) {
            twl[i - 1][0] = tw2[pw + g];
            twl[i - 1][1] = tw2[2 * pw + 2 * g];
            twl[i - 1][2] = tw2[2 * pw + 2 * g + 1];
        } else {
            twl[i - 1][0] = tw2[pw + g].x;
            twl[i - 1][1] = tw2[2 * pw + 2 * g].x;
            twl[i - 1][2] = tw2[2 * pw + 2 * g + 1].x;
        }
    }
}

// One wave-task = NR row groups (NR * 64 / (D/4) consecutive polynomials), one task per wave, WAVES waves per workgroup,
// grid = tasks / WAVES: no persistent loop (a loop's bookkeeping -- 64-bit task arithmetic, the prefetch state, the
// conditional refill -- cost the one-row-per-wave kernel 7-9 % at the bench's 4096 rows: 4.74 -> 4.32 us cold).
// Measured on one box, forward, degree 256, cold operands (tools/microbench/ntt_variants.hip, profiles/r03_ntt_variants.txt):
//   4096 rows: NR = 1 4.32 us (the loop kernel 4.74; NR = 2 4.51; NR = 4 5.4 -- too few waves);
//   8192 rows: NR = 2 6.03 us (NR = 1 6.23-6.51; the loop kernel 7.09; the 16-per-lane kernel 6.52);
//   16384 rows: NR = 4 9.24 us (NR = 2 10.3; NR = 1 10.6; the loop kernel 11.3; 16-per-lane 9.40);
//   from 32768 rows the 16-per-lane kernel leads (15.0 us against 16.6).
// The waves of a workgroup never talk to each other (wave-private LDS regions, no s_barrier).
// one wave-task of the forward transform: NR row groups starting at polynomial poly0 (this lane's polynomial of group 0)
template <int LOGD, bool FAST, int NR>
__device__ __forceinline__ void fwd4_task(const int32_t *in, int32_t *out, size_t batch, size_t poly0, double *region, int mm,
                                          const double2 *__restrict__ tw2, const FzTw4 &twA, const FzMod &m) {
    constexpr int D = 1 << LOGD, LP = D / 4, PPW = 64 / LP, P = LOGD / 2;
    int x[NR][4];                                 // the data loads first: they have the longest way to go
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const size_t poly = poly0 + (size_t)r * PPW;
        const int32_t *src = in + (poly < batch ? poly : batch - 1) * D + mm;
#pragma unroll
        for (int k = 0; k < 4; ++k) x[r][k] = src[k * LP];
    }
    double2 twl[P - 1][3];
    fwd4_load_twiddles<LOGD>(twl, tw2, mm);
    double a[NR][4];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) a[r][k] = (double)x[r][k];
    fwd4_passes_n<LOGD, FAST, NR>(a, region, twl, twA, m, mm);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const size_t poly = poly0 + (size_t)r * PPW;
        if (poly < batch)
            nt_store4(out + poly * D + 4 * mm, make_int4((int)fz_cent(a[r][0], m), (int)fz_cent(a[r][1], m), (int)fz_cent(a[r][2], m),
                                                          (int)fz_cent(a[r][3], m)));
    }
}

template <int LOGD, bool FAST, int NR, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void ntt_fwd4(const int32_t *in, int32_t *out, size_t batch,
                                                       const double2 *__restrict__ tw2, FzTw4 twA, FzMod m) {
    constexpr int D = 1 << LOGD, LP = D / 4, PPW = 64 / LP;
    static_assert(LOGD % 2 == 0 && LOGD >= 6 && LOGD <= 8, "radix-4 kernel: degree 64 or 256");
    __shared__ __attribute__((aligned(16))) double lds[WAVES * NR * 256];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;      // the wave index is uniform: say so (scalar address arithmetic)
    const int p = lane / LP, mm = lane % LP;
    double *region = lds + wave * NR * 256 + p * D;
    const size_t task = (size_t)blockIdx.x * WAVES + wave;
    if (task * (NR * PPW) >= batch) return;
    fwd4_task<LOGD, FAST, NR>(in, out, batch, task * (NR * PPW) + p, region, mm, tw2, twA, m);      // row group r: polynomial poly0 + r * PPW
}

// the log4(D) in-place passes of the radix-4 inverse on one lane's 4 values per row group (bit-reversed positions
// 4mm..4mm+3 in, natural positions mm + (D/4)k out, n^-1 applied, NOT yet centred); NR row groups in lock step (see
// fwd4_passes_n)
template <int LOGD, bool FAST, int NR, typename TW = double2, typename TWA = FzTwA>
__device__ __forceinline__ void inv4_passes_n(double (&a)[NR][4], double *region, const TW (&twl)[LOGD / 2 - 1][3],
                                              const TWA &twA, const FzMod &m, int mm) {
    constexpr int P = LOGD / 2;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int s = 1 << (2 * i);
        const int base = (mm / s) * 4 * s + mm % s;
        if (i > 0) {
            wave_sync();
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const double *reg = region + r * 256;
#pragma unroll
                for (int k = 0; k < 4; ++k) a[r][k] = reg[swz4(base + k * s)];
            }
        }
        if (i < P - 1) {
            // GS stage 2i (distance s, two twiddles) then stage 2i+1 (distance 2s, one twiddle);
            // operands stay below 2^(33+2i+1) <= 2^38
            const double w0 = tw_w(twl[i][0]), q0 = tw_q(twl[i][0], m), w1 = tw_w(twl[i][1]), q1 = tw_q(twl[i][1], m),
                         w2_ = tw_w(twl[i][2]), q2 = tw_q(twl[i][2], m);
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                double u = a[r][0], v = a[r][1];
                a[r][0] = u + v; a[r][1] = tw_mul<FAST>(u - v, w0, q0, m);
                u = a[r][2]; v = a[r][3];
                a[r][2] = u + v; a[r][3] = tw_mul<FAST>(u - v, w1, q1, m);
                u = a[r][0]; v = a[r][2];
                a[r][0] = u + v; a[r][2] = tw_mul<FAST>(u - v, w2_, q2, m);
                u = a[r][1]; v = a[r][3];
                a[r][1] = u + v; a[r][3] = tw_mul<FAST>(u - v, w2_, q2, m);
                // Degree 256 with raw int32 inputs: a[0] is the only value no multiply has reduced (the sum of four inputs, up
                // to 2^33; a[1] <= 2^31.1, a[2], a[3] <= 2^30.1).  Folding it once (2 ops) keeps every later operand below
                // 2^31.1 * 2^6 = 2^37.1, inside the 4-op multiply's 2^38 bound up to and including the final stage -- which
                // otherwise needs the general 6-op form four times (8 extra ops per lane).
                if (FAST && i == 0 && 31 + LOGD > 38) a[r][0] = fz_fold(a[r][0], m);
                double *reg = region + r * 256;
                if (s == 1) {
                    *reinterpret_cast<double2 *>(reg + swz4(base)) = make_double2(a[r][0], a[r][1]);
                    *reinterpret_cast<double2 *>(reg + swz4(base + 2)) = make_double2(a[r][2], a[r][3]);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) reg[swz4(base + k * s)] = a[r][k];
                }
            }
        } else {
            // last pass: uniform twiddles itw[2], itw[3], itw[1]; n^-1 folded into the final stage.
            // Its operands are below 2^38 for raw int32 inputs: 2^(31+LOGD) up to degree 128, 2^37.1 at degree 256
            // thanks to the fold after pass 0 -- so the 4-op multiply serves whenever the modulus admits it.
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                double u = a[r][0], v = a[r][1];
                a[r][0] = u + v; a[r][1] = tw_mul<FAST>(u - v, twA.w[2], twA.w2[2], m);
                u = a[r][2]; v = a[r][3];
                a[r][2] = u + v; a[r][3] = tw_mul<FAST>(u - v, twA.w[3], twA.w2[3], m);
                u = a[r][0]; v = a[r][2];
                a[r][0] = tw_mul<FAST>(u + v, twA.n_inv, twA.n_inv2, m);
                a[r][2] = tw_mul<FAST>(u - v, twA.w1_n_inv, twA.w1_n_inv2, m);
                u = a[r][1]; v = a[r][3];
                a[r][1] = tw_mul<FAST>(u + v, twA.n_inv, twA.n_inv2, m);
                a[r][3] = tw_mul<FAST>(u - v, twA.w1_n_inv, twA.w1_n_inv2, m);
            }
        }
    }
}

template <int LOGD, bool FAST>
__device__ __forceinline__ void inv4_passes(double (&a)[4], double *region, const double2 (&twl)[LOGD / 2 - 1][3],
                                            const FzTwA &twA, const FzMod &m, int mm) {
    inv4_passes_n<LOGD, FAST, 1>(reinterpret_cast<double (&)[1][4]>(a), region, twl, twA, m, mm);
}

template <int LOGD, typename TW = double2>
__device__ __forceinline__ void inv4_load_twiddles(TW (&twl)[LOGD / 2 - 1][3], const double2 *__restrict__ itw2, int mm) {
    constexpr int D = 1 << LOGD, P = LOGD / 2;
#pragma unroll
    for (int i = 0; i < P - 1; ++i) {
        const int s = 1 << (2 * i), g = mm / s;
        tw_set(twl[i][0], itw2[D / (2 * s) + 2 * g]);
        tw_set(twl[i][1], itw2[D / (2 * s) + 2 * g + 1]);
        tw_set(twl[i][2], itw2[D / (4 * s) + g]);
    }
}

template <int LOGD, bool FAST, int NR>
__device__ __forceinline__ void inv4_task(const int32_t *in, int32_t *out, size_t batch, size_t poly0, double *region, int mm,
                                          const double2 *__restrict__ itw2, const FzTw4 &twA, const FzMod &m) {
    constexpr int D = 1 << LOGD, LP = D / 4, PPW = 64 / LP, P = LOGD / 2;
    int4 x[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const size_t poly = poly0 + (size_t)r * PPW;
        x[r] = *reinterpret_cast<const int4 *>(in + (poly < batch ? poly : batch - 1) * D + 4 * mm);
    }
    double2 twl[P - 1][3];
    inv4_load_twiddles<LOGD>(twl, itw2, mm);
    double a[NR][4];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        a[r][0] = (double)x[r].x; a[r][1] = (double)x[r].y; a[r][2] = (double)x[r].z; a[r][3] = (double)x[r].w;
    }
    inv4_passes_n<LOGD, FAST, NR>(a, region, twl, twA, m, mm);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const size_t poly = poly0 + (size_t)r * PPW;
        if (poly < batch) {
            int32_t *dst = out + poly * D + mm;
#pragma unroll
            for (int k = 0; k < 4; ++k) __builtin_nontemporal_store((int)fz_cent(a[r][k], m), dst + k * LP);
        }
    }
}

template <int LOGD, bool FAST, int NR, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void ntt_inv4(const int32_t *in, int32_t *out, size_t batch,
                                                       const double2 *__restrict__ itw2, FzTw4 twA, FzMod m) {
    constexpr int D = 1 << LOGD, LP = D / 4, PPW = 64 / LP;
    static_assert(LOGD % 2 == 0 && LOGD >= 6 && LOGD <= 8, "radix-4 kernel: degree 64 or 256");
    __shared__ __attribute__((aligned(16))) double lds[WAVES * NR * 256];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;      // the wave index is uniform: say so (scalar address arithmetic)
    const int p = lane / LP, mm = lane % LP;
    double *region = lds + wave * NR * 256 + p * D;
    const size_t task = (size_t)blockIdx.x * WAVES + wave;      // one task per wave: see ntt_fwd4
    if (task * (NR * PPW) >= batch) return;
    inv4_task<LOGD, FAST, NR>(in, out, batch, task * (NR * PPW) + p, region, mm, itw2, twA, m);
}

// ------------------------------------------------------------------------------------------
// Many independent transforms in ONE dispatch (fz_ntt_multi).  The reference issues its transforms one polynomial at a
// time (fusion/fusion.py:363-368: 2*rank per key; :690-692: rank per verification), and a launch of a few thousand rows
// sits on the dispatch floor (a 4096-row launch costs ~4 us of which ~2 are the empty dispatch).  Here a job table --
// (in, out, rows, direction) per job, in the kernarg segment -- is served by one grid of the SAME wave-tasks the one-job
// kernels above run: every job owns a run of workgroups (J.end[j] = workgroups up to and including job j), a workgroup finds
// its job by a scalar scan of at most 32 entries and then is a workgroup of ntt_fwd4 or ntt_inv4 (the direction is uniform
// per workgroup: one twiddle set in registers, as there).  Round 4: this replaced the persistent one-wave-per-workgroup
// kernel of rounds 2-3 (ntt_multi4), which cost 6.9 us for two jobs of 4096 rows where ONE job of 8192 rows costs 5.55
// (profiles/r03_ntt_small_batches.txt) -- a forward job and an inverse job of 4096 rows in one launch is the software-pipelined
// form of BASELINE configs[1]'s step (forward of batch i+1 beside the inverse of batch i), so that gap was the headline's.
// ------------------------------------------------------------------------------------------
// which job a workgroup belongs to: first / last = the run of workgroups [first, last) the job owns, rw = its rows (bit 31:
// inverse), in / out its buffers
template <typename JT>
__device__ __forceinline__ void pick_job(const JT &J, const unsigned b, unsigned &first, unsigned &last, unsigned &rw,
                                         const int32_t *&in, int32_t *&out) {
    constexpr int NJ = JT::kJobs;
    first = 0;
    if constexpr (NJ <= 8) {
        // The job WITHOUT a dependent load: every table entry is requested together with everything else the workgroup
        // reads from the kernel arguments and the job is picked by scalar selects (a scan would be a chain of scalar-load
        // round trips in front of the first data load: one job through a scanning kernel cost 5.0 us against 4.2 for
        // ntt_fwd4; round 4 did this for four entries and scanned from the fifth on, round 5's headline launch has eight).
        // Entries past the last job hold the launch's total, so they are never chosen.
        unsigned e[NJ], r[NJ];
        const int32_t *ip[NJ];
        int32_t *op[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            e[j] = J.end[j]; r[j] = J.rows[j]; ip[j] = J.in[j]; op[j] = J.out[j];
            // (opaque to the optimiser: it would otherwise turn the selects below into branches that load only the chosen
            // entry -- the dependent load this path exists to avoid)
            asm volatile("" : "+s"(e[j]), "+s"(r[j]), "+s"(ip[j]), "+s"(op[j]));
        }
        rw = r[0]; in = ip[0]; out = op[0]; last = e[0];
#pragma unroll
        for (int j = 1; j < NJ; ++j) {
            const bool g = b >= e[j - 1];                               // (ends never decrease: the last true one wins)
            first = g ? e[j - 1] : first;
            last = g ? e[j] : last;
            rw = g ? r[j] : rw;
            in = g ? ip[j] : in;
            out = g ? op[j] : out;
        }
    } else {
        // Larger tables would not fit the scalar registers (32 entries = 192 of them): the workgroup counts of ALL jobs come
        // with the first request (32 dwords), the job's index is found by scalar compares, and its three other entries are
        // ONE dependent round of scalar loads -- instead of a scan's one round trip per job passed.
        unsigned mask = 0;
#pragma unroll
        for (int k = 0; k < NJ - 1; ++k) {
            const unsigned ek = J.end[k];
            const bool g = b >= ek;
            first = g ? ek : first;                                     // (ends never decrease: the last true one wins)
            mask |= g ? (1u << k) : 0u;
        }
        const unsigned j = __builtin_amdgcn_readfirstlane(__builtin_popcount(mask));
        rw = J.rows[j]; in = J.in[j]; out = J.out[j]; last = J.end[j];
    }
}

// JT: FzJobsN<4 | 8 | 32> -- the table of a launch of at most that many jobs (24 bytes of kernel arguments per entry).
// `stamp` (diagnostics, NULL otherwise): one {entry, exit} pair of the 100 MHz reference counter per WORKGROUP, written by
// the workgroup's first wave after its stores have left (fz_diag_stamps_*): the chip's own record of when a launch ran,
// which no profiler serialises.
template <int LOGD, bool FAST, int NR, int WAVES, typename JT>
__global__ __launch_bounds__(64 * WAVES) void ntt_jobs4(JT J, const double2 *__restrict__ tw2, const double2 *__restrict__ itw2,
                                                        FzTw4 twA, FzTw4 itwA, FzMod m, unsigned long long *stamp) {
    constexpr int D = 1 << LOGD, LP = D / 4, PPW = 64 / LP, NJ = JT::kJobs;
    __shared__ __attribute__((aligned(16))) double lds[WAVES * NR * 256];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int p = lane / LP, mm = lane % LP;
    double *region = lds + wave * NR * 256 + p * D;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // (unconditional: a branch on `stamp` here would put a scalar-load round trip in front of everything else)
    unsigned first, last, rw;
    const int32_t *in;
    int32_t *out;
    pick_job(J, blockIdx.x, first, last, rw, in, out);
    const size_t rows = rw & 0x7fffffffu;
    const bool inverse = (rw >> 31) != 0;
    const size_t task = (size_t)(blockIdx.x - first) * WAVES + wave;
    if (task * (NR * PPW) >= rows) return;                              // (never a workgroup's first wave: the grid is cut to whole tasks)
    if (inverse) inv4_task<LOGD, FAST, NR>(in, out, rows, task * (NR * PPW) + p, region, mm, itw2, itwA, m);
    else fwd4_task<LOGD, FAST, NR>(in, out, rows, task * (NR * PPW) + p, region, mm, tw2, twA, m);
    if (stamp && wave == 0) {
        __builtin_amdgcn_s_waitcnt(0);                                  // this wave's stores have been acknowledged
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            stamp[2 * (size_t)blockIdx.x] = t0;
            stamp[2 * (size_t)blockIdx.x + 1] = t1;
        }
    }
}

// The same job table served by the 16-per-lane kernels (round 5): a launch of 24 576 rows and more (fz_ctx::small_batch_rows)
// -- the headline's sixteen forward + sixteen inverse batches of 4096 -- is past the point where the radix-4 wave-tasks lead
// (one job: 25.5 us against 30 us at 65 536 rows; eight jobs of 4096: 15.9 us against 17.2).  A job owns a run of workgroups;
// the run IS a grid of ntt_fwd16 or ntt_inv16 over the job's batch (resident workgroups striding over 4 KiB chunks, next
// chunk prefetched into registers), sized by the launcher in proportion to the job's share of the launch.
template <int LOGD, bool FAST, typename JT>
__global__ __launch_bounds__(64 * kWavesPerBlock) void ntt_jobs16(JT J, const double2 *__restrict__ twB, const double2 *__restrict__ itwB,
                                                                  FzTwA twA, FzTwA itwA, FzMod m, unsigned long long *stamp) {
    __shared__ __attribute__((aligned(16))) double lds[lds16_doubles<LOGD>()];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned first, last, rw;
    const int32_t *in;
    int32_t *out;
    pick_job(J, blockIdx.x, first, last, rw, in, out);
    const size_t rows = rw & 0x7fffffffu;
    if ((rw >> 31) != 0) inv16_run<LOGD, FAST>(in, out, rows, blockIdx.x - first, last - first, lds, itwB, itwA, m);
    else fwd16_run<LOGD, FAST>(in, out, rows, blockIdx.x - first, last - first, lds, twB, twA, m);
    if (stamp && threadIdx.x < 64) {                                    // fz_diag_stamps_*: see ntt_jobs4
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0) {
            stamp[2 * (size_t)blockIdx.x] = t0;
            stamp[2 * (size_t)blockIdx.x + 1] = t1;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Negacyclic product INTT(NTT(f) * NTT(g)) in one launch (algebra/ntt.py:380-484 ntt_poly_mult; the product the
// reference's schoolbook PolynomialCoefficientRepresentation.__mul__, polynomials.py:171-216, is tested against):
// both forward transforms, the pointwise product and the inverse stay in registers / LDS; HBM sees 12*D bytes per
// product (f, g in; f*g out) instead of the 36*D of three transform launches plus a pointwise one.  Radix-4 layout:
// the forward passes leave a lane's values at bit-reversed positions 4mm..4mm+3, exactly where the inverse picks up.
// `out` may alias `f` or `g` (a wave has read its whole polynomials before it writes).
// ------------------------------------------------------------------------------------------
template <int LOGD, bool FAST>
__global__ __launch_bounds__(64 * kWavesPerBlock) void polymul_fused(const int32_t *f, const int32_t *g, int32_t *out, size_t batch,
                                                                     const double2 *__restrict__ tw2,
                                                                     const double2 *__restrict__ itw2, FzTwA twA, FzTwA itwA,
                                                                     FzMod m) {
    using TW = double2;
    constexpr int D = 1 << LOGD, LP = D / 4, PPW = 64 / LP, P = LOGD / 2;
    static_assert(LOGD % 2 == 0 && LOGD >= 6 && LOGD <= 8, "radix-4 kernel: degree 64 or 256");
    __shared__ __attribute__((aligned(16))) double lds[kWavesPerBlock * 256];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;      // the wave index is uniform: say so (scalar address arithmetic)
    const int p = lane / LP, mm = lane % LP;
    double *region = lds + wave * 256 + p * D;
    const size_t tasks = (batch + PPW - 1) / PPW;
    const size_t first = (size_t)blockIdx.x * kWavesPerBlock + wave, stride = (size_t)gridDim.x * kWavesPerBlock;
    if (first >= tasks) return;

    TW twf[P - 1][3], twi[P - 1][3];
    fwd4_load_twiddles<LOGD, TW>(twf, tw2, mm);
    inv4_load_twiddles<LOGD, TW>(twi, itw2, mm);

    for (size_t task = first; task < tasks; task += stride) {
        const size_t poly = task * PPW + p;
        const bool valid = poly < batch;
        const size_t row = (valid ? poly : batch - 1) * D + mm;
        int xf[4], xg[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) xf[k] = f[row + k * LP];
#pragma unroll
        for (int k = 0; k < 4; ++k) xg[k] = g[row + k * LP];
        double a[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = (double)xf[k];
        fwd4_passes_n<LOGD, FAST, 1, TW>(reinterpret_cast<double (&)[1][4]>(a), region, twf, twA, m, mm);
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = fz_cent(a[k], m);         // one centred factor keeps the product below 2^66
        wave_sync();                                                 // g's first-pass writes vs f's last-pass reads
#pragma unroll
        for (int k = 0; k < 4; ++k) b[k] = (double)xg[k];
        fwd4_passes_n<LOGD, FAST, 1, TW>(reinterpret_cast<double (&)[1][4]>(b), region, twf, twA, m, mm);
#pragma unroll
        for (int k = 0; k < 4; ++k) b[k] = fz_mulmod(a[k], b[k], m);
        wave_sync();
        inv4_passes_n<LOGD, FAST, 1, TW>(reinterpret_cast<double (&)[1][4]>(b), region, twi, itwA, m, mm);
        if (valid) {
            int32_t *dst = out + poly * D + mm;
#pragma unroll
            for (int k = 0; k < 4; ++k) __builtin_nontemporal_store((int)fz_cent(b[k], m), dst + k * LP);
        }
        wave_sync();      // the next product's first-pass writes must not overtake this one's last reads
    }
}

// ------------------------------------------------------------------------------------------
// The same product on the 16-per-lane transforms (32 <= D <= 256): ONE exchange through LDS per transform instead of the three
// of the radix-4 passes (polymul_fused above spends half its LDS pipe and a fifth of its cycles waiting on them,
// profiles/r06_shape_ceilings.txt), global traffic as 16 bytes per lane like ntt_fwd16 / ntt_inv16.  For batches that give
// every SIMD a few of these 128-register waves; smaller ones stay with the radix-4 kernel (fz_launch_polymul_fused chooses).
// A wave-task is one 4 KiB chunk of f, of g and of the product (PPW polynomials).  Pipeline: f's next chunk is requested at
// the top of an iteration and g's next chunk once g's current image has left the registers, so at most two chunks are held
// in registers; f waits in the staging image, g in registers; the stores are the youngest operations (see fwd16_run).
// `out` may alias `f` or `g`: a wave reads chunk t of both before it writes chunk t, and no other wave touches chunk t.
// ------------------------------------------------------------------------------------------
template <int LOGD> constexpr int lds_pm16_doubles() {
    using G = Geom<LOGD>;
    return kWavesPerBlock * G::PPW * G::PS + 4 * G::NE * G::L;      // a transpose region per wave + both per-lane twiddle tables
}

template <int LOGD, bool FAST>
__global__ __launch_bounds__(64 * kWavesPerBlock, 3) void polymul16(const int32_t *f, const int32_t *g, int32_t *out, size_t batch,
                                                                 const double2 *__restrict__ twB, const double2 *__restrict__ itwB,
                                                                 const FzTwA *tabs, FzMod m) {
    using G = Geom<LOGD>;
    constexpr int D = G::D, L = G::L, PPW = G::PPW, NE = G::NE, PS = G::PS;
    constexpr int REGION = PPW * PS;
    // The wave-uniform tables of both directions are 2 x 60 scalar registers where 102 exist: as kernel arguments they are loaded
    // once and then spilled into vector lanes (180 v_readlane per iteration).  They are read from constant memory instead, each
    // direction where it is used: the empty asm makes the pointer opaque, so the loads cannot be hoisted back out of the loop.
    typedef const __attribute__((address_space(4))) FzTwA *TabPtr;
    __shared__ __attribute__((aligned(16))) double lds[lds_pm16_doubles<LOGD>()];
    double2 *s_tw = reinterpret_cast<double2 *>(lds + kWavesPerBlock * REGION), *s_itw = s_tw + NE * L;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int p = lane / L, r = lane % L;
    const size_t total = batch * D;
    const size_t tasks = (total + kChunk - 1) / kChunk;
    const size_t first = (size_t)blockIdx.x * kWavesPerBlock + wave;
    const size_t stride = (size_t)gridDim.x * kWavesPerBlock;
    Chunk rawF = {}, rawG = {};
    if (first < tasks) {                                  // before the tables: see fwd16_run
        rawF = chunk_load(f, first, total, lane);
        rawG = chunk_load(g, first, total, lane);
    }
    for (int i = threadIdx.x; i < NE * L; i += 64 * kWavesPerBlock) {
        s_tw[i] = twB[i];
        s_itw[i] = itwB[i];
    }
    __syncthreads();                                      // the only workgroup-wide barrier
    double *region = lds + wave * REGION;
    int32_t *stage = reinterpret_cast<int32_t *>(region);
    double *row = region + p * PS;
    if (first >= tasks) return;
    chunk_to_lds(stage, lane, rawF);

    // element r + L*k of the lane's polynomial in the staging image: pad4(p * D + r + L * k) = pad4(p * D) + r + pad4(L * k), because
    // r < L and L divides 16 -- one address register and sixteen constant offsets instead of sixteen registers
    int32_t *strided = stage + pad4(p * D) + r;
    auto strided_from_stage = [&](double (&a)[16]) __attribute__((always_inline)) {
        int x[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = strided[pad4(L * k)];
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = (double)x[k];
    };
    auto iteration = [&](const size_t task, auto more_tag) __attribute__((always_inline)) {       // peeling: see fwd16_run
        constexpr bool more = decltype(more_tag)::value;
        if (more) rawF = chunk_load(f, task + stride, total, lane);
        wave_sync();
        double b[16];
        int fa[16];                                       // NTT(f), centred: 16 registers while g is transformed, not 32
        TabPtr tf = (TabPtr)tabs;
        asm volatile("" : "+s"(tf));
        strided_from_stage(b);
        wave_sync();
        fwd16_passes<LOGD, FAST>(b, row, r, s_tw, tf[0], m);
#pragma unroll
        for (int k = 0; k < 16; ++k) fa[k] = (int)fz_cent(b[k], m);
        chunk_to_lds(stage, lane, rawG);                               // waits for g's chunk; f's next one is younger
        if (more) rawG = chunk_load(g, task + stride, total, lane);
        wave_sync();
        strided_from_stage(b);
        wave_sync();
        fwd16_passes<LOGD, FAST>(b, row, r, s_tw, tf[0], m);
        TabPtr ti = (TabPtr)tabs + 1;
        asm volatile("" : "+s"(ti));
#pragma unroll
        for (int k = 0; k < 16; ++k) b[k] = fz_mulmod(b[k], (double)fa[k], m);      // |b * fa| < 2^69; |result| <= q/2 + 1: an input the inverse accepts
        inv16_passes<LOGD, FAST>(b, row, r, s_itw, ti[0], m);
#pragma unroll
        for (int k = 0; k < 16; ++k) strided[pad4(L * k)] = (int)fz_cent(b[k], m);
        wave_sync();
        const int4 o0 = *reinterpret_cast<const int4 *>(stage + pad4(4 * lane));
        const int4 o1 = *reinterpret_cast<const int4 *>(stage + pad4(256 + 4 * lane));
        const int4 o2 = *reinterpret_cast<const int4 *>(stage + pad4(512 + 4 * lane));
        const int4 o3 = *reinterpret_cast<const int4 *>(stage + pad4(768 + 4 * lane));
        wave_sync();
        if (more) chunk_to_lds(stage, lane, rawF);        // waits for f's next chunk; g's next one and the stores are younger
        chunk_store(out, task, total, lane, o0, o1, o2, o3);
    };
    size_t task = first;
    for (; task + stride < tasks; task += stride) iteration(task, std::true_type());
    iteration(task, std::false_type());
}

// ------------------------------------------------------------------------------------------
// Fused keygen arithmetic (fusion/fusion.py:363-370), one workgroup per (key, half): every secret row is
// transformed (radix-4 forward), written to sk_hat, and -- while still in registers -- multiplied by the
// matching row of the public challenge A and accumulated; the l partial products are reduced through LDS
// into the verification-key row.  sk_hat is never re-read: 342 KB of HBM traffic per key instead of 508 KB.
// ------------------------------------------------------------------------------------------
// IMAD: A[k] (.) y accumulates in 64-bit INTEGERS: A = hi * 2^16 + lo (hi = A >> 16, lo = A & 0xffff, two integer ops on the
// int32 row as it is loaded), then acc_hi += y * hi and acc_lo += y * lo are one v_mad_i64_i32 each -- |y * hi|, |y * lo| < 2^47,
// so 2^15 rows sum without overflow (fz_arith.h; the launcher falls back to the fp64 form beyond) and nothing is reduced inside the loop: 4 operations per coefficient instead of 8
// (conversion of A, the 6-op FMA-Barrett multiply, the accumulate).  The integer form of y is the value keygen stores anyway.
// (Measured and dropped: A pre-split into fp64 (hi, lo) pairs by the host -- two FMAs per coefficient, but 64 bytes of L2
// traffic per lane and row instead of 16: keygen 79 -> 109 us per 1024 keys, verify 256 -> 270 us per 8192 aggregates,
// profiles/r03_presplit_A_experiment.txt.)
// (fz_imad_total, the sums' way back to fp64, lives in fz_arith.h.)
// One row group per wave iteration, rows requested one iteration ahead, per-lane twiddles as (w, w * K/q) pairs: round 3 measured
// two row groups, a second iteration of prefetch and twiddles kept as w alone (five waves per SIMD) -- 81.7 / 81.7 / 82.0 / 81.1
// and 79.2 / 77.7 us per 1024 keys, all within 2 % (profiles/r03_keygen_ab.txt) -- and round 4 removed those instantiations.
// Round 5: the secret rows (read once) by streaming loads, sk_hat (never read here) by streaming stores: 81.3 -> 77.6 us alone,
// keygen + sign chained 119.4 -> 112.1 us per 1024 keys.  The `if`s around the next rows' request and around the store make the
// compiler's wait before the store a wait for ALL outstanding operations (one in-order counter); the form with exact wait
// counts (everything unconditional, rows clamped) is 4-5 % faster alone and 2-6 % SLOWER between two sign launches, the
// scheme's order -- measured on three boxes and dropped (profiles/r05_keygen_exact_waits_experiment.txt).
template <int LOGD, bool FAST, bool IMAD>
__global__ __launch_bounds__(64 * kWavesPerBlock) __attribute__((amdgpu_waves_per_eu(4, 6))) void keygen_fused(const int32_t *A, const int32_t *coef,
                                                                    size_t coef_seg_stride,
                                                                    size_t coef_row_stride, int32_t *sk_hat,
                                                                    int32_t *vk, int l, const double2 *__restrict__ tw2,
                                                                    FzTwA twA, FzMod m) {
    constexpr int NR = 1, PF = 1;
    using TW = double2;
    constexpr int D = 1 << LOGD, LP = D / 4, PPW = 64 / LP;
    __shared__ __attribute__((aligned(16))) double lds[kWavesPerBlock * 256 * (NR + 1)];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;      // the wave index is uniform: say so (scalar address arithmetic)
    const int p = lane / LP, mm = lane % LP;
    double *region = lds + wave * NR * 256 + p * D;
    double *accbuf = lds + kWavesPerBlock * NR * 256;
    const size_t seg = blockIdx.x;                      // (key, half)
    coef += seg * coef_seg_stride;                      // row stride 0: one secret polynomial per (key, half), as the
    sk_hat += seg * (size_t)l * D;                      // reference's seeded sampler produces (polynomials.py:436-467)

    TW twl[LOGD / 2 - 1][3];
    fwd4_load_twiddles<LOGD, TW>(twl, tw2, mm);

    double acc[4] = {0, 0, 0, 0};
    long long ihi[4] = {0, 0, 0, 0}, ilo[4] = {0, 0, 0, 0};      // IMAD: exact integer sums of y * hi and y * lo
    const int tasks = (l + PPW - 1) / PPW;
    constexpr int STEP = kWavesPerBlock * NR;           // a wave's iteration covers tasks t, t + 4, .. (NR of them)
    int xq[PF][NR][4];                                  // the next PF iterations' rows, in flight
    auto fetch = [&](int (&x)[NR][4], int task) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int row = (task + r * kWavesPerBlock) * PPW + p;
            const int32_t *src = coef + (size_t)(row < l ? row : l - 1) * coef_row_stride + mm;
#pragma unroll
            for (int k = 0; k < 4; ++k) x[r][k] = __builtin_nontemporal_load(src + k * LP);
        }
    };
#pragma unroll
    for (int h = 0; h < PF; ++h)
        if (wave + h * STEP < tasks) fetch(xq[h], wave + h * STEP);
    for (int task0 = wave; task0 < tasks; task0 += PF * STEP) {
#pragma unroll
        for (int h = 0; h < PF; ++h) {
            const int task = task0 + h * STEP;
            if (task >= tasks) break;
            double a[NR][4];
            int4 ak[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
#pragma unroll
                for (int k = 0; k < 4; ++k) a[r][k] = (double)xq[h][r][k];
                const int row = (task + r * kWavesPerBlock) * PPW + p;
                ak[r] = *reinterpret_cast<const int4 *>(A + (size_t)(row < l ? row : l - 1) * D + 4 * mm);
            }
            if (task + PF * STEP < tasks) fetch(xq[h], task + PF * STEP);
            fwd4_passes_n<LOGD, FAST, NR, TW>(a, region, twl, twA, m, mm);
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int row = (task + r * kWavesPerBlock) * PPW + p;
                const double y0 = fz_cent(a[r][0], m), y1 = fz_cent(a[r][1], m), y2 = fz_cent(a[r][2], m), y3 = fz_cent(a[r][3], m);
                if (row < l) {
                    const int4 yi = make_int4((int)y0, (int)y1, (int)y2, (int)y3);
                    nt_store4(sk_hat + (size_t)row * D + 4 * mm, yi);
                    if constexpr (IMAD) {
                        const int yv[4] = {yi.x, yi.y, yi.z, yi.w}, av[4] = {ak[r].x, ak[r].y, ak[r].z, ak[r].w};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            ihi[k] += (long long)yv[k] * (long long)(av[k] >> 16);
                            ilo[k] += (long long)yv[k] * (long long)(av[k] & 0xffff);
                        }
                    } else {
                        acc[0] += fz_mulmod(y0, (double)ak[r].x, m);
                        acc[1] += fz_mulmod(y1, (double)ak[r].y, m);
                        acc[2] += fz_mulmod(y2, (double)ak[r].z, m);
                        acc[3] += fz_mulmod(y3, (double)ak[r].w, m);
                    }
                }
            }
            wave_sync();
        }
    }
    if constexpr (IMAD) {
        const bool small = tasks <= 32 * kWavesPerBlock;          // rows per wave <= 32
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = fz_imad_total(ihi[k], ilo[k], small, m);
    }
    double *mine = accbuf + wave * 256 + p * D + 4 * mm;
    mine[0] = acc[0]; mine[1] = acc[1]; mine[2] = acc[2]; mine[3] = acc[3];
    __syncthreads();
    if (threadIdx.x < D) {
        double sum = 0;
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; ++w)
#pragma unroll
            for (int q = 0; q < PPW; ++q) sum += accbuf[w * 256 + q * D + threadIdx.x];
        vk[seg * D + threadIdx.x] = (int)fz_cent(sum, m);
    }
}

// The reference's SEEDED keygen samples every entry of a key half with the same seed (fusion.py:156-173): the l rows of a half
// are one polynomial, so their transforms are one transform.  fz_keygen_core_bcast (one polynomial per (key, half)) therefore
// transforms it ONCE per workgroup -- every wave for itself: a transform is cheaper than an exchange -- and the rest is the l
// stores of the row and the accumulation of A_k (.) y over k: a streaming kernel (85 KiB written per half, A from the L2)
// instead of l transforms.  Same results as the three launches other degrees take (rows expanded, transformed, multiplied by
// A: FZ_UNFUSED=1 runs them at these degrees too; tests/test_gpu_variants.py compares both with the oracle).
template <int LOGD, bool FAST>
__global__ __launch_bounds__(64 * kWavesPerBlock) void keygen_bcast_fused(const int32_t *A, const int32_t *coef, int32_t *sk_hat,
                                                                          int32_t *vk, int l, const double2 *__restrict__ tw2,
                                                                          FzTwA twA, FzMod m) {
    constexpr int D = 1 << LOGD, LP = D / 4, PPW = 64 / LP;
    __shared__ __attribute__((aligned(16))) double lds[kWavesPerBlock * 256 * 2];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int p = lane / LP, mm = lane % LP;
    double *region = lds + wave * 256 + p * D;
    double *accbuf = lds + kWavesPerBlock * 256;
    const size_t seg = blockIdx.x;                      // (key, half)
    coef += seg * (size_t)D;
    sk_hat += seg * (size_t)l * D;
    double2 twl[LOGD / 2 - 1][3];
    fwd4_load_twiddles<LOGD, double2>(twl, tw2, mm);
    double a[1][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[0][k] = (double)coef[mm + k * LP];
    fwd4_passes_n<LOGD, FAST, 1, double2>(a, region, twl, twA, m, mm);
    const int4 yi = make_int4((int)fz_cent(a[0][0], m), (int)fz_cent(a[0][1], m), (int)fz_cent(a[0][2], m), (int)fz_cent(a[0][3], m));
    const int yv[4] = {yi.x, yi.y, yi.z, yi.w};
    // sum_k A_k (.) y = (sum_k A_k) (.) y: the rows of A this lane's row slots cover, summed in integers (|A| <= 2^31, l <= 2^31
    // rows: no overflow of int64), one multiply at the end
    long long asum[4] = {0, 0, 0, 0};
    constexpr int U = 4;
    const int step = kWavesPerBlock * PPW;
    for (int row0 = wave * PPW + p; row0 < l; row0 += U * step) {
        int4 ak[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int row = row0 + u * step;
            ak[u] = *reinterpret_cast<const int4 *>(A + (size_t)(row < l ? row : l - 1) * D + 4 * mm);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int row = row0 + u * step;
            if (row < l) {
                nt_store4(sk_hat + (size_t)row * D + 4 * mm, yi);
                asum[0] += ak[u].x; asum[1] += ak[u].y; asum[2] += ak[u].z; asum[3] += ak[u].w;
            }
        }
    }
    double *mine = accbuf + wave * 256 + p * D + 4 * mm;
#pragma unroll
    for (int k = 0; k < 4; ++k) mine[k] = fz_mulmod(fz_cent_i64(asum[k], m), (double)yv[k], m);      // |.| <= q/2 + eps each
    __syncthreads();
    if (threadIdx.x < D) {
        double sum = 0;
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; ++w)
#pragma unroll
            for (int q = 0; q < PPW; ++q) sum += accbuf[w * 256 + q * D + threadIdx.x];
        vk[seg * D + threadIdx.x] = (int)fz_cent(sum, m);
    }
}

// ------------------------------------------------------------------------------------------
// Fused verification (fusion/fusion.py:690-727): sigma is read ONCE.  While a row of sigma is in registers it
// feeds both (a) observed += A[k] (.) sigma[k] and (b) the radix-4 inverse transform, whose centred outputs are
// only reduced (max |x| per aggregate, weight per row) and never stored.  The l rows of one aggregate are spread
// over gridDim.x workgroups (a single aggregate -- the common call -- would otherwise occupy one CU): each
// adds its exact partial of `observed` into the aggregate's accumulator and counts itself (and its norm / weight
// failures) in the aggregate's state word; the workgroup that arrives last compares with the target, applies the
// reference's verdict order (target mismatch, norm, weight) and re-arms accumulator and state for the next launch.
// ------------------------------------------------------------------------------------------
constexpr int kVerifyWaves = 4;

// 4 consecutive stored values of a row, as loaded (the request is issued one row ahead of its use) and as doubles:
// int32 rows as they are (any int32), int64 rows -- exact partial sums straight from the cross-GPU all-reduce --
// centred on unpacking
template <typename T> struct Raw4;
template <> struct Raw4<int32_t> {
    int4 v;
    __device__ __forceinline__ void load(const int32_t *p) { v = nt_load4(p); }      // an aggregate's rows are read once
    __device__ __forceinline__ void unpack(double (&a)[4], const FzMod &) const {
        a[0] = (double)v.x; a[1] = (double)v.y; a[2] = (double)v.z; a[3] = (double)v.w;
    }
    __device__ __forceinline__ void ints(int (&s)[4], const double (&)[4]) const { s[0] = v.x; s[1] = v.y; s[2] = v.z; s[3] = v.w; }
};
template <> struct Raw4<int64_t> {
    longlong2 lo, hi;
    __device__ __forceinline__ void load(const int64_t *p) {
        lo = reinterpret_cast<const longlong2 *>(p)[0];
        hi = reinterpret_cast<const longlong2 *>(p)[1];
    }
    __device__ __forceinline__ void unpack(double (&a)[4], const FzMod &m) const {
        a[0] = fz_cent_i64(lo.x, m); a[1] = fz_cent_i64(lo.y, m);          // exact for any int64
        a[2] = fz_cent_i64(hi.x, m); a[3] = fz_cent_i64(hi.y, m);
    }
    __device__ __forceinline__ void ints(int (&s)[4], const double (&a)[4]) const {       // the centred values just unpacked
        s[0] = (int)a[0]; s[1] = (int)a[1]; s[2] = (int)a[2]; s[3] = (int)a[3];
    }
};
__device__ __forceinline__ int centred_any(int32_t v, const FzMod &) { return v; }
__device__ __forceinline__ int centred_any(int64_t v, const FzMod &m) { return (int)fz_cent_i64(v, m); }

// IMAD: A (.) sigma in 64-bit integer multiply-adds (see keygen_fused).  `lazy` (host-decided, uniform): beta < q/2 - q * 2^-12, so the norm
// test needs no centring at all -- the inverse transform's outputs r satisfy |r| <= q/2 + q * 2^-13; if |r| <= beta then r is
// already the centred residue and passes; if |r| > beta then |cent(r)| >= q - |r| >= q/2 - q * 2^-13 > beta (or cent(r) = r):
// max |r| > beta <=> max |cent(r)| > beta.  Likewise r == 0 (mod q) <=> r == 0, since |r| < q.  Saves 8 of ~180 ops per row.
template <int LOGD, bool FAST, typename T, bool ORDERED, bool IMAD>
__global__ __launch_bounds__(64 * kVerifyWaves) void verify_fused(const int32_t *A, const T *sig,
                                                                  size_t sig_stride,
                                                                  const T *target, size_t target_stride, int l, long long beta,
                                                                  long long omega, int lazy, const double2 *__restrict__ itw2,
                                                                  FzTwA twA, FzMod m, double *part, int *state, int *verdict) {
    constexpr int NR = 1;                 // one row group per wave iteration (two: 248.5 against 243.9 us per 8192 aggregates, round 3)
    using TW = double2;
    constexpr int D = 1 << LOGD, LP = D / 4, PPW = 64 / LP;
    static_assert(D <= 64 * kVerifyWaves, "one thread per coefficient in the combine steps");
    __shared__ __attribute__((aligned(16))) double lds[kVerifyWaves * 256 * (NR + 1)];
    __shared__ int s_flags, s_last;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;      // the wave index is uniform: say so (scalar address arithmetic)
    const int p = lane / LP, mm = lane % LP;
    double *region = lds + wave * NR * 256 + p * D;
    double *accbuf = lds + kVerifyWaves * NR * 256;
    if (threadIdx.x == 0) s_flags = 0;
    const int R = gridDim.x, r = blockIdx.x, g = blockIdx.y;
    sig += (size_t)g * sig_stride;
    target += (size_t)g * target_stride;
    part += (size_t)g * D;                          // [groups][D] exact fp64 sums of `observed`, zero between launches
    state += g;                                     // arrivals (bits 0-15), norm failures (16-23), weight failures (24-31)

    TW twl[LOGD / 2 - 1][3];
    inv4_load_twiddles<LOGD, TW>(twl, itw2, mm);

    double acc[4] = {0, 0, 0, 0};
    double mx = 0.0;                                // max |centred output|, kept as a double: |.| <= q/2, exact
    int wfail = 0;
    const bool weigh = omega < (long long)D;        // a row has D coefficients: a bound of D or more cannot fail
    const unsigned long long gmask = LP == 64 ? ~0ull : (((1ull << (LP & 63)) - 1ull) << ((LP * p) & 63));
    const int tasks = (l + PPW - 1) / PPW, step = R * kVerifyWaves;
    // a wave's rows are a sequential chain: the next row (sigma from HBM, A from the L2) is requested before this row's
    // passes start -- unconditionally, clamped to the last task, so that no branch stands between request and use.  Without
    // it a workgroup per aggregate (many aggregates per launch) paid one memory latency per row: 24 % of the HBM peak.
    // NR row groups per iteration (tasks t, t + step, ..): they go through the inverse passes in lock step (inv4_passes_n)
    Raw4<T> rn[NR];
    int4 an[NR];
    auto fetch = [&](int t) {
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int row = (t + j * step) * PPW + p;
            const size_t off = (size_t)(row < l ? row : l - 1) * D + 4 * mm;
            an[j] = *reinterpret_cast<const int4 *>(A + off);
            rn[j].load(sig + off);
        }
    };
    long long ihi[4] = {0, 0, 0, 0}, ilo[4] = {0, 0, 0, 0};      // IMAD: exact integer sums of sigma * hi and sigma * lo
    int task = r * kVerifyWaves + wave;
    if (task < tasks) fetch(task);
    for (; task < tasks; task += NR * step) {
        double a[NR][4];
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int row = (task + j * step) * PPW + p;
            const bool valid = row < l;
            const int4 ak = an[j];
            int si[4];
            rn[j].unpack(a[j], m);
            if constexpr (IMAD) rn[j].ints(si, a[j]);
            if (valid) {
                if constexpr (IMAD) {
                    const int av[4] = {ak.x, ak.y, ak.z, ak.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {       // any int32 sigma, any int32 A: |sigma * hi|, |sigma * lo| < 2^47
                        ihi[k] += (long long)si[k] * (long long)(av[k] >> 16);
                        ilo[k] += (long long)si[k] * (long long)(av[k] & 0xffff);
                    }
                } else {
                    acc[0] += fz_mulmod(a[j][0], (double)ak.x, m);
                    acc[1] += fz_mulmod(a[j][1], (double)ak.y, m);
                    acc[2] += fz_mulmod(a[j][2], (double)ak.z, m);
                    acc[3] += fz_mulmod(a[j][3], (double)ak.w, m);
                }
            }
        }
        fetch(task + NR * step < tasks ? task + NR * step : tasks - 1);
        inv4_passes_n<LOGD, FAST, NR, TW>(a, region, twl, twA, m, mm);
        // norm and weight of the rows stay in the fp64 lanes (no conversions): a slot past the last row repeats row l - 1, which
        // changes neither the maximum nor any row's weight.  Weight = population count of "non-zero" ballots (scalar unit).
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            if (!lazy) {
#pragma unroll
                for (int k = 0; k < 4; ++k) a[j][k] = fz_cent(a[j][k], m);          // canonical: zero mod q <=> 0
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) mx = __builtin_fmax(mx, __builtin_fabs(a[j][k]));
            if (weigh) {
                int cnt = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) cnt += __popcll(__ballot(a[j][k] != 0.0) & gmask);
                if ((long long)cnt > omega) wfail = 1;
            }
        }
        wave_sync();      // the next rows' first-pass writes must not overtake these rows' last reads
    }
    if constexpr (IMAD) {
        const bool small = tasks <= 32 * step;                    // rows per wave <= 32
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = fz_imad_total(ihi[k], ilo[k], small, m);
    }
    // partial products of this wave, indexed by (row slot p, position)
    double *mine = accbuf + wave * 256 + p * D + 4 * mm;
    mine[0] = acc[0]; mine[1] = acc[1]; mine[2] = acc[2]; mine[3] = acc[3];
    if (mx > (double)beta) atomicOr(&s_flags, 2);       // mx < 2^31 and integer-valued; beta as a double rounds only above 2^53
    if (wfail) atomicOr(&s_flags, 4);
    __syncthreads();
    if (R == 1) {                                   // the whole aggregate is this workgroup's: nothing to share
        if (threadIdx.x < D) {
            double sum = 0;
            for (int w = 0; w < kVerifyWaves; ++w)
#pragma unroll
                for (int q = 0; q < PPW; ++q) sum += accbuf[w * 256 + q * D + threadIdx.x];
            if ((int)fz_cent_wide(sum, m) != centred_any(target[threadIdx.x], m)) atomicOr(&s_flags, 1);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int f = s_flags;
            verdict[g] = (f & 1) ? FZ_VERDICT_TARGET_MISMATCH : ((f & 2) ? FZ_VERDICT_NORM : ((f & 4) ? FZ_VERDICT_WEIGHT : FZ_VERDICT_OK));
        }
        return;
    }
    // Cross-workgroup combine WITHOUT device-scope fences (a __threadfence() is an L2 write-back on this chip: several
    // microseconds each, serialised over the workgroups).  Everything shared travels in device-scope atomics, which
    // are performed at the memory side: exact fp64 adds of integer partials (|.| < l * q < 2^53, order-independent),
    // then ONE integer add that counts the arrival and the norm / weight failures.  A returning atomic has been
    // performed when its result is back, so "data before arrival" needs no fence.
    if (threadIdx.x < D) {
        double sum = 0;
        for (int w = 0; w < kVerifyWaves; ++w)
#pragma unroll
            for (int q = 0; q < PPW; ++q) sum += accbuf[w * 256 + q * D + threadIdx.x];
        const double before = __hip_atomic_fetch_add(part + threadIdx.x, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        region[0] = before;                         // consume the result: the add is complete before the barrier below
    }
    // ... and say so to the hardware in so many words (inline asm: no compiler pass may drop or move it): every
    // add of this wave has been performed -- its old value is back -- before the wave reaches the barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const int f = s_flags;
        const unsigned inc = 1u + ((f & 2) ? (1u << 16) : 0u) + ((f & 4) ? (1u << 24) : 0u);
        // ORDERED (FZ_VERIFY_ORDERED=1): the arrival carries release/acquire semantics at agent scope as the HIP memory
        // model words it (one L2 write-back + L1 invalidate per workgroup).  The default relies on what the hardware
        // does with these operations: every shared word is ONLY ever touched by agent-scope atomics, which execute at
        // the memory side (MI355X_MICROARCH.md, "Global float atomics"), so no cache holds a copy that could be stale,
        // and the arrival cannot overtake the adds because they have returned (the wait above, the barrier).
        const unsigned old = __hip_atomic_fetch_add(reinterpret_cast<unsigned *>(state), inc,
                                                    ORDERED ? __ATOMIC_ACQ_REL : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned now = old + inc;
        s_last = ((old & 0xffffu) == (unsigned)(R - 1));
        s_flags = (((now >> 16) & 0xffu) ? 2 : 0) | ((now >> 24) ? 4 : 0);
    }
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x < D) {                          // read and re-arm in one operation
        const double sum = __hip_atomic_exchange(part + threadIdx.x, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)fz_cent_wide(sum, m) != centred_any(target[threadIdx.x], m)) atomicOr(&s_flags, 1);   // both centred
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_exchange(reinterpret_cast<unsigned *>(state), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int f = s_flags;
        verdict[g] = (f & 1) ? FZ_VERDICT_TARGET_MISMATCH : ((f & 2) ? FZ_VERDICT_NORM : ((f & 4) ? FZ_VERDICT_WEIGHT : FZ_VERDICT_OK));
    }
}

// ------------------------------------------------------------------------------------------
// D <= 16: one thread per polynomial, everything in registers, twiddles uniform
// ------------------------------------------------------------------------------------------
template <int LOGD, bool INVERSE>
__global__ __launch_bounds__(256) void ntt_small(const int32_t *in, int32_t *out, size_t batch,
                                                 FzTwA twA, FzMod m) {
    constexpr int D = 1 << LOGD;
    const size_t poly = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (poly >= batch) return;
    double a[D];
#pragma unroll
    for (int k = 0; k < D; ++k) a[k] = (double)in[poly * D + k];
    if (!INVERSE) {
#pragma unroll
        for (int s = 0; s < LOGD; ++s) {
            const int t = D >> (s + 1);
#pragma unroll
            for (int k = 0; k < D; ++k) {
                if (k & t) continue;
                const double w = twA.w[(1 << s) + (k / (2 * t))];
                const double v = fz_mulmod(a[k + t], w, m);
                const double u = a[k];
                a[k] = u + v;
                a[k + t] = u - v;
            }
        }
    } else {
#pragma unroll
        for (int s = 0; s < LOGD; ++s) {
            const int t = 1 << s;
            const int h = D >> (s + 1);
#pragma unroll
            for (int k = 0; k < D; ++k) {
                if (k & t) continue;
                const double w = twA.w[h + (k / (2 * t))];
                const double u = a[k], v = a[k + t];
                a[k] = u + v;
                a[k + t] = fz_mulmod(u - v, w, m);
            }
        }
#pragma unroll
        for (int k = 0; k < D; ++k) a[k] = fz_mulmod(a[k], twA.n_inv, m);
    }
#pragma unroll
    for (int k = 0; k < D; ++k) out[poly * D + k] = (int)fz_cent(a[k], m);
}

// ------------------------------------------------------------------------------------------
// D = 512 .. 4096 (round 5): one workgroup per polynomial, the polynomial in LDS as doubles, one workgroup barrier per stage,
// twiddles from the device table.  The reference transforms any power-of-two length over any odd prime with a 2n-th root
// (algebra/ntt.py:239-270); the scheme's own degrees (64, 256) never come here -- this is generality, not a hot path.
// Forward: lazy sums (a value grows by at most q/2 per stage: 12 stages stay below 2^35).  Inverse: u + v doubles per
// stage, so it is folded below q at every stage (fz_fold: 2 operations) and u - v stays inside fz_mulmod's bound.
// ------------------------------------------------------------------------------------------
template <bool INVERSE>
__global__ __launch_bounds__(256) void ntt_big(const int32_t *in, int32_t *out, size_t batch, int logd, const double *__restrict__ tw,
                                               FzMod m, double n_inv) {
    extern __shared__ double big_a[];
    const int d = 1 << logd, half = d >> 1, tid = threadIdx.x;
    for (size_t poly = blockIdx.x; poly < batch; poly += gridDim.x) {
        for (int i = tid; i < d; i += 256) big_a[i] = (double)in[poly * d + i];
        __syncthreads();
        if (!INVERSE) {
            for (int mm = 1, t = half; mm < d; mm <<= 1, t >>= 1) {            // ntt.py:274-290
                for (int b = tid; b < half; b += 256) {
                    const int i = b / t, j = 2 * i * t + (b - i * t);
                    const double v = fz_mulmod(big_a[j + t], tw[mm + i], m), u = big_a[j];
                    big_a[j] = u + v;
                    big_a[j + t] = u - v;
                }
                __syncthreads();
            }
            for (int i = tid; i < d; i += 256) out[poly * d + i] = (int)fz_cent(big_a[i], m);
        } else {
            for (int h = half, t = 1; h >= 1; h >>= 1, t <<= 1) {               // ntt.py:354-372
                for (int b = tid; b < half; b += 256) {
                    const int i = b / t, j = 2 * i * t + (b - i * t);
                    const double u = big_a[j], v = big_a[j + t];
                    big_a[j] = fz_fold(u + v, m);
                    big_a[j + t] = fz_mulmod(u - v, tw[h + i], m);
                }
                __syncthreads();
            }
            for (int i = tid; i < d; i += 256) out[poly * d + i] = (int)fz_cent(fz_mulmod(big_a[i], n_inv, m), m);      // ntt.py:373-376
        }
        __syncthreads();                                                         // (the next polynomial reuses the array)
    }
}

int launch_big(fz_ctx *ctx, const int32_t *in, int32_t *out, size_t batch, bool inverse) {
    const size_t lds = sizeof(double) << ctx->logd;
    const size_t cap = (size_t)ctx->num_cu * 8;
    const unsigned grid = (unsigned)(batch < cap ? batch : cap);
    if (!inverse)
        hipLaunchKernelGGL((ntt_big<false>), dim3(grid), dim3(256), lds, ctx->stream, in, out, batch, ctx->logd, (const double *)ctx->d_tw, ctx->mod, 0.0);
    else
        hipLaunchKernelGGL((ntt_big<true>), dim3(grid), dim3(256), lds, ctx->stream, in, out, batch, ctx->logd, (const double *)ctx->d_itw, ctx->mod,
                           ctx->itwA.n_inv);
    return fz_check_hip(hipGetLastError(), "ntt_big launch");
}

template <int LOGD, bool FAST>
int launch16f(fz_ctx *ctx, const int32_t *in, int32_t *out, size_t batch, bool inverse) {
    const size_t tasks = (batch * Geom<LOGD>::D + kChunk - 1) / kChunk;
    const size_t blocks = (tasks + kWavesPerBlock - 1) / kWavesPerBlock;
    const size_t cap = (size_t)(inverse ? ctx->grid_inv : ctx->grid_fwd);
    const unsigned grid = (unsigned)(blocks < cap ? blocks : cap);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ctx->prof_on && ctx->prof_n < ctx->prof_cap && (ctx->prof_seen[inverse ? 1 : 0]++ % ctx->prof_every) == 0) {
        e0 = ctx->prof_ev[2 * ctx->prof_n];
        e1 = ctx->prof_ev[2 * ctx->prof_n + 1];
        ctx->prof_kind[ctx->prof_n++] = inverse ? 1 : 0;
    }
    const dim3 block(64 * kWavesPerBlock);
    if (!inverse)
        hipExtLaunchKernelGGL((ntt_fwd16<LOGD, FAST>), dim3(grid), block, 0, ctx->stream, e0, e1, 0, in, out, batch,
                              (const double2 *)ctx->d_twB, ctx->twA, ctx->mod);
    else
        hipExtLaunchKernelGGL((ntt_inv16<LOGD, FAST>), dim3(grid), block, 0, ctx->stream, e0, e1, 0, in, out, batch,
                              (const double2 *)ctx->d_itwB, ctx->itwA, ctx->mod);
    return fz_check_hip(hipGetLastError(), "ntt16 launch");
}

// rows per wave by batch size: enough waves to fill the chip first (about four per SIMD), then more rows per wave
template <int LOGD, bool FAST, int NR, int WAVES>
void launch4n(fz_ctx *ctx, const int32_t *in, int32_t *out, size_t batch, bool inverse, hipEvent_t e0, hipEvent_t e1) {
    constexpr int PPW = 64 / ((1 << LOGD) / 4);
    const size_t tasks = (batch + (size_t)NR * PPW - 1) / ((size_t)NR * PPW);
    const dim3 grid((unsigned)((tasks + WAVES - 1) / WAVES)), block(64 * WAVES);
    if (!inverse)
        hipExtLaunchKernelGGL((ntt_fwd4<LOGD, FAST, NR, WAVES>), grid, block, 0, ctx->stream, e0, e1, 0, in, out, batch,
                              (const double2 *)ctx->d_tw2, fz_tw4(ctx->twA), ctx->mod);
    else
        hipExtLaunchKernelGGL((ntt_inv4<LOGD, FAST, NR, WAVES>), grid, block, 0, ctx->stream, e0, e1, 0, in, out, batch,
                              (const double2 *)ctx->d_itw2, fz_tw4(ctx->itwA), ctx->mod);
}

template <int LOGD, bool FAST>
int launch4f(fz_ctx *ctx, const int32_t *in, int32_t *out, size_t batch, bool inverse) {
    constexpr int PPW = 64 / ((1 << LOGD) / 4);
    const size_t waves1 = (batch + PPW - 1) / PPW;                   // waves at one row group per wave
    if (waves1 > 0x7fffffffull) return fz_set_error(FZ_E_UNSUPPORTED, "batch too large for the radix-4 schedule");
    int nr = ctx->knob_ntt_rows;                                      // FZ_NTT_ROWS (tests: every row count at small sizes)
    if (nr != 1 && nr != 2 && nr != 4) nr = waves1 <= (size_t)24 * ctx->num_cu ? 1 : (waves1 <= (size_t)48 * ctx->num_cu ? 2 : 4);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ctx->prof_on && ctx->prof_n < ctx->prof_cap && (ctx->prof_seen[inverse ? 1 : 0]++ % ctx->prof_every) == 0) {
        e0 = ctx->prof_ev[2 * ctx->prof_n];
        e1 = ctx->prof_ev[2 * ctx->prof_n + 1];
        ctx->prof_kind[ctx->prof_n++] = inverse ? 1 : 0;
    }
    if (nr == 1) {
        // waves per workgroup: 8 once that still leaves a workgroup for every CU (fewer, fatter workgroups are handed out
        // sooner), else 4, else 1 -- 4096 rows of degree 64 are 1024 waves: as 128 workgroups they would leave half the chip idle
        const int w = waves1 >= (size_t)8 * ctx->num_cu ? 8 : (waves1 >= (size_t)4 * ctx->num_cu ? 4 : 1);
        if (w == 1) launch4n<LOGD, FAST, 1, 1>(ctx, in, out, batch, inverse, e0, e1);
        else if (w == 4) launch4n<LOGD, FAST, 1, 4>(ctx, in, out, batch, inverse, e0, e1);
        else launch4n<LOGD, FAST, 1, 8>(ctx, in, out, batch, inverse, e0, e1);
    }
    else if (nr == 2) launch4n<LOGD, FAST, 2, 2>(ctx, in, out, batch, inverse, e0, e1);
    else launch4n<LOGD, FAST, 4, 2>(ctx, in, out, batch, inverse, e0, e1);
    return fz_check_hip(hipGetLastError(), "ntt4 launch");
}

template <int LOGD>
int launch16(fz_ctx *ctx, const int32_t *in, int32_t *out, size_t batch, bool inverse) {
    if constexpr (LOGD == 6 || LOGD == 8) {
        // schedule choice: the radix-4 kernel below `small_batch_rows` rows (latency-bound regime)
        const bool small = ctx->force_kernel == 4 || (ctx->force_kernel == 0 && batch < (size_t)ctx->small_batch_rows);
        if (small)
            return ctx->mod.fast ? launch4f<LOGD, true>(ctx, in, out, batch, inverse)
                                 : launch4f<LOGD, false>(ctx, in, out, batch, inverse);
    }
    return ctx->mod.fast ? launch16f<LOGD, true>(ctx, in, out, batch, inverse)
                         : launch16f<LOGD, false>(ctx, in, out, batch, inverse);
}

template <int LOGD>
int launch_small(fz_ctx *ctx, const int32_t *in, int32_t *out, size_t batch, bool inverse) {
    const unsigned grid = (unsigned)((batch + 255) / 256);
    if (!inverse)
        hipLaunchKernelGGL((ntt_small<LOGD, false>), dim3(grid), dim3(256), 0, ctx->stream, in, out, batch,
                           ctx->twA, ctx->mod);
    else
        hipLaunchKernelGGL((ntt_small<LOGD, true>), dim3(grid), dim3(256), 0, ctx->stream, in, out, batch,
                           ctx->itwA, ctx->mod);
    return fz_check_hip(hipGetLastError(), "ntt_small launch");
}

template <int LOGD, bool FAST>
int query16f(fz_ctx *ctx) {
    int nf = 0, ni = 0;
    const int threads = 64 * kWavesPerBlock;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nf, ntt_fwd16<LOGD, FAST>, threads, 0);
    if (e != hipSuccess) return fz_check_hip(e, "occupancy query (fwd)");
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&ni, ntt_inv16<LOGD, FAST>, threads, 0);
    if (e != hipSuccess) return fz_check_hip(e, "occupancy query (inv)");
    if (nf < 1) nf = 1;
    if (ni < 1) ni = 1;
    ctx->grid_fwd = nf * ctx->num_cu;
    ctx->grid_inv = ni * ctx->num_cu;
    return FZ_OK;
}

template <int LOGD>
int query16(fz_ctx *ctx) {
    return ctx->mod.fast ? query16f<LOGD, true>(ctx) : query16f<LOGD, false>(ctx);
}

}  // namespace

int fz_launch_keygen_fused(fz_ctx *ctx, const int32_t *A, const int32_t *coef, int32_t *sk_hat, int32_t *vk, size_t segments,
                           int l, bool broadcast) {
    const dim3 grid((unsigned)segments), block(64 * kWavesPerBlock);
    if (broadcast && (ctx->logd == 6 || ctx->logd == 8)) {
        // one polynomial per (key, half): ONE transform per workgroup, then l stores (see keygen_bcast_fused)
#define FZ_KB(LOGD, FAST) hipLaunchKernelGGL((keygen_bcast_fused<LOGD, FAST>), grid, block, 0, ctx->stream, A, coef, sk_hat, vk, l, \
                                             (const double2 *)ctx->d_tw2, ctx->twA, ctx->mod)
        if (ctx->logd == 8) { if (ctx->mod.fast) FZ_KB(8, true); else FZ_KB(8, false); }
        else { if (ctx->mod.fast) FZ_KB(6, true); else FZ_KB(6, false); }
#undef FZ_KB
        return fz_check_hip(hipGetLastError(), "keygen_bcast_fused launch");
    }
    if (broadcast) return fz_set_error(FZ_E_UNSUPPORTED, "one-polynomial keygen: degree 64 or 256 only");    // (other degrees: the caller's three launches)
    const size_t seg_stride = (size_t)l * ctx->degree, row_stride = (size_t)ctx->degree;
    // integer accumulation is exact for at most 2^15 products per lane (fz_arith.h): longer sums take the fp64 form
    const bool imad_k = !ctx->knob_no_imad && l <= (1 << 15);
#define FZ_KF2(LOGD, FAST, IM) hipLaunchKernelGGL((keygen_fused<LOGD, FAST, IM>), grid, block, 0, ctx->stream, A, coef, seg_stride, row_stride, sk_hat, vk, l, \
                                             (const double2 *)ctx->d_tw2, ctx->twA, ctx->mod)
#define FZ_KF(LOGD, FAST) do { if (imad_k) FZ_KF2(LOGD, FAST, true); else FZ_KF2(LOGD, FAST, false); } while (0)
    if (ctx->logd == 8) { if (ctx->mod.fast) FZ_KF(8, true); else FZ_KF(8, false); }
    else if (ctx->logd == 6) { if (ctx->mod.fast) FZ_KF(6, true); else FZ_KF(6, false); }
    else return fz_set_error(FZ_E_UNSUPPORTED, "fused keygen: degree 64 or 256 only");
#undef FZ_KF
#undef FZ_KF2
    return fz_check_hip(hipGetLastError(), "keygen_fused launch");
}

template <typename T>
static int launch_verify_fused(fz_ctx *ctx, const int32_t *A, const T *sig, size_t sig_stride, const T *target,
                               size_t target_stride, size_t groups, int l, int64_t beta, int64_t omega, int *d_verdict) {
    if (ctx->logd != 6 && ctx->logd != 8) return fz_set_error(FZ_E_UNSUPPORTED, "fused verify: degree 64 or 256 only");
    // about one row per wave while that leaves the chip under-filled (measured: one aggregate 22 us with one workgroup,
    // 4.5 us with 21; 64 aggregates 9.2 us with 4-8 workgroups each, 14.6 us with 22)
    const int ppw = 64 / (ctx->degree / 4), tasks = (l + ppw - 1) / ppw;
    int R = (tasks + kVerifyWaves - 1) / kVerifyWaves;
    const int fill = (int)((size_t)ctx->num_cu * 2 / groups);
    if (R > fill) R = fill;
    if (R < 1) R = 1;
    if (R > 64) R = 64;
    // the inverse passes leave |r| <= q/2 + q * 2^-13 (4-op multiply) -- see the kernel's header for why no centring is needed then
    const int lazy = (beta >= 0 && (double)beta < 0.5 * ctx->mod.q - ctx->mod.q / 4096.0 && !ctx->knob_verify_cent) ? 1 : 0;
    // (Round 3 also ran launches with a workgroup per aggregate through a 16-per-lane kernel, verify_many16: 245 us against 237
    // per 8192 aggregates at (83, 256), 130 against 119 at (195, 64) -- profiles/r03_verify_ab.txt -- although its transform
    // structure is 28-54 % faster from registers and LDS alone (profiles/r03_ntt_structures.txt): 149 VGPRs, 3 waves per SIMD
    // against 5.  Removed in round 4; tools/microbench/ntt_structures.hip keeps the structure comparison.)
    double *part = nullptr;
    int *state = nullptr;
    int rc = fz_verify_scratch(ctx, groups, (size_t)ctx->degree, &part, &state);
    if (rc != FZ_OK) return rc;
    const dim3 grid((unsigned)R, (unsigned)groups), block(64 * kVerifyWaves);
#define FZ_VF3(LOGD, FAST, ORD, IM) hipLaunchKernelGGL((verify_fused<LOGD, FAST, T, ORD, IM>), grid, block, 0, ctx->stream, A, sig, sig_stride, target, \
                                                   target_stride, l, (long long)beta, (long long)omega, lazy, (const double2 *)ctx->d_itw2, \
                                                   ctx->itwA, ctx->mod, part, state, d_verdict)
    // integer accumulation of A * sigma pays its once-per-wave conversion back only over several rows per wave (measured: 1.18 M
    // vector instructions against 1.10 M per launch when the l rows are spread one per wave over 21 workgroups)
    // ... and it is exact for at most 2^15 products per lane (fz_arith.h): a longer sum takes the fp64 form
    const bool imad = !ctx->knob_no_imad && l <= (1 << 15) && (tasks + R * kVerifyWaves - 1) / (R * kVerifyWaves) >= 4;
#define FZ_VF2(LOGD, FAST, ORD) do { if (imad) FZ_VF3(LOGD, FAST, ORD, true); else FZ_VF3(LOGD, FAST, ORD, false); } while (0)
#define FZ_VF(LOGD, FAST) do { if (ctx->knob_verify_ordered) FZ_VF2(LOGD, FAST, true); else FZ_VF2(LOGD, FAST, false); } while (0)
    if (ctx->logd == 8) { if (ctx->mod.fast) FZ_VF(8, true); else FZ_VF(8, false); }
    else { if (ctx->mod.fast) FZ_VF(6, true); else FZ_VF(6, false); }
#undef FZ_VF
#undef FZ_VF2
#undef FZ_VF3
    rc = fz_check_hip(hipGetLastError(), "verify_fused launch");
    if (rc != FZ_OK) ctx->verify_dirty = 1;          // the accumulators may be left non-zero: re-zeroed before the next launch
    return rc;
}

int fz_launch_verify_fused(fz_ctx *ctx, const int32_t *A, const int32_t *sig, const int32_t *target, size_t groups, int l,
                           int64_t beta, int64_t omega, int *d_verdict) {
    return launch_verify_fused<int32_t>(ctx, A, sig, (size_t)l * ctx->degree, target, (size_t)ctx->degree, groups, l, beta, omega,
                                        d_verdict);
}

// the aggregates and targets as int64 partial sums (e.g. straight after the all-reduce), group g at base + g * stride
int fz_launch_verify_fused_i64(fz_ctx *ctx, const int32_t *A, const int64_t *sig, size_t sig_stride, const int64_t *target,
                               size_t target_stride, size_t groups, int l, int64_t beta, int64_t omega, int *d_verdict) {
    return launch_verify_fused<int64_t>(ctx, A, sig, sig_stride, target, target_stride, groups, l, beta, omega, d_verdict);
}

// the 16-per-lane form of the fused product (degrees 32..256, 16-byte aligned operands)
template <int LOGD, bool FAST>
static int launch_polymul16(fz_ctx *ctx, const int32_t *f, const int32_t *g, int32_t *out, size_t batch) {
    if (ctx->grid_pm16 == 0) {
        int n = 0;
        const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, polymul16<LOGD, FAST>, 64 * kWavesPerBlock, 0);
        if (e != hipSuccess) return fz_check_hip(e, "occupancy query (polymul16)");
        ctx->grid_pm16 = (n < 1 ? 1 : n) * ctx->num_cu;
    }
    const size_t tasks = (batch * (size_t)ctx->degree + kChunk - 1) / kChunk, blocks = (tasks + kWavesPerBlock - 1) / kWavesPerBlock;
    const dim3 grid((unsigned)(blocks < (size_t)ctx->grid_pm16 ? blocks : (size_t)ctx->grid_pm16)), block(64 * kWavesPerBlock);
    hipLaunchKernelGGL((polymul16<LOGD, FAST>), grid, block, 0, ctx->stream, f, g, out, batch, (const double2 *)ctx->d_twB,
                       (const double2 *)ctx->d_itwB, (const FzTwA *)ctx->d_twAB, ctx->mod);
    return fz_check_hip(hipGetLastError(), "polymul16 launch");
}

// Which form (tools/probes/polymul_crossover.py, profiles/r06_polymul_crossover.txt): at degree 256 the 16-per-lane kernel from
// 2^14 products on (18.2 against 19.3 us there, 118 against 130 us at 2^17: 42.6 % of 8 TB/s against 38.8 %; below, its start-up
// -- a table twice the size, two chunks per wave before the first butterfly -- costs more than the exchanges it saves); at
// degree 64 the radix-4 kernel at every size (three passes instead of four: 45.5 % at 2^17 products against 44.7 %); degrees
// 32 and 128 have no radix-4 form.  FZ_POLYMUL_FORM = 1 | 2 forces one.
constexpr size_t kPolymul16MinRows256 = (size_t)1 << 14;

bool fz_polymul16_ok(const fz_ctx *ctx, const int32_t *f, const int32_t *g, const int32_t *out, size_t batch) {
    if (ctx->logd < 5 || ctx->logd > 8 || ctx->knob_polymul_form == 1) return false;
    if ((((uintptr_t)f | (uintptr_t)g | (uintptr_t)out) & 15) != 0) return false;
    if (ctx->logd == 5 || ctx->logd == 7 || ctx->knob_polymul_form == 2) return true;
    return ctx->logd == 8 && batch >= kPolymul16MinRows256;
}

// fused product: degrees 64 / 256 in either form, 32 / 128 in the 16-per-lane form; the caller composes the generic path otherwise
int fz_launch_polymul_fused(fz_ctx *ctx, const int32_t *f, const int32_t *g, int32_t *out, size_t batch) {
    if (batch == 0) return FZ_OK;
    if (fz_polymul16_ok(ctx, f, g, out, batch)) {
        switch (ctx->logd) {
            case 5: return ctx->mod.fast ? launch_polymul16<5, true>(ctx, f, g, out, batch) : launch_polymul16<5, false>(ctx, f, g, out, batch);
            case 6: return ctx->mod.fast ? launch_polymul16<6, true>(ctx, f, g, out, batch) : launch_polymul16<6, false>(ctx, f, g, out, batch);
            case 7: return ctx->mod.fast ? launch_polymul16<7, true>(ctx, f, g, out, batch) : launch_polymul16<7, false>(ctx, f, g, out, batch);
            default: return ctx->mod.fast ? launch_polymul16<8, true>(ctx, f, g, out, batch) : launch_polymul16<8, false>(ctx, f, g, out, batch);
        }
    }
    if (ctx->logd != 6 && ctx->logd != 8) return fz_set_error(FZ_E_UNSUPPORTED, "fused product: degree 64 or 256, or 16-byte aligned operands of degree 32..256");
    const int ppw = 64 / (ctx->degree / 4);
    const size_t tasks = (batch + ppw - 1) / ppw, blocks = (tasks + kWavesPerBlock - 1) / kWavesPerBlock;
    if (ctx->grid_pm == 0) {
        int n = 0;
        hipError_t e;
#define FZ_PQ(LOGD, FAST) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, polymul_fused<LOGD, FAST>, 64 * kWavesPerBlock, 0)
        if (ctx->logd == 8) { if (ctx->mod.fast) FZ_PQ(8, true); else FZ_PQ(8, false); }
        else { if (ctx->mod.fast) FZ_PQ(6, true); else FZ_PQ(6, false); }
#undef FZ_PQ
        if (e != hipSuccess) return fz_check_hip(e, "occupancy query (polymul)");
        ctx->grid_pm = (n < 1 ? 1 : n) * ctx->num_cu;
    }
    const dim3 grid((unsigned)(blocks < (size_t)ctx->grid_pm ? blocks : (size_t)ctx->grid_pm)), block(64 * kWavesPerBlock);
#define FZ_PM(LOGD, FAST) hipLaunchKernelGGL((polymul_fused<LOGD, FAST>), grid, block, 0, ctx->stream, f, g, out, batch, \
                                             (const double2 *)ctx->d_tw2, (const double2 *)ctx->d_itw2, ctx->twA, ctx->itwA, ctx->mod)
    if (ctx->logd == 8) { if (ctx->mod.fast) FZ_PM(8, true); else FZ_PM(8, false); }
    else { if (ctx->mod.fast) FZ_PM(6, true); else FZ_PM(6, false); }
#undef FZ_PM
    return fz_check_hip(hipGetLastError(), "polymul_fused launch");
}

int fz_ntt_query_grid(fz_ctx *ctx) {
    switch (ctx->logd) {
        case 5: return query16<5>(ctx);
        case 6: return query16<6>(ctx);
        case 7: return query16<7>(ctx);
        case 8: return query16<8>(ctx);
        default: ctx->grid_fwd = ctx->grid_inv = 0; return FZ_OK;
    }
}

int fz_launch_ntt(fz_ctx *ctx, const int32_t *d_in, int32_t *d_out, size_t batch, bool inverse) {
    if (ctx->logd < 0) return fz_set_error(FZ_E_UNSUPPORTED, "ring-only context (created with root 0) has no transforms");
    if (batch == 0) return FZ_OK;
    if ((((uintptr_t)d_in | (uintptr_t)d_out) & 15) != 0 && ctx->logd >= 2)
        return fz_set_error(FZ_E_BADARG, "transform buffers must be 16-byte aligned");
    switch (ctx->logd) {
        case 1: return launch_small<1>(ctx, d_in, d_out, batch, inverse);
        case 2: return launch_small<2>(ctx, d_in, d_out, batch, inverse);
        case 3: return launch_small<3>(ctx, d_in, d_out, batch, inverse);
        case 4: return launch_small<4>(ctx, d_in, d_out, batch, inverse);
        case 5: return launch16<5>(ctx, d_in, d_out, batch, inverse);
        case 6: return launch16<6>(ctx, d_in, d_out, batch, inverse);
        case 7: return launch16<7>(ctx, d_in, d_out, batch, inverse);
        case 8: return launch16<8>(ctx, d_in, d_out, batch, inverse);
        case 9: case 10: case 11: case 12: return launch_big(ctx, d_in, d_out, batch, inverse);
        default: return fz_set_error(FZ_E_UNSUPPORTED, "degree %d not supported (2..%d)", ctx->degree, kFzMaxDegree);
    }
}

// fz_diag_stamps_*: a launch's workgroups get a run of {entry, exit} slots (NULL: stamps off, or the recording is full)
static unsigned long long *stamp_slots(fz_ctx *ctx, unsigned total) {
    if (!ctx->stamp_on || !ctx->d_stamp || ctx->stamp_n >= ctx->stamp_launch_cap || ctx->stamp_used + total > ctx->stamp_wg_cap) return nullptr;
    unsigned long long *stamp = ctx->d_stamp + 2 * ctx->stamp_used;
    ctx->stamp_first[ctx->stamp_n] = ctx->stamp_used;
    ctx->stamp_count[ctx->stamp_n++] = total;
    ctx->stamp_used += total;
    return stamp;
}

// one launch over a job table (at most kFzMultiMax jobs, degree 64 / 256)
template <int LOGD, bool FAST, int NR, int WAVES, int NJ>
static void launch_jobs_n(fz_ctx *ctx, const FzMultiJobs &J, unsigned total, hipEvent_t e0, hipEvent_t e1, unsigned long long *stamp) {
    FzJobsN<NJ> S;
    for (int j = 0; j < NJ; ++j) { S.in[j] = J.in[j]; S.out[j] = J.out[j]; S.end[j] = J.end[j]; S.rows[j] = J.rows[j]; }
    hipExtLaunchKernelGGL((ntt_jobs4<LOGD, FAST, NR, WAVES, FzJobsN<NJ>>), dim3(total), dim3(64 * WAVES), 0, ctx->stream, e0, e1, 0, S,
                          (const double2 *)ctx->d_tw2, (const double2 *)ctx->d_itw2, fz_tw4(ctx->twA), fz_tw4(ctx->itwA), ctx->mod, stamp);
}

template <int LOGD, bool FAST, int NR, int WAVES>
static void launch_jobs(fz_ctx *ctx, FzMultiJobs &J, hipEvent_t e0, hipEvent_t e1) {
    constexpr unsigned PPW = 64 / ((1 << LOGD) / 4);
    unsigned total = 0;
    for (int j = 0; j < J.n; ++j) {
        const unsigned rows = J.rows[j] & 0x7fffffffu;
        const unsigned tasks = (rows + NR * PPW - 1) / (NR * PPW);
        total += (tasks + WAVES - 1) / WAVES;
        J.end[j] = total;
    }
    for (int j = J.n; j < kFzMultiMax; ++j) { J.end[j] = total; J.rows[j] = 0; J.in[j] = nullptr; J.out[j] = nullptr; }   // (never chosen: see the kernel)
    unsigned long long *stamp = stamp_slots(ctx, total);
    if (J.n <= 4) launch_jobs_n<LOGD, FAST, NR, WAVES, 4>(ctx, J, total, e0, e1, stamp);
    else if (J.n <= 8) launch_jobs_n<LOGD, FAST, NR, WAVES, 8>(ctx, J, total, e0, e1, stamp);
    else launch_jobs_n<LOGD, FAST, NR, WAVES, kFzMultiMax>(ctx, J, total, e0, e1, stamp);
}

// the 16-per-lane form of a multi-job launch: job j gets min(its workgroups, its share of the resident grid) workgroups
template <int LOGD, bool FAST, int NJ>
static void launch_jobs16_n(fz_ctx *ctx, const FzMultiJobs &J, unsigned total, hipEvent_t e0, hipEvent_t e1, unsigned long long *stamp) {
    FzJobsN<NJ> S;
    for (int j = 0; j < NJ; ++j) { S.in[j] = J.in[j]; S.out[j] = J.out[j]; S.end[j] = J.end[j]; S.rows[j] = J.rows[j]; }
    hipExtLaunchKernelGGL((ntt_jobs16<LOGD, FAST, FzJobsN<NJ>>), dim3(total), dim3(64 * kWavesPerBlock), 0, ctx->stream, e0, e1, 0, S,
                          (const double2 *)ctx->d_twB, (const double2 *)ctx->d_itwB, ctx->twA, ctx->itwA, ctx->mod, stamp);
}

template <int LOGD, bool FAST>
static int launch_jobs16(fz_ctx *ctx, FzMultiJobs &J, hipEvent_t e0, hipEvent_t e1) {
    constexpr size_t D = (size_t)1 << LOGD;
    size_t all_tasks = 0;
    for (int j = 0; j < J.n; ++j) all_tasks += ((J.rows[j] & 0x7fffffffu) * D + kChunk - 1) / kChunk;
    const size_t cap = (size_t)std::min(ctx->grid_fwd, ctx->grid_inv);      // workgroups the chip holds at once
    unsigned total = 0;
    for (int j = 0; j < J.n; ++j) {
        const size_t tasks = ((J.rows[j] & 0x7fffffffu) * D + kChunk - 1) / kChunk;
        const size_t blocks = (tasks + kWavesPerBlock - 1) / kWavesPerBlock;
        size_t share = (cap * tasks + all_tasks - 1) / all_tasks;            // proportional, rounded up, at least one
        if (share < 1) share = 1;
        total += (unsigned)(tasks ? std::min(blocks, share) : 0);
        J.end[j] = total;
    }
    for (int j = J.n; j < kFzMultiMax; ++j) { J.end[j] = total; J.rows[j] = 0; J.in[j] = nullptr; J.out[j] = nullptr; }
    if (total == 0) return FZ_OK;
    unsigned long long *stamp = stamp_slots(ctx, total);
    if (J.n <= 4) launch_jobs16_n<LOGD, FAST, 4>(ctx, J, total, e0, e1, stamp);
    else if (J.n <= 8) launch_jobs16_n<LOGD, FAST, 8>(ctx, J, total, e0, e1, stamp);
    else launch_jobs16_n<LOGD, FAST, kFzMultiMax>(ctx, J, total, e0, e1, stamp);
    return fz_check_hip(hipGetLastError(), "ntt_jobs16 launch");
}

template <int LOGD, bool FAST>
static int launch_jobs_f(fz_ctx *ctx, FzMultiJobs &J) {
    constexpr unsigned PPW = 64 / ((1 << LOGD) / 4);
    unsigned long long waves1 = 0;                                     // waves at one row group per wave, over all jobs
    for (int j = 0; j < J.n; ++j) waves1 += ((J.rows[j] & 0x7fffffffu) + PPW - 1) / PPW;
    if (waves1 == 0) return FZ_OK;
    if (waves1 > 0x7fffffffull) return fz_set_error(FZ_E_UNSUPPORTED, "too many rows for one multi-job launch");
    // the schedule by the launch's TOTAL rows, as for one job (launch16): the 16-per-lane kernels from `small_batch_rows` on
    // (FZ_NTT_KERNEL forces either)
    size_t all_rows = 0;
    for (int j = 0; j < J.n; ++j) all_rows += J.rows[j] & 0x7fffffffu;
    const bool big = ctx->force_kernel == 16 || (ctx->force_kernel == 0 && all_rows >= (size_t)ctx->small_batch_rows);
    if (big) {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (ctx->prof_on && ctx->prof_n < ctx->prof_cap && (ctx->prof_seen[0]++ % ctx->prof_every) == 0) {
            e0 = ctx->prof_ev[2 * ctx->prof_n];
            e1 = ctx->prof_ev[2 * ctx->prof_n + 1];
            ctx->prof_kind[ctx->prof_n++] = 2;
        }
        return launch_jobs16<LOGD, FAST>(ctx, J, e0, e1);
    }
    // the same launch shapes, by the same rule, as the one-job kernels (launch4f): rows per wave by the launch's total
    int nr = ctx->knob_ntt_rows;
    if (nr != 1 && nr != 2 && nr != 4) nr = waves1 <= (size_t)24 * ctx->num_cu ? 1 : (waves1 <= (size_t)48 * ctx->num_cu ? 2 : 4);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ctx->prof_on && ctx->prof_n < ctx->prof_cap && (ctx->prof_seen[0]++ % ctx->prof_every) == 0) {
        e0 = ctx->prof_ev[2 * ctx->prof_n];
        e1 = ctx->prof_ev[2 * ctx->prof_n + 1];
        ctx->prof_kind[ctx->prof_n++] = 2;                             // a multi-job launch
    }
    if (nr == 1) {
        const int w = waves1 >= (size_t)8 * ctx->num_cu ? 8 : (waves1 >= (size_t)4 * ctx->num_cu ? 4 : 1);
        if (w == 1) launch_jobs<LOGD, FAST, 1, 1>(ctx, J, e0, e1);
        else if (w == 4) launch_jobs<LOGD, FAST, 1, 4>(ctx, J, e0, e1);
        else launch_jobs<LOGD, FAST, 1, 8>(ctx, J, e0, e1);
    }
    else if (nr == 2) launch_jobs<LOGD, FAST, 2, 2>(ctx, J, e0, e1);    // (4 or 8 waves per workgroup: 6.11 / 6.04 us against 5.99 for the two-job launch)
    else launch_jobs<LOGD, FAST, 4, 2>(ctx, J, e0, e1);
    return fz_check_hip(hipGetLastError(), "ntt_jobs4 launch");
}

// J.in / J.out / J.rows (bit 31: inverse) / J.n filled by the caller; J.end is computed here (workgroups per job)
int fz_launch_ntt_multi(fz_ctx *ctx, FzMultiJobs &J) {
    if (ctx->logd != 6 && ctx->logd != 8) return fz_set_error(FZ_E_UNSUPPORTED, "multi-job transform: degree 64 or 256 only");
    if (J.n <= 0) return FZ_OK;
    if (ctx->logd == 8) return ctx->mod.fast ? launch_jobs_f<8, true>(ctx, J) : launch_jobs_f<8, false>(ctx, J);
    return ctx->mod.fast ? launch_jobs_f<6, true>(ctx, J) : launch_jobs_f<6, false>(ctx, J);
}

// ------------------------------------------------------------------------------------------
// Launch-floor diagnostics (fz_diag_*): what a dispatch of NO work and a plain copy of the same bytes cost on this
// device, measured next to the transforms so that a small batch can be judged against a same-run floor.
// ------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(64) void diag_empty_kernel() {}
__global__ __launch_bounds__(64) void diag_copy_kernel(const int4 *__restrict__ src, int4 *__restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 64;
    for (size_t i = (size_t)blockIdx.x * 64 + threadIdx.x; i < n16; i += stride) {
        const fz_v4i t = *reinterpret_cast<const fz_v4i *>(src + i);
        __builtin_nontemporal_store(t, reinterpret_cast<fz_v4i *>(dst + i));
    }
}
}  // namespace

// one wave that watches the clocks for `ticks` periods of the 100 MHz reference counter: shader cycles (s_memtime) per
// reference tick = the frequency the chip actually runs at while whatever else is resident executes
namespace {
__global__ __launch_bounds__(64) void diag_clock_kernel(unsigned long long ticks, unsigned long long *out) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < ticks) {             // every wave reaches the exit: the reference counter never stops
        __builtin_amdgcn_s_sleep(8);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    if (threadIdx.x == 0 && out) { out[0] = __builtin_amdgcn_s_memtime() - t0; out[1] = r1 - r0; }
}
}  // namespace

int fz_launch_diag_clock(hipStream_t stream, unsigned long long ticks, unsigned long long *d_out) {
    hipLaunchKernelGGL(diag_clock_kernel, dim3(1), dim3(64), 0, stream, ticks, d_out);
    return fz_check_hip(hipGetLastError(), "diag clock launch");
}

int fz_launch_diag(fz_ctx *ctx, int what, const void *src, void *dst, size_t bytes) {
    if (what == 0) {
        hipLaunchKernelGGL(diag_empty_kernel, dim3(4096), dim3(64), 0, ctx->stream);
    } else {
        const size_t n16 = bytes / 16;
        if (n16 == 0) return FZ_OK;
        // flat grid, one 16-byte item per thread: a capped grid-stride loop streams 1 GiB at 4.9-5.5 TB/s, the flat grid at
        // 6.2 TB/s (profiles/r02_launch_floor.txt) -- the ceiling this kernel exists to show
        const size_t blocks = (n16 + 63) / 64, cap = (size_t)0x7fffffff;
        hipLaunchKernelGGL(diag_copy_kernel, dim3((unsigned)(blocks < cap ? blocks : cap)), dim3(64), 0, ctx->stream,
                           (const int4 *)src, (int4 *)dst, n16);
    }
    return fz_check_hip(hipGetLastError(), "diag launch");
}
