// ntt_variants.hip -- A/B harness for the headline launch (BASELINE configs[1]: 4096 rows of degree 256, forward transform).
//
// Round 2's radix-4 forward kernel gave every row its own one-wave workgroup inside a persistent loop: 4096 workgroups, each
// loading nine per-lane (w, w * K/q) twiddle pairs (9 KiB through the vector-memory path for 1 KiB of data), computing its LDS
// offsets, transforming ONE row and leaving.  This harness timed variants of that schedule in one process, back to back, each
// checked bit for bit against the library kernel's output (profiles/r03_ntt_variants.txt, measured against round 2's kernel);
// the library's ntt_fwd4<LOGD, FAST, NR, WAVES> is what came out of it, and is now the reference line here:
//   NR     rows per wave (1 / 2 / 4): twiddle loads, index arithmetic and the wave's start-up amortise over NR rows, and the NR
//          independent rows give one wave instruction-level parallelism across the dependent fp64 chains;
//   WAVES  waves per workgroup (1 / 4 / 8 / 16);
//   TW     0 = nine per-lane global loads (as the library), 1 = the 4 KiB table staged once per workgroup in LDS;
// plus the same-process floors: an empty dispatch of each grid shape and a plain 4 MiB -> 4 MiB copy.
//
// It compiles the library's own kernel source into this translation unit (the kernels live in an anonymous namespace), so
// the baseline IS the shipped kernel.  usage: ntt_variants [reps=300]   (sweeps 2^12 .. 2^18 rows, warm and cold)
#include "../../fusion-cryptography_amd/csrc/fz_ntt.hip"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

// the few symbols fz_ntt.hip expects from fz_capi.hip
int fz_set_error(int code, const char *, ...) { return code; }
int fz_check_hip(hipError_t e, const char *what) {
    if (e != hipSuccess) { printf("HIP error in %s: %s\n", what, hipGetErrorString(e)); return FZ_E_HIP; }
    return FZ_OK;
}
int fz_verify_scratch(fz_ctx *, size_t, size_t, double **, int **) { return FZ_E_UNSUPPORTED; }

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

namespace {

// NR rows per wave, WAVES waves per workgroup, TW: twiddle source.  Degree 256 only (LOGD = 8, 64 lanes x 4 coefficients).
template <int NR, int WAVES, int TW, bool FAST>
__global__ __launch_bounds__(64 * WAVES) void fwd4_variant(const int32_t *in, int32_t *out, size_t batch,
                                                           const double2 *__restrict__ tw2, FzTwA twA, FzMod m) {
    constexpr int LOGD = 8, D = 256, LP = 64, P = 4;
    __shared__ __attribute__((aligned(16))) double lds[WAVES * NR * 256 + (TW ? 512 : 0)];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, mm = lane;
    double *region = lds + wave * NR * 256;
    const size_t tasks = (batch + NR - 1) / NR;
    const size_t task = (size_t)blockIdx.x * WAVES + wave;

    // data loads first: they have the longest way to go
    int x[NR][4];
    const bool active = task < tasks;
    if (active) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const size_t row = task * NR + r;
            const int32_t *src = in + (row < batch ? row : batch - 1) * D + mm;
#pragma unroll
            for (int k = 0; k < 4; ++k) x[r][k] = src[k * LP];
        }
    }
    double2 twl[P - 1][3];
    double2 *s_tw = reinterpret_cast<double2 *>(lds + WAVES * NR * 256);
    auto read_table = [&]() {
#pragma unroll
        for (int i = 1; i < P; ++i) {
            const int s = D >> (2 * i + 2), g = mm / s, pw = 1 << (2 * i);
            twl[i - 1][0] = s_tw[pw + g];
            twl[i - 1][1] = s_tw[2 * pw + 2 * g];
            twl[i - 1][2] = s_tw[2 * pw + 2 * g + 1];
        }
    };
    if (TW == 0) {
        fwd4_load_twiddles<LOGD>(twl, tw2, mm);
    } else {
        // the (w, w2) table, 256 entries of 16 bytes, staged once per workgroup, the loads spread over its waves
        for (int i = threadIdx.x; i < 256; i += 64 * WAVES) s_tw[i] = tw2[i];
        if (TW == 1) {
            if (WAVES == 1) wave_sync(); else __syncthreads();
            read_table();
        }
    }
    if (TW != 2 && !active) return;

    double a[NR][4];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) a[r][k] = active ? (double)x[r][k] : 0.0;

    // the four passes, all NR rows in lock step (one wave-local synchronisation per pass, not per row)
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int s = D >> (2 * i + 2);
        const int base = (mm / s) * 4 * s + mm % s;
        double wA, wA2, wB0, wB02, wB1, wB12;
        if (i == 0) {
            wA = twA.w[1]; wA2 = twA.w2[1]; wB0 = twA.w[2]; wB02 = twA.w2[2]; wB1 = twA.w[3]; wB12 = twA.w2[3];
        } else {
            if (TW == 2 && i == 1) {                 // pass 0 needed only the uniform twiddles: the table is published behind it
                if (WAVES == 1) wave_sync(); else __syncthreads();
                read_table();
            }
            wA = twl[i - 1][0].x; wA2 = twl[i - 1][0].y;
            wB0 = twl[i - 1][1].x; wB02 = twl[i - 1][1].y;
            wB1 = twl[i - 1][2].x; wB12 = twl[i - 1][2].y;
            wave_sync();
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                double *reg = region + r * 256;
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
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const size_t row = task * NR + r;
        if (active && row < batch)
            nt_store4(out + row * D + 4 * mm, make_int4((int)fz_cent(a[r][0], m), (int)fz_cent(a[r][1], m), (int)fz_cent(a[r][2], m),
                                                         (int)fz_cent(a[r][3], m)));
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Round 4: the two structures never tried.
//
// (a) fwd4_xlane -- REGISTER EXCHANGE instead of the LDS round trips (north_star: "wavefront shuffles for the in-warp butterfly
//     passes"; the stride <= 32 stages of algebra/ntt.py:274-290).  A lane holds 4 coefficients; position p = p7..p0, the two
//     register bits (r1, r0) start as (p7, p6), the lane bits as p5..p0.  Stages b = 7, 6 run on the register bits; for each later
//     stage b = 5 .. 0 the lane bit b is SWAPPED with one register bit (r1 for odd b, r0 for even b): lanes whose bit is 0 keep
//     their rb = 0 registers and receive the partner's, lanes whose bit is 1 keep rb = 1 -- every lane then holds both operands of
//     two butterflies.  Distance 32 / 16: v_permlane32_swap / v_permlane16_swap (one instruction per dword); distance 8 / 4: three
//     bank-masked DPP moves per dword; distance 2 / 1: quad_perm DPP + selects.  After stage 0 the lane holds positions 4L .. 4L+3:
//     the same coalesced 16-byte store as the library kernel.  No LDS at all.
// (b) fwd2_pair -- ONE ROW OVER TWO WAVES, 2 coefficients per lane (8 waves per SIMD at 4096 rows, half the per-wave chain):
//     four passes of two stages; inside a pass the second stage's bit sits on lane bit 5 and comes into the register with ONE
//     v_permlane32_swap per dword; between passes the row goes through LDS (three exchanges, as in the library kernel, each with
//     its own conflict-free XOR layout) and the two waves meet at a workgroup barrier.
// ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void swap32(unsigned &a, unsigned &b) {      // a.lanes[32..63] <-> b.lanes[0..31]
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0]; b = r[1];
}
__device__ __forceinline__ void swap16(unsigned &a, unsigned &b) {      // odd 16-lane rows of a <-> even rows of b
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0]; b = r[1];
}

// lanes with bit S clear keep A and receive the partner's A into B; lanes with bit S set keep B and receive the partner's B into A
template <int S>
__device__ __forceinline__ void xchg_dword(unsigned &A, unsigned &B, bool upper) {
    if constexpr (S == 32) swap32(A, B);
    else if constexpr (S == 16) swap16(A, B);
    else if constexpr (S == 8 || S == 4) {
        constexpr int SHR = 0x110 + S, SHL = 0x100 + S;                  // row_shr:S (lane i reads i - S), row_shl:S (reads i + S)
        constexpr int LOWER = S == 8 ? 0x3 : 0x5, UPPER = S == 8 ? 0xC : 0xA;       // banks = groups of 4 lanes within a row of 16
        const unsigned t = (unsigned)__builtin_amdgcn_update_dpp((int)B, (int)B, SHR, 0xF, 0xF, false);     // upper lanes: the lower partner's B
        B = (unsigned)__builtin_amdgcn_update_dpp((int)B, (int)A, SHL, 0xF, LOWER, false);                   // lower lanes: B <- the upper partner's A
        A = (unsigned)__builtin_amdgcn_update_dpp((int)A, (int)t, 0xE4, 0xF, UPPER, false);                  // upper lanes: A <- t
    } else {
        constexpr int QP = S == 2 ? 0x4E : 0xB1;                          // quad_perm [2,3,0,1] / [1,0,3,2]
        const unsigned x = upper ? A : B;
        const unsigned y = (unsigned)__builtin_amdgcn_mov_dpp((int)x, QP, 0xF, 0xF, true);
        A = upper ? y : A;
        B = upper ? B : y;
    }
}

template <int S>
__device__ __forceinline__ void xchg(double &A, double &B, bool upper) {
    unsigned a0 = (unsigned)__double2loint(A), a1 = (unsigned)__double2hiint(A), b0 = (unsigned)__double2loint(B), b1 = (unsigned)__double2hiint(B);
    xchg_dword<S>(a0, b0, upper);
    xchg_dword<S>(a1, b1, upper);
    A = __hiloint2double((int)a1, (int)a0);
    B = __hiloint2double((int)b1, (int)b0);
}

template <int WAVES, bool FAST>
__global__ __launch_bounds__(64 * WAVES) void fwd4_xlane(const int32_t *in, int32_t *out, size_t batch,
                                                         const double2 *__restrict__ tw2, FzTwA twA, FzMod m) {
    constexpr int D = 256;
    const int lane = threadIdx.x & 63;
    const size_t row = (size_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
    if (row >= batch) return;
    const int32_t *src = in + row * D + lane;
    int x[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = src[k * 64];
    // per-lane twiddles of the six cross-lane stages: index 2^(7-b) + ((lane >> b) << 1) + (the other register bit)
    double2 tw[6][2];
#pragma unroll
    for (int b = 5; b >= 0; --b)
#pragma unroll
        for (int o = 0; o < 2; ++o) tw[5 - b][o] = tw2[(1 << (7 - b)) + ((lane >> b) << 1) + o];
    double a[4];                                   // a[2 * r1 + r0]
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = (double)x[k];
    {   // stages 7 and 6 on the register bits (uniform twiddles)
        double v = tw_mul<FAST>(a[2], twA.w[1], twA.w2[1], m), u = a[0];
        a[0] = u + v; a[2] = u - v;
        v = tw_mul<FAST>(a[3], twA.w[1], twA.w2[1], m); u = a[1];
        a[1] = u + v; a[3] = u - v;
        v = tw_mul<FAST>(a[1], twA.w[2], twA.w2[2], m); u = a[0];
        a[0] = u + v; a[1] = u - v;
        v = tw_mul<FAST>(a[3], twA.w[3], twA.w2[3], m); u = a[2];
        a[2] = u + v; a[3] = u - v;
    }
#define XSTAGE(BIT)                                                                                                     \
    {                                                                                                                   \
        constexpr int S = 1 << BIT;                                                                                     \
        const bool upper = (lane >> BIT) & 1;                                                                           \
        if constexpr (BIT & 1) {            /* lane bit <-> r1: pairs (a0, a2), (a1, a3); other bit r0 = 0, 1 */       \
            xchg<S>(a[0], a[2], upper);                                                                                 \
            xchg<S>(a[1], a[3], upper);                                                                                 \
            double v = tw_mul<FAST>(a[2], tw[5 - BIT][0].x, tw[5 - BIT][0].y, m), u = a[0];                             \
            a[0] = u + v; a[2] = u - v;                                                                                 \
            v = tw_mul<FAST>(a[3], tw[5 - BIT][1].x, tw[5 - BIT][1].y, m); u = a[1];                                    \
            a[1] = u + v; a[3] = u - v;                                                                                 \
        } else {                            /* lane bit <-> r0: pairs (a0, a1), (a2, a3); other bit r1 = 0, 1 */       \
            xchg<S>(a[0], a[1], upper);                                                                                 \
            xchg<S>(a[2], a[3], upper);                                                                                 \
            double v = tw_mul<FAST>(a[1], tw[5 - BIT][0].x, tw[5 - BIT][0].y, m), u = a[0];                             \
            a[0] = u + v; a[1] = u - v;                                                                                 \
            v = tw_mul<FAST>(a[3], tw[5 - BIT][1].x, tw[5 - BIT][1].y, m); u = a[2];                                    \
            a[2] = u + v; a[3] = u - v;                                                                                 \
        }                                                                                                               \
    }
    XSTAGE(5) XSTAGE(4) XSTAGE(3) XSTAGE(2) XSTAGE(1) XSTAGE(0)
#undef XSTAGE
    nt_store4(out + row * D + 4 * lane, make_int4((int)fz_cent(a[0], m), (int)fz_cent(a[1], m), (int)fz_cent(a[2], m), (int)fz_cent(a[3], m)));
}

// (b) one row = one workgroup of two waves; ROWS rows per workgroup (2 * ROWS waves), each pair with its own LDS region
template <int ROWS, bool FAST>
__global__ __launch_bounds__(128 * ROWS) void fwd2_pair(const int32_t *in, int32_t *out, size_t batch,
                                                        const double2 *__restrict__ tw2, FzTwA twA, FzMod m) {
    constexpr int D = 256;
    __shared__ __attribute__((aligned(16))) double lds[ROWS * 256];
    const int lane = threadIdx.x & 63, w = (threadIdx.x >> 6) & 1, rl = threadIdx.x >> 7;
    const int l5 = lane >> 5, l4 = (lane >> 4) & 1, L31 = lane & 31, L15 = lane & 15;
    const size_t row = (size_t)blockIdx.x * ROWS + rl;
    const bool active = row < batch;                    // (every thread reaches every barrier)
    double *reg = lds + rl * 256;
    // pass 0 positions: p7 = register, p6 = l5, p5 = w, p4..p0 = lane & 31
    const int p_in = (l5 << 6) | (w << 5) | L31;
    int x0 = 0, x1 = 0;
    if (active) { x0 = in[row * D + p_in]; x1 = in[row * D + p_in + 128]; }
    // per-lane twiddles: stage b of a butterfly at position p uses index 2^(7-b) + (p >> (b+1))
    const int l3 = (lane >> 3) & 1, l2 = (lane >> 2) & 1;
    const double2 t6 = tw2[2 + l5];
    const double2 t5 = tw2[4 + 2 * w + l4], t4 = tw2[8 + 4 * w + 2 * l4 + l5];
    const double2 t3 = tw2[16 + 8 * w + 4 * l4 + 2 * l3 + l2], t2 = tw2[32 + 16 * w + 8 * l4 + 4 * l3 + 2 * l2 + l5];
    const double2 t1 = tw2[64 + 32 * w + L31], t0 = tw2[128 + 64 * w + 2 * L31 + l5];
    double a0 = (double)x0, a1 = (double)x1;
    auto bfly = [&](double w_, double w2_) { const double v = tw_mul<FAST>(a1, w_, w2_, m), u = a0; a0 = u + v; a1 = u - v; };
    const bool up = l5;
    // exchange layouts: sigma_k permutes the low 6 position bits so that both the writing and the reading pattern of exchange k
    // touch 16 distinct 8-byte slots per 16-lane group
    auto sig1 = [](int p) { return p ^ (((p >> 4) & 3) << 2); };
    auto sig2 = [](int p) { return (p & 0xC0) | ((p & 3) << 4) | ((((p >> 2) ^ p) & 3) << 2) | ((p >> 4) & 3); };
    // ---- pass 0: stages 7, 6
    bfly(twA.w[1], twA.w2[1]);
    xchg<32>(a0, a1, up);                                // register = p6, l5 = p7
    bfly(t6.x, t6.y);
    {
        const int p = (l5 << 7) | (w << 5) | L31;        // + (r << 6)
        reg[p] = a0; reg[p | 64] = a1;
    }
    __syncthreads();
    // ---- pass 1: stages 5, 4: register = p5, l5 = p4, w = p7, l4 = p6, lane & 15 = p3..p0
    {
        const int p = (w << 7) | (l4 << 6) | (l5 << 4) | L15;
        a0 = reg[p]; a1 = reg[p | 32];
    }
    bfly(t5.x, t5.y);
    xchg<32>(a0, a1, up);                                // register = p4, l5 = p5
    bfly(t4.x, t4.y);
    __syncthreads();                                     // everyone has read exchange 0
    {
        const int p = (w << 7) | (l4 << 6) | (l5 << 5) | L15;
        reg[sig1(p)] = a0; reg[sig1(p | 16)] = a1;
    }
    __syncthreads();
    // ---- pass 2: stages 3, 2: register = p3, l5 = p2, w = p7, l4 = p6, l3 = p5, l2 = p4, l1 l0 = p1 p0
    {
        const int p = (w << 7) | (l4 << 6) | (l3 << 5) | (l2 << 4) | (l5 << 2) | (lane & 3);
        a0 = reg[sig1(p)]; a1 = reg[sig1(p | 8)];
    }
    bfly(t3.x, t3.y);
    xchg<32>(a0, a1, up);                                // register = p2, l5 = p3
    bfly(t2.x, t2.y);
    __syncthreads();
    {
        const int p = (w << 7) | (l4 << 6) | (l3 << 5) | (l2 << 4) | (l5 << 3) | (lane & 3);
        reg[sig2(p)] = a0; reg[sig2(p | 4)] = a1;
    }
    __syncthreads();
    // ---- pass 3: stages 1, 0: register = p1, l5 = p0, w = p7, lane & 31 = p6..p2
    {
        const int p = (w << 7) | (L31 << 2) | l5;
        a0 = reg[sig2(p)]; a1 = reg[sig2(p | 2)];
    }
    bfly(t1.x, t1.y);
    xchg<32>(a0, a1, up);                                // register = p0, l5 = p1
    bfly(t0.x, t0.y);
    if (active) {
        const int p = (w << 7) | (L31 << 2) | (l5 << 1);
        int2 o = make_int2((int)fz_cent(a0, m), (int)fz_cent(a1, m));
        __builtin_nontemporal_store(o.x, out + row * D + p);
        __builtin_nontemporal_store(o.y, out + row * D + p + 1);
    }
}

__global__ __launch_bounds__(1024) void empty_kernel() {}

uint64_t powmod(uint64_t b, uint64_t e, uint64_t q) {
    unsigned __int128 r = 1, x = b % q;
    while (e) { if (e & 1) r = (r * x) % q; x = (x * x) % q; e >>= 1; }
    return (uint64_t)r;
}
unsigned brev(unsigned i, int k) { unsigned r = 0; for (int b = 0; b < k; ++b) r |= ((i >> b) & 1u) << (k - 1 - b); return r; }

struct Timer {
    hipEvent_t a, b;
    Timer() { (void)hipEventCreate(&a); (void)hipEventCreate(&b); }
};

}  // namespace

template <typename F>
static double time_us(F launch, int reps, hipStream_t st, Timer &t) {
    for (int i = 0; i < 50; ++i) launch();                       // clocks up, code resident
    (void)hipStreamSynchronize(st);
    double best = 1e30;
    for (int pass = 0; pass < 3; ++pass) {
        (void)hipEventRecord(t.a, st);
        for (int i = 0; i < reps; ++i) launch();
        (void)hipEventRecord(t.b, st);
        (void)hipEventSynchronize(t.b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, t.a, t.b);
        if (ms * 1e3 / reps < best) best = ms * 1e3 / reps;
    }
    return best;
}

// the same launch timed the way rocprofv3 --kernel-trace (and bench.py's roofline.achieved) sees it: begin / end events bound to
// EVERY dispatch (hipExtLaunchKernelGGL), so each dispatch owns a completion signal and runs serialised; mean of `n` launches
template <typename F>
static double isolated_us(F launch_with_events, int n, hipStream_t st) {
    std::vector<hipEvent_t> ev(2 * n);
    for (auto &e : ev) (void)hipEventCreate(&e);
    for (int i = 0; i < 30; ++i) launch_with_events(ev[0], ev[1]);
    (void)hipStreamSynchronize(st);
    for (int i = 0; i < n; ++i) launch_with_events(ev[2 * i], ev[2 * i + 1]);
    (void)hipStreamSynchronize(st);
    double sum = 0;
    for (int i = 0; i < n; ++i) { float ms = 0; (void)hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]); sum += ms * 1e3; }
    for (auto &e : ev) (void)hipEventDestroy(e);
    return sum / n;
}

int main(int argc, char **argv) {
    const int reps_arg = argc > 1 ? atoi(argv[1]) : 300;
    const uint32_t q = 2147465729u, root = 3337519u;
    const int n = 256, k = 8;
    FzMod mod = fz_make_mod(q);
    std::vector<double> pairs(2 * n), tw(n);
    FzTwA twA;
    memset(&twA, 0, sizeof(twA));
    for (int i = 0; i < n; ++i) {
        const double w = (double)powmod(root, brev((unsigned)i, k), q);
        tw[i] = w;
        pairs[2 * i] = w;
        pairs[2 * i + 1] = w * mod.kq;
        if (i < 16) { twA.w[i] = w; twA.w2[i] = w * mod.kq; }
    }
    double *d_tw2, *d_twB;
    CHECK(hipMalloc((void **)&d_tw2, sizeof(double) * 2 * n));
    CHECK(hipMemcpy(d_tw2, pairs.data(), sizeof(double) * 2 * n, hipMemcpyHostToDevice));
    {   // per-lane table of the 16-per-lane kernel's contiguous pass (as fz_ctx_create builds it)
        const int L = n / 16, SB = k - 4, NE = 16 - (16 >> SB);
        std::vector<double> twB((size_t)NE * L * 2);
        for (int ls = 0; ls < SB; ++ls) {
            const int t = 1 << (SB - 1 - ls), ng = 16 / (2 * t), ebase = (16 >> SB) * ((1 << ls) - 1);
            for (int g = 0; g < ng; ++g)
                for (int b = 0; b < L; ++b) {
                    const double w = tw[(16 << ls) + b * ng + g];
                    twB[((size_t)(ebase + g) * L + b) * 2] = w;
                    twB[((size_t)(ebase + g) * L + b) * 2 + 1] = w * mod.kq;
                }
        }
        CHECK(hipMalloc((void **)&d_twB, twB.size() * sizeof(double)));
        CHECK(hipMemcpy(d_twB, twB.data(), twB.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    int occ16 = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ16, ntt_fwd16<8, true>, 256, 0));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const size_t POOL = 3ull << 29;                       // 1.5 GiB of inputs and as much of outputs: rotating operand sets
    int32_t *pool_in, *pool_out, *d_ref;
    CHECK(hipMalloc((void **)&pool_in, POOL));
    CHECK(hipMalloc((void **)&pool_out, POOL));
    {
        std::vector<int32_t> h(POOL / 4 / 64);
        uint64_t z = 20261003;
        for (auto &v : h) { z = z * 6364136223846793005ull + 1442695040888963407ull; v = (int32_t)((int64_t)((z >> 33) % q) - (int64_t)(q / 2)); }
        for (int c = 0; c < 64; ++c) CHECK(hipMemcpy((char *)pool_in + c * (POOL / 64), h.data(), POOL / 64, hipMemcpyHostToDevice));
    }
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    Timer t;
    int bad = 0;
    for (int cold = 0; cold < 2; ++cold)
    for (size_t B : {(size_t)4096, (size_t)8192, (size_t)16384, (size_t)32768, (size_t)65536, (size_t)262144}) {
        if (argc > 2 && B > (size_t)atoll(argv[2])) continue;
        if (!cold && B > 16384) continue;
        const size_t rowbytes = B * n * 4;
        const size_t sets = cold ? POOL / rowbytes : 1;
        const int reps = B >= 65536 ? reps_arg / 4 : reps_arg;
        const double bytes = 2048.0 * B;
        size_t kk = 0;
        auto in_ptr = [&]() { return (const int32_t *)((char *)pool_in + (kk % sets) * rowbytes); };
        auto out_ptr = [&]() { return (int32_t *)((char *)pool_out + (kk++ % sets) * rowbytes); };
        printf("# forward NTT, %zu rows of degree 256 (%.0f MiB in + out), %s, back-to-back launches, best of 3 passes of %d\n", B,
               bytes / 1048576.0, cold ? "COLD (operand sets rotate through 1.5 GiB pools)" : "warm (one operand set)", reps);
        CHECK(hipMalloc((void **)&d_ref, rowbytes));
        std::vector<int32_t> ref(B * n), got(B * n);
        {   // reference: the library kernels as fz_ntt_forward launches them
            hipLaunchKernelGGL((ntt_fwd4<8, true, 1, 8>), dim3((unsigned)((B + 7) / 8)), dim3(512), 0, st, (const int32_t *)pool_in, d_ref, B, (const double2 *)d_tw2, fz_tw4(twA), mod);
            CHECK(hipStreamSynchronize(st));
            CHECK(hipMemcpy(ref.data(), d_ref, rowbytes, hipMemcpyDeviceToHost));
            auto base = [&]() { const int32_t *i_ = in_ptr(); hipLaunchKernelGGL((ntt_fwd4<8, true, 1, 8>), dim3((unsigned)((B + 7) / 8)), dim3(512), 0, st, i_, out_ptr(), B,
                                                   (const double2 *)d_tw2, fz_tw4(twA), mod); };
            const double us = time_us(base, reps, st, t);
            printf("%-44s grid %6zu x %4d  %8.3f us  %5.1f %% of 8 TB/s\n", "library ntt_fwd4<8, NR=1, WAVES=8>", (B + 7) / 8, 512, us,
                   bytes / (us * 1e-6) / 8e12 * 100);
            const size_t tasks16 = (B * n + 1023) / 1024, blocks16 = (tasks16 + 3) / 4, cap16 = (size_t)occ16 * prop.multiProcessorCount;
            const unsigned g16 = (unsigned)(blocks16 < cap16 ? blocks16 : cap16);
            auto b16 = [&]() { const int32_t *i_ = in_ptr(); hipLaunchKernelGGL((ntt_fwd16<8, true>), dim3(g16), dim3(256), 0, st, i_, out_ptr(), B,
                                                   (const double2 *)d_twB, twA, mod); };
            const double us16 = time_us(b16, reps, st, t);
            printf("%-44s grid %6u x %4d  %8.3f us  %5.1f %% of 8 TB/s\n", "library ntt_fwd16<8> (16 per lane, persistent)", g16, 256, us16,
                   bytes / (us16 * 1e-6) / 8e12 * 100);
        }
#define VARIANT(NR, WAVES, TW)                                                                                                  \
        {                                                                                                                       \
            const size_t tasks = (B + NR - 1) / NR;                                                                             \
            const unsigned grid = (unsigned)((tasks + WAVES - 1) / WAVES);                                                      \
            auto f = [&]() { const int32_t *i_ = in_ptr(); hipLaunchKernelGGL((fwd4_variant<NR, WAVES, TW, true>), dim3(grid), dim3(64 * WAVES), 0, st, \
                                                i_, out_ptr(), B, (const double2 *)d_tw2, twA, mod); };                         \
            hipLaunchKernelGGL((fwd4_variant<NR, WAVES, TW, true>), dim3(grid), dim3(64 * WAVES), 0, st, (const int32_t *)pool_in, d_ref, B, \
                               (const double2 *)d_tw2, twA, mod);                                                               \
            CHECK(hipStreamSynchronize(st));                                                                                    \
            CHECK(hipMemcpy(got.data(), d_ref, rowbytes, hipMemcpyDeviceToHost));                                               \
            const bool ok = memcmp(got.data(), ref.data(), rowbytes) == 0;                                                      \
            if (!ok) ++bad;                                                                                                     \
            const double us = time_us(f, reps, st, t);                                                                          \
            auto fe = [&](hipEvent_t e0, hipEvent_t e1) { const int32_t *i_ = in_ptr(); hipExtLaunchKernelGGL((fwd4_variant<NR, WAVES, TW, true>), \
                        dim3(grid), dim3(64 * WAVES), 0, st, e0, e1, 0, i_, out_ptr(), B, (const double2 *)d_tw2, twA, mod); }; \
            const double iso = isolated_us(fe, 200, st);                                                                        \
            char name[64];                                                                                                      \
            snprintf(name, sizeof(name), "NR=%d WAVES=%-2d TW=%s", NR, WAVES, TW == 0 ? "global" : TW == 1 ? "lds" : "lds, barrier after pass 0"); \
            printf("%-44s grid %6u x %4d  %8.3f us  %5.1f %% of 8 TB/s   per-dispatch events %7.3f us  %5.1f %%   %s\n", name, grid, 64 * WAVES, us, \
                   bytes / (us * 1e-6) / 8e12 * 100, iso, bytes / (iso * 1e-6) / 8e12 * 100, ok ? "bit-exact" : "MISMATCH");    \
        }
#define XVARIANT(LABEL, KERNEL, GRID, BLOCK)                                                                                    \
        {                                                                                                                       \
            const unsigned grid = (unsigned)(GRID);                                                                             \
            auto f = [&]() { const int32_t *i_ = in_ptr(); hipLaunchKernelGGL((KERNEL), dim3(grid), dim3(BLOCK), 0, st,        \
                                                i_, out_ptr(), B, (const double2 *)d_tw2, twA, mod); };                         \
            hipLaunchKernelGGL((KERNEL), dim3(grid), dim3(BLOCK), 0, st, (const int32_t *)pool_in, d_ref, B,                   \
                               (const double2 *)d_tw2, twA, mod);                                                               \
            CHECK(hipStreamSynchronize(st));                                                                                    \
            CHECK(hipMemcpy(got.data(), d_ref, rowbytes, hipMemcpyDeviceToHost));                                               \
            const bool ok = memcmp(got.data(), ref.data(), rowbytes) == 0;                                                      \
            if (!ok) ++bad;                                                                                                     \
            const double us = time_us(f, reps, st, t);                                                                          \
            auto fe = [&](hipEvent_t e0, hipEvent_t e1) { const int32_t *i_ = in_ptr(); hipExtLaunchKernelGGL((KERNEL),        \
                        dim3(grid), dim3(BLOCK), 0, st, e0, e1, 0, i_, out_ptr(), B, (const double2 *)d_tw2, twA, mod); };      \
            const double iso = isolated_us(fe, 200, st);                                                                        \
            printf("%-44s grid %6u x %4d  %8.3f us  %5.1f %% of 8 TB/s   per-dispatch events %7.3f us  %5.1f %%   %s\n", LABEL, grid, BLOCK, us, \
                   bytes / (us * 1e-6) / 8e12 * 100, iso, bytes / (iso * 1e-6) / 8e12 * 100, ok ? "bit-exact" : "MISMATCH");    \
        }
        XVARIANT("xlane (register exchange) WAVES=1", (fwd4_xlane<1, true>), B, 64)
        XVARIANT("xlane (register exchange) WAVES=4", (fwd4_xlane<4, true>), (B + 3) / 4, 256)
        XVARIANT("xlane (register exchange) WAVES=8", (fwd4_xlane<8, true>), (B + 7) / 8, 512)
        XVARIANT("row over 2 waves, 1 row / workgroup", (fwd2_pair<1, true>), B, 128)
        XVARIANT("row over 2 waves, 2 rows / workgroup", (fwd2_pair<2, true>), (B + 1) / 2, 256)
        XVARIANT("row over 2 waves, 4 rows / workgroup", (fwd2_pair<4, true>), (B + 3) / 4, 512)
        VARIANT(1, 1, 0) VARIANT(1, 4, 0) VARIANT(1, 8, 0) VARIANT(1, 8, 1) VARIANT(1, 8, 2) VARIANT(1, 16, 2)
        VARIANT(2, 2, 0) VARIANT(2, 4, 0) VARIANT(2, 8, 2)
        VARIANT(4, 2, 0) VARIANT(4, 4, 0) VARIANT(4, 4, 2) VARIANT(4, 8, 0)
        VARIANT(8, 1, 0) VARIANT(8, 2, 0) VARIANT(8, 4, 0)
        {
            const size_t n16 = rowbytes / 16;
            auto c = [&]() { const int32_t *i_ = in_ptr(); hipLaunchKernelGGL(diag_copy_kernel, dim3((unsigned)((n16 + 63) / 64)), dim3(64), 0, st, (const int4 *)i_, (int4 *)out_ptr(), n16); };
            const double us = time_us(c, reps, st, t);
            auto ce = [&](hipEvent_t e0, hipEvent_t e1) { const int32_t *i_ = in_ptr(); hipExtLaunchKernelGGL(diag_copy_kernel, dim3((unsigned)((n16 + 63) / 64)), dim3(64), 0, st, e0, e1, 0, (const int4 *)i_, (int4 *)out_ptr(), n16); };
            const double iso = isolated_us(ce, 200, st);
            printf("%-44s grid %6zu x %4d  %8.3f us  %5.1f %% of 8 TB/s   per-dispatch events %7.3f us  %5.1f %%\n", "plain copy of the same bytes (16 B per lane)", (n16 + 63) / 64, 64, us,
                   bytes / (us * 1e-6) / 8e12 * 100, iso, bytes / (iso * 1e-6) / 8e12 * 100);
            for (int wv : {1, 4, 8}) {
                const unsigned grid = (unsigned)((B + wv - 1) / wv);
                auto e = [&]() { hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(64 * wv), 0, st); };
                auto ee = [&](hipEvent_t e0, hipEvent_t e1) { hipExtLaunchKernelGGL(empty_kernel, dim3(grid), dim3(64 * wv), 0, st, e0, e1, 0); };
                printf("empty dispatch, %u workgroups of %d threads: %.3f us back to back, %.3f us per-dispatch events\n", grid, 64 * wv, time_us(e, reps, st, t),
                       isolated_us(ee, 200, st));
            }
        }
        CHECK(hipFree(d_ref));
    }
    printf(bad ? "# %d variant(s) MISMATCHED\n" : "# all variants bit-exact against the library kernel\n", bad);
    return bad ? 2 : 0;
}
