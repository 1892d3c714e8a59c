// fz_sample.hip -- the reference's seeded secret-key sampler on the device, bit for bit.
//
// keygen(params, seed) draws every entry of a secret matrix with sample_polynomial_coefficient_representation
// (algebra/polynomials.py:436-467) after random.seed(seed) (fusion.py:339-362: seed for the left matrix, seed + 1 for the
// right one), i.e. CPython's MT19937: init_by_array over the 32-bit words of the seed, getrandbits(k) = genrand >> (32 - k),
// randrange(n) by rejection (Modules/_randommodule.c, Lib/random.py _randbelow_with_getrandbits).  With weight bound =
// degree (both parameter sets) a polynomial is `degree` pairs (1 + randrange(bound)) * (1 - 2 * randrange(2)), no shuffle.
//
// One LANE per polynomial, 64 polynomials per one-wave workgroup, the 624-word generator states of the wave in LDS as
// [word][lane] (156 KiB: one wave per CU; every access is a conflict-free row).  Seeding is a chain of 1247 dependent
// steps per generator -- nothing to parallelise inside one seed, so the parallelism is the 64 lanes and the CUs.  The
// wave walks the OUTPUT STREAM in lockstep (every lane consumes its generator's r-th output in iteration r: the state
// position is wave-uniform, the state is regenerated for all lanes at once every 624 outputs); what differs per lane is
// only how far its polynomial has got (rejections), a small state machine in registers.
#include "fz_internal.h"
#include "../../include/fusion_hip.h"

namespace {

constexpr int kMtN = 624, kMtM = 397;
constexpr int kMaxGenerations = 16;          // 9984 outputs per polynomial; the expected need is ~3.3 per coefficient

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// init_by_array(key) on a lane per generator (Modules/_randommodule.c): the 1247 dependent steps that turn init_tab -- the state
// after init_genrand(19650218), the same for every seed -- and the seed's one or two 32-bit words into the generator's state,
// left in LDS as mt[word * 64 + lane] (word 0 = 0x80000000); out(i, value) sees every FINAL word i >= 1 as it is made.
// The chain is four dependent vector instructions per step (shift, xor, multiply, xor-add: 12.3 ns on a lone wave,
// tools/microbench/salu_chain.hip; the scalar unit is no faster: 15.4); what rounds 2-5's loops added to it were WAITS: the
// table words came by scalar loads used at once (a scalar wait cannot be partial: the loads return out of order and share a
// counter with LDS), the second loop read its LDS words where it needed them, and its stores were predicated (an exec-mask
// save, a branch and a restore per step) -- 41 ns per step.  Now every lane loads the table word itself (one address per
// wave: a broadcast) a CHUNK AHEAD -- vector loads return in order, the wait before a chunk leaves the next chunk's in
// flight --, the LDS reads run a chunk ahead too, and nothing in a step is conditional: 22 ns per step (mt_seed_kernel of
// 2048 generators 51.4 -> 27.7 us).
template <class OUT>
__device__ __forceinline__ void mt_seed_state(uint32_t *mt, const uint32_t *init_tab, uint32_t key0, uint32_t key1, int lane, OUT out) {
    constexpr int U = 8;                     // steps per chunk = the distance a request runs ahead of its use (16: 32.7 us against 27.7)
    const bool two = key1 != 0;              // init_by_array key length 2 (else 1)
#define FZ_MTS(i) mt[(i) * 64 + lane]
    uint32_t prev = init_tab[0];
    uint32_t j = 0;
    int voff = 0;
    asm volatile("" : "+v"(voff));           // (a vector register the compiler cannot prove uniform: vector loads)
    const uint32_t *vtab = init_tab + voff;
    auto load_tab = [&](uint32_t (&t)[U], int i0) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < U; ++u) t[u] = vtab[i0 + u < kMtN ? i0 + u : kMtN - 1];
    };
    // first loop (624 steps: i = 1..623, then i = 1 again after the wrap)
    auto steps1 = [&](const uint32_t (&t)[U], int i0) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (i0 + u < kMtN) {
                prev = (t[u] ^ ((prev ^ (prev >> 30)) * 1664525u)) + (j ? key1 : key0) + j;
                FZ_MTS(i0 + u) = prev;
                j = two ? (j ^ 1u) : 0u;
            }
        }
    };
    {
        uint32_t ta[U], tb[U];
        load_tab(ta, 1);
        int i0 = 1;
        for (; i0 + 2 * U < kMtN; i0 += 2 * U) {             // two chunks per trip: the buffers swap roles without a copy
            load_tab(tb, i0 + U);
            steps1(ta, i0);
            load_tab(ta, i0 + 2 * U);
            steps1(tb, i0 + U);
        }
        for (; i0 < kMtN; i0 += U) {                          // the tail (623 = 38 * 16 + 15 steps)
            if (i0 + U < kMtN) load_tab(tb, i0 + U);
            steps1(ta, i0);
#pragma unroll
            for (int u = 0; u < U; ++u) ta[u] = tb[u];
        }
    }
    prev = (FZ_MTS(1) ^ ((prev ^ (prev >> 30)) * 1664525u)) + (j ? key1 : key0) + j;      // mt[0] = mt[623] = prev
    FZ_MTS(1) = prev;
    // second loop (623 steps: i = 2..623, then i = 1 after the wrap); a chunk's reads are words the FIRST loop wrote (the
    // second loop's own stores go to words below the ones in flight)
    auto load_mt = [&](uint32_t (&w)[U], int i0) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < U; ++u) w[u] = FZ_MTS(i0 + u < kMtN ? i0 + u : kMtN - 1);
    };
    auto steps2 = [&](const uint32_t (&w)[U], int i0) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (i0 + u < kMtN) {
                prev = (w[u] ^ ((prev ^ (prev >> 30)) * 1566083941u)) - (uint32_t)(i0 + u);
                FZ_MTS(i0 + u) = prev;
                out(i0 + u, prev);
            }
        }
    };
    {
        uint32_t wa[U], wb[U];
        load_mt(wa, 2);
        int i0 = 2;
        for (; i0 + 2 * U < kMtN; i0 += 2 * U) {
            load_mt(wb, i0 + U);
            steps2(wa, i0);
            load_mt(wa, i0 + 2 * U);
            steps2(wb, i0 + U);
        }
        for (; i0 < kMtN; i0 += U) {
            if (i0 + U < kMtN) load_mt(wb, i0 + U);
            steps2(wa, i0);
#pragma unroll
            for (int u = 0; u < U; ++u) wa[u] = wb[u];
        }
    }
    prev = (FZ_MTS(1) ^ ((prev ^ (prev >> 30)) * 1566083941u)) - 1u;
    FZ_MTS(1) = prev;
    FZ_MTS(0) = 0x80000000u;
    out(1, prev);
#undef FZ_MTS
}

// init_tab: the state after init_genrand(19650218), the same for every seed
__global__ __launch_bounds__(64) void mt_sample_kernel(const unsigned long long *seeds, size_t npoly, int degree, uint32_t bound,
                                                       int kbits, const uint32_t *__restrict__ init_tab, int32_t *out, int *fail) {
    extern __shared__ __attribute__((aligned(16))) uint32_t mt[];          // [624][64]
    const int lane = threadIdx.x & 63;
    const size_t p_raw = (size_t)blockIdx.x * 64 + lane;
    const bool live = p_raw < npoly;
    const size_t p = live ? p_raw : npoly - 1;                             // idle lanes shadow the last polynomial
    const unsigned long long seed = seeds[p >> 1] + (unsigned long long)(p & 1);      // left: seed, right: seed + 1
    const uint32_t key0 = (uint32_t)seed, key1 = (uint32_t)(seed >> 32);
#define FZ_MT(i) mt[(i) * 64 + lane]
    mt_seed_state(mt, init_tab, key0, key1, lane, [](int, uint32_t) {});

    int t = 0;                               // coefficients finished
    uint32_t mag = 0;                        // 0: the next accepted draw is a magnitude; else its sign
    bool done = false;
    int32_t *row = out + p * (size_t)degree;
    constexpr int U = 8;
    static_assert(kMtN % U == 0, "chunks of 8 words");
    for (int gen = 0; gen < kMaxGenerations; ++gen) {
        // regenerate all 624 words in place (genrand_uint32's refill), every lane its own column.  Word k needs the OLD
        // words k and k + 1 and word k + 397 (old below k = 227, else the new word k - 227, written 227 steps earlier; the
        // last step reads the new word 0): a chunk of 8 steps can read all its 17 inputs before its first store.
        for (int k0 = 0; k0 < kMtN; k0 += U) {
            uint32_t a[U + 1], b[U];
#pragma unroll
            for (int u = 0; u <= U; ++u) a[u] = FZ_MT(k0 + u < kMtN ? k0 + u : 0);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int kk = k0 + u + kMtM;
                b[u] = FZ_MT(kk < kMtN ? kk : kk - kMtN);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t y = (a[u] & 0x80000000u) | (a[u + 1] & 0x7fffffffu);
                FZ_MT(k0 + u) = b[u] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
        }
        for (int r0 = 0; r0 < kMtN; r0 += U) {
            uint32_t ys[U];
#pragma unroll
            for (int u = 0; u < U; ++u) ys[u] = FZ_MT(r0 + u);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t y = mt_temper(ys[u]);
                const uint32_t v = y >> (32 - kbits), sg = y >> 30;
                // getrandbits(kbits) kept if below the bound, then randrange(2) = getrandbits(2) kept if < 2
                const bool take_sign = !done && mag != 0 && sg < 2u;
                const bool take_mag = !done && mag == 0 && v < bound;
                if (take_sign) {
                    if (live) row[t] = sg ? -(int32_t)mag : (int32_t)mag;          // (1 + r1) * (1 - 2 * r2)
                    ++t;
                    done = t == degree;
                }
                mag = take_sign ? 0u : (take_mag ? v + 1u : mag);
            }
            if ((r0 & 8) && __all(done)) return;
        }
    }
    if (!done && live) *(volatile int *)fail = 1;      // never seen: the caller falls back to the host sampler (a plain store: the flag may sit in pinned host memory)
#undef FZ_MT
}

// ---------------------------------------------------------------------------------------------------------------
// The same sampler in two kernels (the default up to 4096 keys per call since round 3: fz_sample_secret_polys_dev).  Only the SEEDING is a chain nothing can shorten (1247
// dependent steps per generator): mt_seed_kernel keeps it on a lane per generator as above and writes the finished states
// out word-major.  Everything after it -- regenerating the 624 words, tempering, the rejection sampling -- is parallel
// over the words except for a two-state automaton along the output stream (is the next accepted draw a magnitude or a
// sign?), so mt_draw_kernel gives every generator a WAVE: a thousand keys are 2048 waves on the whole chip instead of 32
// waves on 32 CUs (a generator state fills a lane's share of one CU's LDS in the one-kernel form).
//   regeneration: word k needs the old words k, k + 1 and word k + 397 -- old below k = 227, else the NEW word k - 227: three
//                 chunks (0..226, 227..453, 454..623), each read completely before it is written;
//   outputs:      lane L owns outputs 10 L .. 10 L + 9 of the generation.  It walks them twice -- entering in state "magnitude
//                 next" and in state "sign next" -- which makes its ten outputs a map (pending in) -> (pending out, coefficients
//                 completed); the maps compose, so the lanes' true entry states are an exclusive scan over the wave (six shuffle
//                 steps; rounds 3-5: lane 0 walked the 64 summaries through LDS); every lane then walks its outputs once more
//                 from its true entry state and writes its coefficients at their final positions.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kDrawWaves = 4;
constexpr int kPerLane = 10;                 // 64 x 10 >= 624 outputs of a generation

__global__ __launch_bounds__(64) void mt_seed_kernel(const unsigned long long *seeds, size_t npoly, const uint32_t *__restrict__ init_tab,
                                                     uint32_t *state) {
    extern __shared__ __attribute__((aligned(16))) uint32_t mt[];          // [624][64]: the first loop's words, read back by the second
    const int lane = threadIdx.x & 63;
    const size_t p_raw = (size_t)blockIdx.x * 64 + lane;
    const bool live = p_raw < npoly;
    const size_t p = live ? p_raw : npoly - 1;
    const unsigned long long seed = seeds[p >> 1] + (unsigned long long)(p & 1);      // left: seed, right: seed + 1
    const uint32_t key0 = (uint32_t)seed, key1 = (uint32_t)(seed >> 32);
    const uint32_t p32 = (uint32_t)p;                                      // (npoly < 2^32: a 32-bit lane offset beside a scalar row base)
    // final words go out as they are made, word-major (64 generators = one 256-byte store; the draw kernel gathers its generator's
    // column -- 624 sectors per wave from the L2 -- rather than this kernel transposing 160 KB through LDS on 32 waves: 27 us of
    // its 72 in round 3).  Unconditionally: an idle lane shadows the LAST polynomial and stores that polynomial's own word a second time.
    mt_seed_state(mt, init_tab, key0, key1, lane, [&](int i, uint32_t v) { (state + (size_t)i * npoly)[p32] = v; });
    state[p32] = 0x80000000u;
}

__global__ __launch_bounds__(64 * kDrawWaves) void mt_draw_kernel(const uint32_t *__restrict__ state, size_t npoly, int degree, uint32_t bound,
                                                                  int kbits, int32_t *out, int *fail) {
    __shared__ uint32_t s_mt[kDrawWaves][kMtN + 16];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const size_t gen_id = (size_t)blockIdx.x * kDrawWaves + wave;
    if (gen_id >= npoly) return;                          // whole waves only: the waves of a workgroup never synchronise
    uint32_t *mt = s_mt[wave];
    auto wsync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    for (int i = lane; i < kMtN; i += 64) mt[i] = state[(size_t)i * npoly + gen_id];      // word-major: see mt_seed_kernel
    wsync();
    int32_t *row = out + gen_id * (size_t)degree;
    int t = 0;                   // coefficients finished before this generation (wave-uniform)
    uint32_t carry_mag = 0;      // magnitude waiting for its sign at the start of this generation (0: none; wave-uniform)
    const int shift = 32 - kbits;
    for (int gen = 0; gen < kMaxGenerations; ++gen) {
        // ---- regenerate the 624 words in place, three chunks ----
        for (int c = 0; c < 3; ++c) {
            const int lo = c == 0 ? 0 : (c == 1 ? kMtN - kMtM : 2 * (kMtN - kMtM)), hi = c == 0 ? kMtN - kMtM : (c == 1 ? 2 * (kMtN - kMtM) : kMtN - 1);
            uint32_t nw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = lo + lane + 64 * u;
                if (k < hi) {
                    const uint32_t y = (mt[k] & 0x80000000u) | (mt[k + 1] & 0x7fffffffu);
                    const uint32_t m397 = mt[k < kMtN - kMtM ? k + kMtM : k - (kMtN - kMtM)];     // old below 227, else the new word k - 227
                    nw[u] = m397 ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
                }
            }
            wsync();
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = lo + lane + 64 * u;
                if (k < hi) mt[k] = nw[u];
            }
            wsync();
        }
        if (lane == 0) {         // the last word reads the NEW word 0
            const uint32_t y = (mt[kMtN - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
            mt[kMtN - 1] = mt[kMtM - 1] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        wsync();
        // ---- this lane's outputs ----
        uint32_t v[kPerLane], sg[kPerLane];
        const int base = lane * kPerLane;
#pragma unroll
        for (int u = 0; u < kPerLane; ++u) {
            const int i = base + u;
            const uint32_t y = mt_temper(mt[i < kMtN ? i : kMtN - 1]);
            v[u] = i < kMtN ? (y >> shift) : 0xffffffffu;            // past the end: neither a magnitude nor a sign
            sg[u] = i < kMtN ? (y >> 30) : 3u;
        }
        // walk from both entry states: state 0 = magnitude next, else the pending magnitude (> 0)
        auto walk = [&](uint32_t mag, uint32_t &cnt) {
            cnt = 0;
#pragma unroll
            for (int u = 0; u < kPerLane; ++u) {
                const bool take_sign = mag != 0 && sg[u] < 2u, take_mag = mag == 0 && v[u] < bound;
                cnt += take_sign ? 1u : 0u;
                mag = take_sign ? 0u : (take_mag ? v[u] + 1u : mag);
            }
            return mag;
        };
        uint32_t cm, cs;
        const uint32_t em = walk(0u, cm);                // entered expecting a magnitude: exit pending magnitude (0: none)
        const uint32_t es = walk(0xffffffffu, cs);       // entered with SOME magnitude pending: the value is patched below
        // entering with a pending magnitude P: the walk is the same until the first sign is taken; if none is taken the exit
        // still holds P (es == 0xffffffff marks that).
        // A lane's ten outputs are thus a map (pending in) -> (pending out, coefficients completed) given by (em, cm, es, cs), and
        // such maps compose: the lanes' true entry states are an exclusive scan of them under composition -- six shuffle steps
        // of the wave instead of lane 0 walking the 64 summaries through LDS (rounds 3-5: 64 dependent LDS round trips per
        // generation, a third of this kernel).  compose(A, B) = A's lanes first, then B's.
        uint32_t a_em = em, a_cm = cm, a_es = es, a_cs = cs;         // inclusive: lanes 0 .. lane composed
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t p_em = __shfl_up(a_em, off), p_cm = __shfl_up(a_cm, off), p_es = __shfl_up(a_es, off), p_cs = __shfl_up(a_cs, off);
            if (lane >= off) {
                const uint32_t n_em = p_em == 0u ? a_em : (a_es == 0xffffffffu ? p_em : a_es);
                const uint32_t n_cm = p_cm + (p_em == 0u ? a_cm : a_cs);
                const uint32_t n_es = p_es == 0xffffffffu ? a_es : (p_es == 0u ? a_em : (a_es == 0xffffffffu ? p_es : a_es));
                const uint32_t n_cs = p_cs + (p_es == 0u ? a_cm : a_cs);
                a_em = n_em; a_cm = n_cm; a_es = n_es; a_cs = n_cs;
            }
        }
        auto apply = [&](uint32_t s_em, uint32_t s_cm, uint32_t s_es, uint32_t s_cs, uint32_t &pend, int &tt) {
            if (carry_mag == 0u) { pend = s_em; tt = t + (int)s_cm; }
            else { pend = s_es == 0xffffffffu ? carry_mag : s_es; tt = t + (int)s_cs; }
        };
        uint32_t mag = carry_mag;                        // this lane's true entry state: the lanes before it applied to the wave's
        int tl = t;
        {
            const uint32_t e_em = __shfl_up(a_em, 1), e_cm = __shfl_up(a_cm, 1), e_es = __shfl_up(a_es, 1), e_cs = __shfl_up(a_cs, 1);
            if (lane > 0) apply(e_em, e_cm, e_es, e_cs, mag, tl);
        }
        uint32_t next_mag;                               // ... and the wave's exit state: all 64 lanes applied
        int next_t;
        apply(__shfl(a_em, 63), __shfl(a_cm, 63), __shfl(a_es, 63), __shfl(a_cs, 63), next_mag, next_t);
        {
#pragma unroll
            for (int u = 0; u < kPerLane; ++u) {
                const bool take_sign = mag != 0 && sg[u] < 2u, take_mag = mag == 0 && v[u] < bound;
                if (take_sign) {
                    if (tl < degree) row[tl] = sg[u] ? -(int32_t)mag : (int32_t)mag;          // (1 + r1) * (1 - 2 * r2)
                    ++tl;
                }
                mag = take_sign ? 0u : (take_mag ? v[u] + 1u : mag);
            }
        }
        carry_mag = next_mag;
        t = next_t;
        wsync();
        if (t >= degree) return;
    }
    if (lane == 0) *(volatile int *)fail = 1;        // never seen: the caller falls back to the host sampler (a plain store: the flag may sit in pinned host memory)
}

}  // namespace

// state after init_genrand(19650218u): 624 words
void fz_mt_init_table(uint32_t *h_tab) {
    h_tab[0] = 19650218u;
    for (int i = 1; i < kMtN; ++i) h_tab[i] = 1812433253u * (h_tab[i - 1] ^ (h_tab[i - 1] >> 30)) + (uint32_t)i;
}

// d_seeds [nkeys] (device), d_out [2 * nkeys][degree]; d_init the table above; d_fail one int, zero before the launch
int fz_launch_mt_sample(fz_ctx *ctx, const unsigned long long *d_seeds, size_t nkeys, int degree, uint32_t bound, int kbits,
                        const uint32_t *d_init, int32_t *d_out, int *d_fail, uint32_t *d_state) {
    if (nkeys == 0) return FZ_OK;
    const size_t lds = (size_t)kMtN * 64 * 4;
    const size_t npoly = 2 * nkeys;
    if (!d_state) {            // more than 4096 keys per call: the lane-per-polynomial kernel of round 2
        hipError_t e = hipFuncSetAttribute((const void *)mt_sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fz_check_hip(e, "sampler LDS attribute");
        hipLaunchKernelGGL(mt_sample_kernel, dim3((unsigned)((npoly + 63) / 64)), dim3(64), lds, ctx->stream, d_seeds, npoly, degree,
                           bound, kbits, d_init, d_out, d_fail);
        return fz_check_hip(hipGetLastError(), "mt_sample launch");
    }
    hipError_t e = hipFuncSetAttribute((const void *)mt_seed_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fz_check_hip(e, "sampler LDS attribute");
    hipLaunchKernelGGL(mt_seed_kernel, dim3((unsigned)((npoly + 63) / 64)), dim3(64), lds, ctx->stream, d_seeds, npoly, d_init, d_state);
    hipLaunchKernelGGL(mt_draw_kernel, dim3((unsigned)((npoly + kDrawWaves - 1) / kDrawWaves)), dim3(64 * kDrawWaves), 0, ctx->stream,
                       (const uint32_t *)d_state, npoly, degree, bound, kbits, d_out, d_fail);
    return fz_check_hip(hipGetLastError(), "mt_seed / mt_draw launch");
}
