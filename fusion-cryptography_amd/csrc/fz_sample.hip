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
    const bool two = key1 != 0;                                            // init_by_array key length 2 (else 1)
#define FZ_MT(i) mt[(i) * 64 + lane]
    // init_by_array, first loop (624 steps: i = 1..623, then i = 1 again after the wrap)
    uint32_t prev = init_tab[0];
    uint32_t j = 0;
    for (int i = 1; i < kMtN; ++i) {
        prev = (init_tab[i] ^ ((prev ^ (prev >> 30)) * 1664525u)) + (j ? key1 : key0) + j;
        FZ_MT(i) = prev;
        j = two ? (j ^ 1u) : 0u;
    }
    prev = (FZ_MT(1) ^ ((prev ^ (prev >> 30)) * 1664525u)) + (j ? key1 : key0) + j;      // mt[0] = mt[623] = prev
    FZ_MT(1) = prev;
    // second loop (623 steps: i = 2..623, then i = 1 after the wrap).  The words it reads are the first loop's: eight are
    // requested ahead of the eight dependent steps that use them (the compiler cannot move an LDS load above the previous
    // step's store by itself; one load latency per step would double the chain)
    constexpr int U = 8;
    for (int i0 = 2; i0 < kMtN; i0 += U) {
        uint32_t m[U];
#pragma unroll
        for (int u = 0; u < U; ++u) m[u] = FZ_MT(i0 + u < kMtN ? i0 + u : kMtN - 1);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (i0 + u < kMtN) {
                prev = (m[u] ^ ((prev ^ (prev >> 30)) * 1566083941u)) - (uint32_t)(i0 + u);
                FZ_MT(i0 + u) = prev;
            }
        }
    }
    prev = (FZ_MT(1) ^ ((prev ^ (prev >> 30)) * 1566083941u)) - 1u;
    FZ_MT(1) = prev;
    FZ_MT(0) = 0x80000000u;

    int t = 0;                               // coefficients finished
    uint32_t mag = 0;                        // 0: the next accepted draw is a magnitude; else its sign
    bool done = false;
    int32_t *row = out + p * (size_t)degree;
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
    if (!done && live) atomicOr(fail, 1);    // never seen: the caller falls back to the host sampler
#undef FZ_MT
}

}  // namespace

// state after init_genrand(19650218u): 624 words
void fz_mt_init_table(uint32_t *h_tab) {
    h_tab[0] = 19650218u;
    for (int i = 1; i < kMtN; ++i) h_tab[i] = 1812433253u * (h_tab[i - 1] ^ (h_tab[i - 1] >> 30)) + (uint32_t)i;
}

// d_seeds [nkeys] (device), d_out [2 * nkeys][degree]; d_init the table above; d_fail one int, zero before the launch
int fz_launch_mt_sample(fz_ctx *ctx, const unsigned long long *d_seeds, size_t nkeys, int degree, uint32_t bound, int kbits,
                        const uint32_t *d_init, int32_t *d_out, int *d_fail) {
    if (nkeys == 0) return FZ_OK;
    const size_t lds = (size_t)kMtN * 64 * 4;
    hipError_t e = hipFuncSetAttribute((const void *)mt_sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fz_check_hip(e, "sampler LDS attribute");
    const size_t npoly = 2 * nkeys;
    hipLaunchKernelGGL(mt_sample_kernel, dim3((unsigned)((npoly + 63) / 64)), dim3(64), lds, ctx->stream, d_seeds, npoly, degree, bound,
                       kbits, d_init, d_out, d_fail);
    return fz_check_hip(hipGetLastError(), "mt_sample launch");
}
