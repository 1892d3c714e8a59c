// ntt_structures.hip -- which transform structure is faster when memory is NOT the limit?
//
// The fused verification and key-generation kernels are bound by vector issue, not by HBM: every row of
// sigma / of the secret costs one transform, and the rows arrive faster than the SIMDs transform them.  They use the radix-4
// structure (4 coefficients per lane: three LDS exchanges per row of degree 256).  The stand-alone kernels for large batches use
// 16 coefficients per lane (one exchange per row, four rows per wave): fewer LDS operations and synchronisations per butterfly,
// twice the registers.  This harness times the INVERSE transform of both structures with nothing but registers and LDS in the
// loop -- each wave transforms its own rows R times, feeding outputs back as inputs, keeps a running max |x| as the fused
// verification does, and stores one value at the end -- at 1 .. 8 waves per SIMD, so the figure is transforms per microsecond
// for the whole chip when vector issue and LDS are all that count.  Each workgroup also reads the shader clock (s_memtime)
// against the 100 MHz reference (s_memrealtime): the chip does not hold its 2.4 GHz under this load.
//
// It compiles the library's own kernel source into this translation unit: inv4_passes_n IS the shipped radix-4 code; the
// 16-per-lane loop body is the shipped ntt_inv16's, between its load and its store.   usage: ntt_structures [R=200]
#include "../../fusion-cryptography_amd/csrc/fz_ntt.hip"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

int fz_set_error(int code, const char *, ...) { return code; }
int fz_check_hip(hipError_t e, const char *what) {
    if (e != hipSuccess) { printf("HIP error in %s: %s\n", what, hipGetErrorString(e)); return FZ_E_HIP; }
    return FZ_OK;
}
int fz_verify_scratch(fz_ctx *, size_t, size_t, double **, int **) { return FZ_E_UNSUPPORTED; }

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

namespace {

// radix-4: NR row groups per wave and iteration (degree 256: one row per group)
template <int NR, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void loop_inv4(double *out, unsigned long long *stamps, int R, const double2 *__restrict__ itw2, FzTwA twA, FzMod m) {
    constexpr int LOGD = 8, P = LOGD / 2;
    __shared__ __attribute__((aligned(16))) double lds[WAVES * NR * 256];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    double *region = lds + wave * NR * 256;
    double2 twl[P - 1][3];
    inv4_load_twiddles<LOGD>(twl, itw2, lane);
    double a[NR][4];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) a[r][k] = (double)((lane * 4 + k + 977 * r + (int)blockIdx.x) % 1000003 - 500000);
    double mx = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < R; ++it) {
        inv4_passes_n<LOGD, true, NR>(a, region, twl, twA, m, lane);
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k) mx = fmax(mx, fabs(a[r][k]));
        wave_sync();      // the next rows' first-pass writes must not overtake these rows' last reads (as in verify_fused)
    }
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0; stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0; }
    if (mx == 12345.5) out[blockIdx.x * blockDim.x + threadIdx.x] = mx + a[0][0];
}

// 16 per lane: four rows per wave and iteration
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void loop_inv16(double *out, unsigned long long *stamps, int R, const double2 *__restrict__ itwB, FzTwA twA, FzMod m) {
    using G = Geom<8>;
    constexpr int L = G::L, PPW = G::PPW, SB = G::SB, NE = G::NE, PS = G::PS;
    constexpr int REGION = PPW * PS;
    __shared__ __attribute__((aligned(16))) double lds[WAVES * REGION + 2 * NE * L];
    double2 *s_tw = reinterpret_cast<double2 *>(lds + WAVES * REGION);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int p = lane / L, r = lane % L;
    for (int i = threadIdx.x; i < NE * L; i += 64 * WAVES) s_tw[i] = itwB[i];
    __syncthreads();
    double *row = lds + wave * REGION + p * PS;
    double a[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = (double)((lane * 16 + k + (int)blockIdx.x) % 1000003 - 500000);
    double mx = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < R; ++it) {
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
                a[k + t] = tw_mul<true>(u - v, w.x, w.y, m);
            }
        }
        a[0] = fz_fold(a[0], m);
        {
            double2 *blk = reinterpret_cast<double2 *>(row + 18 * r);
#pragma unroll
            for (int k = 0; k < 8; ++k) blk[k] = make_double2(a[2 * k], a[2 * k + 1]);
        }
        wave_sync();
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = row[pad16(r + L * k)];
        wave_sync();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int tk = 1 << s;
            const int h = 8 >> s;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (k & tk) continue;
                const double u = a[k], v = a[k + tk];
                if (s == 3) {
                    a[k] = tw_mul<true>(u + v, twA.n_inv, twA.n_inv2, m);
                    a[k + tk] = tw_mul<true>(u - v, twA.w1_n_inv, twA.w1_n_inv2, m);
                } else {
                    const int e = h + (k >> (s + 1));
                    a[k] = u + v;
                    a[k + tk] = tw_mul<true>(u - v, twA.w[e], twA.w2[e], m);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) mx = fmax(mx, fabs(a[k]));
        // outputs are in the strided layout (element r + L*k); the next iteration's contiguous pass wants 16 consecutive
        // elements per lane: in a fused kernel the NEXT rows come from memory in that layout, so no exchange belongs here
    }
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0; stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0; }
    if (mx == 12345.5) out[blockIdx.x * blockDim.x + threadIdx.x] = mx + a[0];
}

uint64_t powmod(uint64_t b, uint64_t e, uint64_t q) {
    unsigned __int128 r = 1, x = b % q;
    while (e) { if (e & 1) r = (r * x) % q; x = (x * x) % q; e >>= 1; }
    return (uint64_t)r;
}
unsigned brev(unsigned i, int k) { unsigned r = 0; for (int b = 0; b < k; ++b) r |= ((i >> b) & 1u) << (k - 1 - b); return r; }

}  // namespace

template <typename F>
static double time_us(F launch, hipStream_t st) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) launch();
    (void)hipStreamSynchronize(st);
    double best = 1e30;
    for (int pass = 0; pass < 3; ++pass) {
        (void)hipEventRecord(a, st);
        launch();
        (void)hipEventRecord(b, st);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        if (ms * 1e3 < best) best = ms * 1e3;
    }
    return best;
}

int main(int argc, char **argv) {
    const int R = argc > 1 ? atoi(argv[1]) : 200;
    const uint32_t q = 2147465729u, root = 3337519u;
    const int n = 256, k = 8;
    FzMod mod = fz_make_mod(q);
    const uint64_t inv_root = powmod(root, q - 2, q);
    std::vector<double> itw(n), pairs(2 * n);
    FzTwA twA;
    memset(&twA, 0, sizeof(twA));
    for (int i = 0; i < n; ++i) {
        itw[i] = (double)powmod(inv_root, brev((unsigned)i, k), q);
        pairs[2 * i] = itw[i];
        pairs[2 * i + 1] = itw[i] * mod.kq;
        if (i < 16) { twA.w[i] = itw[i]; twA.w2[i] = itw[i] * mod.kq; }
    }
    const uint64_t n_inv = powmod(n, q - 2, q);
    twA.n_inv = (double)n_inv;
    twA.n_inv2 = twA.n_inv * mod.kq;
    twA.w1_n_inv = (double)((unsigned __int128)(uint64_t)itw[1] * n_inv % q);
    twA.w1_n_inv2 = twA.w1_n_inv * mod.kq;
    double *d_itw2, *d_itwB, *d_out;
    CHECK(hipMalloc((void **)&d_itw2, sizeof(double) * 2 * n));
    CHECK(hipMemcpy(d_itw2, pairs.data(), sizeof(double) * 2 * n, hipMemcpyHostToDevice));
    {   // per-lane inverse table of the 16-per-lane contiguous pass: any valid twiddles do for timing
        const int L = n / 16, NE = 15;
        std::vector<double> twB((size_t)NE * L * 2);
        for (size_t i = 0; i < twB.size() / 2; ++i) { twB[2 * i] = itw[(i % (n - 1)) + 1]; twB[2 * i + 1] = twB[2 * i] * mod.kq; }
        CHECK(hipMalloc((void **)&d_itwB, twB.size() * sizeof(double)));
        CHECK(hipMemcpy(d_itwB, twB.data(), twB.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    CHECK(hipMalloc((void **)&d_out, sizeof(double) * 64 * 8 * 8192));
    unsigned long long *d_stamps;
    CHECK(hipMalloc((void **)&d_stamps, sizeof(unsigned long long) * 2 * 8192));
    std::vector<unsigned long long> h_stamps(2 * 8192);
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    printf("# inverse transforms of degree 256 from registers and LDS only (no global traffic in the loop), R = %d iterations per wave,\n"
           "# %d CUs; rows/us = transforms per microsecond for the whole chip; the fused verification kernel needs 83 per aggregate\n", R, cus);
    printf("%-34s %6s %10s %12s %10s %8s %12s\n", "structure", "w/SIMD", "waves", "time us", "rows/us", "MHz", "cyc/row/SIMD");
#define RUN(NAME, KERNEL, WAVES, ROWS_PER_WAVE_ITER, TABLE)                                                                    \
    for (int wps : {1, 2, 4, 6, 8}) {                                                                                           \
        int occ = 0;                                                                                                            \
        CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, KERNEL, 64 * WAVES, 0));                                       \
        const int blocks_per_cu = wps * 4 / WAVES;                                                                              \
        if (blocks_per_cu < 1 || blocks_per_cu > occ) continue;                                                                 \
        const int grid = cus * blocks_per_cu;                                                                                   \
        const double us = time_us([&]() { hipLaunchKernelGGL(KERNEL, dim3(grid), dim3(64 * WAVES), 0, st, d_out, d_stamps, R, (const double2 *)TABLE, twA, mod); }, st); \
        CHECK(hipStreamSynchronize(st));                                                                                        \
        CHECK(hipMemcpy(h_stamps.data(), d_stamps, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost));              \
        double mhz = 0;                                                                                                         \
        for (int b = 0; b < grid; ++b) mhz += 100.0 * (double)h_stamps[2 * b] / (double)(h_stamps[2 * b + 1] ? h_stamps[2 * b + 1] : 1); \
        mhz /= grid;                                                                                                            \
        const double rows_us = (double)grid * WAVES * R * ROWS_PER_WAVE_ITER / us;                                              \
        printf("%-34s %6d %10d %12.1f %10.1f %8.0f %12.0f\n", NAME, wps, grid * WAVES, us, rows_us, mhz, mhz * cus * 4 / rows_us); \
    }
    RUN("radix-4, 1 row per wave", (loop_inv4<1, 4>), 4, 1, d_itw2)
    RUN("radix-4, 2 rows per wave", (loop_inv4<2, 4>), 4, 2, d_itw2)
    RUN("radix-4, 4 rows per wave", (loop_inv4<4, 4>), 4, 4, d_itw2)
    RUN("16 per lane, 4 rows per wave", (loop_inv16<4>), 4, 4, d_itwB)
    CHECK(hipStreamSynchronize(st));
    printf("# (occupancy limits which waves-per-SIMD lines appear)\n");
    return 0;
}
