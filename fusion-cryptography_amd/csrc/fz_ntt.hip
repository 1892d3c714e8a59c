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

namespace {

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
// forward: strided pass -> transpose -> contiguous pass
// ------------------------------------------------------------------------------------------
template <int LOGD>
__global__ __launch_bounds__(64) void ntt_fwd16(const int32_t *in, int32_t *out, size_t batch,
                                                const double *__restrict__ twB, FzTwA twA, FzMod m) {
    using G = Geom<LOGD>;
    constexpr int D = G::D, L = G::L, PPW = G::PPW, SB = G::SB, NE = G::NE, PS = G::PS;
    __shared__ __attribute__((aligned(16))) double lds[PPW * PS + NE * L];
    double *s_tw = lds + PPW * PS;

    const int lane = threadIdx.x;
    const int p = lane / L, r = lane % L;
    for (int i = lane; i < NE * L; i += 64) s_tw[i] = twB[i];
    double *row = lds + p * PS;
    __syncthreads();

    const size_t tasks = (batch + PPW - 1) / PPW;
    for (size_t task = blockIdx.x; task < tasks; task += gridDim.x) {
        const size_t poly = task * PPW + p;
        const bool valid = poly < batch;
        const int32_t *src = in + poly * D + r;
        double a[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = valid ? (double)src[k * L] : 0.0;

        // strided pass: a 16-point LN transform over k with table entries 1..15
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int tk = 8 >> s;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (k & tk) continue;
                const double w = twA.w[(1 << s) + (k >> (4 - s))];
                const double v = fz_mulmod(a[k + tk], w, m);
                const double u = a[k];
                a[k] = u + v;
                a[k + tk] = u - v;
            }
        }

        // transpose: element j = r + L*k  ->  lane j/16, register j%16
#pragma unroll
        for (int k = 0; k < 16; ++k) row[pad16(r + L * k)] = a[k];
        __syncthreads();
        {
            const double2 *blk = reinterpret_cast<const double2 *>(row + 18 * r);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                double2 t = blk[k];
                a[2 * k] = t.x;
                a[2 * k + 1] = t.y;
            }
        }

        // contiguous pass: stages with distance 2^(SB-1) .. 1, per-lane twiddles
#pragma unroll
        for (int ls = 0; ls < SB; ++ls) {
            const int t = 1 << (SB - 1 - ls);
            const int ebase = (16 >> SB) * ((1 << ls) - 1);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (k & t) continue;
                const int g = k >> (SB - ls);
                const double w = s_tw[(ebase + g) * L + r];
                const double v = fz_mulmod(a[k + t], w, m);
                const double u = a[k];
                a[k] = u + v;
                a[k + t] = u - v;
            }
        }

        if (valid) {
            int4 *dst = reinterpret_cast<int4 *>(out + poly * D + 16 * r);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int4 o;
                o.x = (int)fz_cent(a[4 * k + 0], m);
                o.y = (int)fz_cent(a[4 * k + 1], m);
                o.z = (int)fz_cent(a[4 * k + 2], m);
                o.w = (int)fz_cent(a[4 * k + 3], m);
                dst[k] = o;
            }
        }
        __syncthreads();   // LDS rows are rewritten by the next task
    }
}

// ------------------------------------------------------------------------------------------
// inverse: contiguous pass -> transpose -> strided pass (n^{-1} folded into the last stage)
// ------------------------------------------------------------------------------------------
template <int LOGD>
__global__ __launch_bounds__(64) void ntt_inv16(const int32_t *in, int32_t *out, size_t batch,
                                                const double *__restrict__ itwB, FzTwA twA, FzMod m) {
    using G = Geom<LOGD>;
    constexpr int D = G::D, L = G::L, PPW = G::PPW, SB = G::SB, NE = G::NE, PS = G::PS;
    __shared__ __attribute__((aligned(16))) double lds[PPW * PS + NE * L];
    double *s_tw = lds + PPW * PS;

    const int lane = threadIdx.x;
    const int p = lane / L, r = lane % L;
    for (int i = lane; i < NE * L; i += 64) s_tw[i] = itwB[i];
    double *row = lds + p * PS;
    __syncthreads();

    const size_t tasks = (batch + PPW - 1) / PPW;
    for (size_t task = blockIdx.x; task < tasks; task += gridDim.x) {
        const size_t poly = task * PPW + p;
        const bool valid = poly < batch;
        double a[16];
        {
            const int4 *src = reinterpret_cast<const int4 *>(in + poly * D + 16 * r);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int4 t = valid ? src[k] : make_int4(0, 0, 0, 0);
                a[4 * k + 0] = (double)t.x;
                a[4 * k + 1] = (double)t.y;
                a[4 * k + 2] = (double)t.z;
                a[4 * k + 3] = (double)t.w;
            }
        }

        // contiguous pass: GS stages with distance 1, 2, .. 2^(SB-1)
#pragma unroll
        for (int ls = 0; ls < SB; ++ls) {
            const int t = 1 << ls;
            const int ebase = 16 - (16 >> ls);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (k & t) continue;
                const int g = k >> (ls + 1);
                const double w = s_tw[(ebase + g) * L + r];
                const double u = a[k], v = a[k + t];
                a[k] = u + v;
                a[k + t] = fz_mulmod(u - v, w, m);
            }
        }

        // transpose back to the strided layout
        {
            double2 *blk = reinterpret_cast<double2 *>(row + 18 * r);
#pragma unroll
            for (int k = 0; k < 8; ++k) blk[k] = make_double2(a[2 * k], a[2 * k + 1]);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = row[pad16(r + L * k)];

        // strided pass: GS stages with distance L, 2L, 4L, 8L; uniform twiddles
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int tk = 1 << s;
            const int h = 8 >> s;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (k & tk) continue;
                const double u = a[k], v = a[k + tk];
                if (s == 3) {
                    a[k] = fz_mulmod(u + v, twA.n_inv, m);
                    a[k + tk] = fz_mulmod(u - v, twA.w1_n_inv, m);
                } else {
                    const double w = twA.w[h + (k >> (s + 1))];
                    a[k] = u + v;
                    a[k + tk] = fz_mulmod(u - v, w, m);
                }
            }
        }

        if (valid) {
            int32_t *dst = out + poly * D + r;
#pragma unroll
            for (int k = 0; k < 16; ++k) dst[k * L] = (int)fz_cent(a[k], m);
        }
        __syncthreads();
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

template <int LOGD>
int launch16(fz_ctx *ctx, const int32_t *in, int32_t *out, size_t batch, bool inverse) {
    using G = Geom<LOGD>;
    const size_t tasks = (batch + G::PPW - 1) / G::PPW;
    const int cap = inverse ? ctx->grid_inv : ctx->grid_fwd;
    const unsigned grid = (unsigned)(tasks < (size_t)cap ? tasks : (size_t)cap);
    if (!inverse)
        hipLaunchKernelGGL(ntt_fwd16<LOGD>, dim3(grid), dim3(64), 0, ctx->stream, in, out, batch,
                           ctx->d_twB, ctx->twA, ctx->mod);
    else
        hipLaunchKernelGGL(ntt_inv16<LOGD>, dim3(grid), dim3(64), 0, ctx->stream, in, out, batch,
                           ctx->d_itwB, ctx->itwA, ctx->mod);
    return fz_check_hip(hipGetLastError(), "ntt16 launch");
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

template <int LOGD>
int query16(fz_ctx *ctx) {
    int nf = 0, ni = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nf, ntt_fwd16<LOGD>, 64, 0);
    if (e != hipSuccess) return fz_check_hip(e, "occupancy query (fwd)");
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&ni, ntt_inv16<LOGD>, 64, 0);
    if (e != hipSuccess) return fz_check_hip(e, "occupancy query (inv)");
    if (nf < 1) nf = 1;
    if (ni < 1) ni = 1;
    ctx->grid_fwd = nf * ctx->num_cu;
    ctx->grid_inv = ni * ctx->num_cu;
    return FZ_OK;
}

}  // namespace

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
        default: return fz_set_error(FZ_E_UNSUPPORTED, "degree %d not supported (2..256)", ctx->degree);
    }
}
