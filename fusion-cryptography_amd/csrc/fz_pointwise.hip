// fz_pointwise.hip -- streaming kernels: pointwise ring ops, the (1 x l).(l x 1) product and
// the fused arithmetic of sign / aggregate / verify.  All are HBM-bound (one modular
// multiply-add per 4..12 bytes moved): 16-byte-per-lane coalesced accesses, grid-stride
// loops over a grid capped at a few blocks per CU, fp64 exact arithmetic from fz_arith.h.
//
// Reference formulas: algebra/polynomials.py:303-385 (pointwise), algebra/matrices.py:108-131
// (scalar and matrix products), fusion/fusion.py:557 (sign), :670-676 (aggregate),
// :706-727 (verify).  Every partial result the reference centres is congruent mod q to the
// lazily accumulated value here, and the final value is centred once -- identical output.
#include "fz_internal.h"
#include "../../include/fusion_hip.h"

namespace {

typedef int fz_v4i __attribute__((ext_vector_type(4)));      // 16-byte vector the non-temporal builtins accept


constexpr int kBlock = 256;

__device__ __forceinline__ int cent_i32(double x, const FzMod m) { return (int)fz_cent(x, m); }

template <int OP>
__device__ __forceinline__ int pw_op(int a, int b, int acc, const FzMod m) {
    if (OP == FZ_OP_MUL) return (int)fz_mulmod_cent((double)a, (double)b, m);
    if (OP == FZ_OP_ADD) return cent_i32((double)a + (double)b, m);
    if (OP == FZ_OP_SUB) return cent_i32((double)a - (double)b, m);
    if (OP == FZ_OP_NEG) {   // -(x mod q), x mod q in [0,q): reference __neg__ (polynomials.py:325-333)
        double c = fz_cent((double)a, m);
        double y = c < 0.0 ? c + m.q : c;
        return (int)(-y);
    }
    // MULACC
    return cent_i32((double)acc + fz_mulmod((double)a, (double)b, m), m);
}

// streaming (non-temporal) 16-byte load: operands a kernel reads exactly once
typedef int fz_pw_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int4 ld_stream4(const int4 *p) {
    const fz_pw_v4i t = __builtin_nontemporal_load(reinterpret_cast<const fz_pw_v4i *>(p));
    return make_int4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void st_stream4(int4 *p, const int4 &v) {
    const fz_pw_v4i t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<fz_pw_v4i *>(p));
}

template <int OP>
__global__ __launch_bounds__(kBlock) void pw_kernel(const int32_t *a, const int32_t *b, int32_t *out,
                                                    size_t count, int vec, FzMod m) {
    const size_t n4 = vec ? count / 4 : 0;   // vec == 0: pointers not 16-byte aligned, all scalar
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int4 *a4 = reinterpret_cast<const int4 *>(a);
    const int4 *b4 = reinterpret_cast<const int4 *>(b);
    int4 *o4 = reinterpret_cast<int4 *>(out);
    for (size_t i = gid; i < n4; i += stride) {
        int4 x = a4[i];
        int4 y = (OP == FZ_OP_NEG) ? x : b4[i];
        int4 z = (OP == FZ_OP_MULACC) ? o4[i] : x;
        int4 o;
        o.x = pw_op<OP>(x.x, y.x, z.x, m);
        o.y = pw_op<OP>(x.y, y.y, z.y, m);
        o.z = pw_op<OP>(x.z, y.z, z.z, m);
        o.w = pw_op<OP>(x.w, y.w, z.w, m);
        o4[i] = o;        // a normal store: the consumer usually follows at once (streaming stores: +2.5 % cold, measured in round 2)
    }
    // ragged tail (count not a multiple of 4) or the whole range when unaligned
    for (size_t i = n4 * 4 + gid; i < count; i += stride) {
        int x = a[i];
        int y = (OP == FZ_OP_NEG) ? x : b[i];
        int z = (OP == FZ_OP_MULACC) ? out[i] : x;
        out[i] = pw_op<OP>(x, y, z, m);
    }
}

// out[row][j] = cent(a[row][j] * s[j]); d4 = degree/4 (degree >= 4) or scalar path
__global__ __launch_bounds__(kBlock) void pw_bcast_kernel(const int32_t *a, const int32_t *s, int32_t *out,
                                                          size_t total, int degree, FzMod m) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int j = (int)(i % (size_t)degree);
        out[i] = (int)fz_mulmod_cent((double)a[i], (double)s[j], m);
    }
}

// out[b][j] = cent(sum_k A[k][j] * S[b][k][j]); one thread per (b, 4 coefficients).  The rows of a product are a
// sequential chain for its thread: eight rows (16 loads of 16 bytes) are requested together before any arithmetic
// (one row at a time the kernel ran at 39 % of HBM peak on cold operands: latency, not bandwidth).
__global__ __launch_bounds__(kBlock) void matvec_kernel(const int32_t *A, const int32_t *S, int32_t *out,
                                                        size_t batch, int l, int degree, FzMod m) {
    constexpr int U = 8;
    const int d4 = degree / 4;
    const size_t total = batch * (size_t)d4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const size_t b = i / d4;
        const int j4 = (int)(i % d4);
        const int4 *Ap = reinterpret_cast<const int4 *>(A) + j4;
        const int4 *Sp = reinterpret_cast<const int4 *>(S + b * (size_t)l * degree) + j4;
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        int k = 0;
        for (; k + U <= l; k += U) {
            int4 x[U], y[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                x[u] = Ap[(size_t)(k + u) * d4];
                y[u] = Sp[(size_t)(k + u) * d4];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                s0 += fz_mulmod((double)x[u].x, (double)y[u].x, m);
                s1 += fz_mulmod((double)x[u].y, (double)y[u].y, m);
                s2 += fz_mulmod((double)x[u].z, (double)y[u].z, m);
                s3 += fz_mulmod((double)x[u].w, (double)y[u].w, m);
            }
        }
        for (; k < l; ++k) {
            const int4 x = Ap[(size_t)k * d4], y = Sp[(size_t)k * d4];
            s0 += fz_mulmod((double)x.x, (double)y.x, m);
            s1 += fz_mulmod((double)x.y, (double)y.y, m);
            s2 += fz_mulmod((double)x.z, (double)y.z, m);
            s3 += fz_mulmod((double)x.w, (double)y.w, m);
        }
        int4 o;
        o.x = cent_i32(s0, m); o.y = cent_i32(s1, m); o.z = cent_i32(s2, m); o.w = cent_i32(s3, m);
        reinterpret_cast<int4 *>(out + b * (size_t)degree)[j4] = o;
    }
}
// The same product as (CX, KS) workgroups, CX = max(64, 256 / KS) columns (product, 4 coefficients) by KS slices of the k
// range: the waves with threadIdx.y = y take slice y -- KS times the rows in flight when one wave per 64 columns would leave
// the chip short of waves (2048 products of degree 64 are 512 of them) -- and accumulate in INTEGERS: A
// split into 16-bit halves on the fly, hi += s * (A >> 16), lo += s * (A & 0xffff) (v_mad_i64_i32), exact for any int32
// operands while l <= 2^15 (fz_arith.h).  Every iteration requests eight rows whether the slice still has eight or not (the
// row index is clamped, the surplus is not accumulated: the slice bounds are wave-uniform) -- a tail of single-row iterations
// is a chain of dependent round trips (five of them at four slices of 83 rows: 37.9 us against 33.6 with two slices).  The
// slices' sums meet in LDS; slice 0 reduces them to the centred row.
template <int KS>
__global__ __launch_bounds__(KS <= 4 ? 256 : 64 * KS) void matvec_sliced_kernel(const int32_t *A, const int32_t *S, int32_t *out,
                                                                                size_t batch, int l, int degree, FzMod m) {
    constexpr int U = 8, CX = KS <= 4 ? 256 / KS : 64;
    __shared__ long long red[(KS > 1 ? KS - 1 : 1) * 8 * CX];
    const int d4 = degree / 4;
    const size_t total = batch * (size_t)d4;
    const size_t i_raw = (size_t)blockIdx.x * CX + threadIdx.x;
    const bool live = i_raw < total;
    const size_t i = live ? i_raw : total - 1;          // idle lanes of the last workgroup shadow the last column and store nothing
    const int sl = threadIdx.y;
    const size_t b = i / d4;
    const int j4 = (int)(i % d4);
    const int per = (l + KS - 1) / KS;
    const int k1 = (sl + 1) * per < l ? (sl + 1) * per : l;
    const int4 *Ap = reinterpret_cast<const int4 *>(A) + j4;
    const int4 *Sp = reinterpret_cast<const int4 *>(S + b * (size_t)l * degree) + j4;
    long long h0 = 0, h1 = 0, h2 = 0, h3 = 0, l0 = 0, l1 = 0, l2 = 0, l3 = 0;
#define FZ_MV_ACC(XA, YS) \
    h0 += (long long)(YS).x * ((XA).x >> 16); l0 += (long long)(YS).x * ((XA).x & 0xffff); \
    h1 += (long long)(YS).y * ((XA).y >> 16); l1 += (long long)(YS).y * ((XA).y & 0xffff); \
    h2 += (long long)(YS).z * ((XA).z >> 16); l2 += (long long)(YS).z * ((XA).z & 0xffff); \
    h3 += (long long)(YS).w * ((XA).w >> 16); l3 += (long long)(YS).w * ((XA).w & 0xffff);
    for (int k = sl * per; k < k1; k += U) {
        int4 x[U], y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = k + u < k1 ? k + u : k1 - 1;
            x[u] = Ap[(size_t)kk * d4];
            y[u] = ld_stream4(Sp + (size_t)kk * d4);        // the vectors are read once (A is every product's: from the L2)
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (k + u < k1) { FZ_MV_ACC(x[u], y[u]) }
    }
#undef FZ_MV_ACC
    if (KS > 1) {
        if (sl > 0) {
            long long *mine = red + (size_t)(sl - 1) * 8 * CX + threadIdx.x;
            mine[0] = h0; mine[CX] = h1; mine[2 * CX] = h2; mine[3 * CX] = h3;
            mine[4 * CX] = l0; mine[5 * CX] = l1; mine[6 * CX] = l2; mine[7 * CX] = l3;
        }
        __syncthreads();
        if (sl > 0) return;
#pragma unroll
        for (int t = 0; t < KS - 1; ++t) {
            const long long *o = red + (size_t)t * 8 * CX + threadIdx.x;
            h0 += o[0]; h1 += o[CX]; h2 += o[2 * CX]; h3 += o[3 * CX];
            l0 += o[4 * CX]; l1 += o[5 * CX]; l2 += o[6 * CX]; l3 += o[7 * CX];
        }
    }
    if (!live) return;
    const bool small = l <= 32;
    int4 o;
    o.x = cent_i32(fz_imad_total(h0, l0, small, m), m); o.y = cent_i32(fz_imad_total(h1, l1, small, m), m);
    o.z = cent_i32(fz_imad_total(h2, l2, small, m), m); o.w = cent_i32(fz_imad_total(h3, l3, small, m), m);
    reinterpret_cast<int4 *>(out + b * (size_t)degree)[j4] = o;
}
// Few products (verify: one per aggregate): one 1024-thread block per product, the k-loop split over
// 1024/(degree/4) slices and reduced through LDS -- 83 dependent-latency iterations become 6.
__global__ __launch_bounds__(1024) void matvec_split_kernel(const int32_t *A, const int32_t *S, int32_t *out,
                                                            int l, int degree, FzMod m) {
    __shared__ double red[1024 * 4];
    const int d4 = degree / 4, slices = 1024 / d4;
    const int j4 = threadIdx.x % d4, sl = threadIdx.x / d4;
    const size_t b = blockIdx.x;
    const int4 *Ap = reinterpret_cast<const int4 *>(A) + j4;
    const int4 *Sp = reinterpret_cast<const int4 *>(S + b * (size_t)l * degree) + j4;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int k = sl; k < l; k += slices) {
        int4 x = Ap[(size_t)k * d4];
        int4 y = Sp[(size_t)k * d4];
        s0 += fz_mulmod((double)x.x, (double)y.x, m);
        s1 += fz_mulmod((double)x.y, (double)y.y, m);
        s2 += fz_mulmod((double)x.z, (double)y.z, m);
        s3 += fz_mulmod((double)x.w, (double)y.w, m);
    }
    double *mine = red + (size_t)threadIdx.x * 4;
    mine[0] = s0; mine[1] = s1; mine[2] = s2; mine[3] = s3;
    __syncthreads();
    if (sl == 0) {
        for (int t = 1; t < slices; ++t) {
            const double *o = red + (size_t)(t * d4 + j4) * 4;
            s0 += o[0]; s1 += o[1]; s2 += o[2]; s3 += o[3];
        }
        int4 r;
        r.x = cent_i32(s0, m); r.y = cent_i32(s1, m); r.z = cent_i32(s2, m); r.w = cent_i32(s3, m);
        reinterpret_cast<int4 *>(out + b * (size_t)degree)[j4] = r;
    }
}
// any degree (degree % 4 != 0, or rows that are not 16-byte aligned): scalar
__global__ __launch_bounds__(kBlock) void matvec_scalar_kernel(const int32_t *A, const int32_t *S, int32_t *out,
                                                               size_t batch, int l, int degree, FzMod m) {
    const size_t total = batch * (size_t)degree;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const size_t b = i / degree;
        const int j = (int)(i % degree);
        double s = 0;
        for (int k = 0; k < l; ++k)
            s += fz_mulmod((double)A[(size_t)k * degree + j], (double)S[(b * l + k) * (size_t)degree + j], m);
        out[i] = cent_i32(s, m);
    }
}

// sig[b][k][j] = cent(cent(L[b][k][j] * c[b][j]) + R[b][k][j]); sk_hat = [batch][2][l][degree]
__global__ __launch_bounds__(kBlock) void sign_kernel(const int32_t *sk_hat, const int32_t *c_hat, int32_t *sig,
                                                      size_t batch, int l, int degree, FzMod m) {
    const int d4 = degree / 4;
    const size_t per_sig = (size_t)l * d4;
    const size_t total = batch * per_sig;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const size_t b = i / per_sig;
        const size_t rem = i % per_sig;          // k*d4 + j4
        const int j4 = (int)(rem % d4);
        const int4 *Lp = reinterpret_cast<const int4 *>(sk_hat + b * 2 * (size_t)l * degree);
        const int4 *Rp = Lp + per_sig;
        const int4 x = ld_stream4(Lp + rem), y = ld_stream4(Rp + rem);      // the key halves are read once
        int4 c = reinterpret_cast<const int4 *>(c_hat + b * (size_t)degree)[j4];
        int4 o;
        o.x = cent_i32(fz_mulmod((double)x.x, (double)c.x, m) + (double)y.x, m);
        o.y = cent_i32(fz_mulmod((double)x.y, (double)c.y, m) + (double)y.y, m);
        o.z = cent_i32(fz_mulmod((double)x.z, (double)c.z, m) + (double)y.z, m);
        o.w = cent_i32(fz_mulmod((double)x.w, (double)c.w, m) + (double)y.w, m);
        // a normal store: the aggregation usually reads the signatures next, and finds most of them in the caches -- sign 44.0 us
        // + aggregate 8.8 us chained (aggregate alone, cold: 17.6); a streaming store makes sign 40.9 us and the pair 58.2
        // (profiles/r05_keygen_sign_pair.txt, tools/probes/keygen_sign_pair.py)
        reinterpret_cast<int4 *>(sig)[i] = o;
    }
}

// any degree (degree % 4 != 0 included): one thread per coefficient
__global__ __launch_bounds__(kBlock) void sign_scalar_kernel(const int32_t *sk_hat, const int32_t *c_hat, int32_t *sig,
                                                             size_t batch, int l, int degree, FzMod m) {
    const size_t per_sig = (size_t)l * degree, total = batch * per_sig;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const size_t b = i / per_sig, rem = i % per_sig;
        const int32_t *Lp = sk_hat + b * 2 * per_sig;
        const double cj = (double)c_hat[b * (size_t)degree + rem % degree];
        sig[i] = cent_i32(fz_mulmod((double)Lp[rem], cj, m) + (double)Lp[per_sig + rem], m);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// One-pass aggregation (fusion/fusion.py:670-676, and the verification target of :706-714 as extra columns).
//
//   out[g][k][j] = sum_i sig[g][i][k][j] * alpha[g][i][j]      (mod q; centred int32 or int64 partial sums)
//
// Work split: a workgroup owns 1024 consecutive coefficients of the aggregate (kAggR = 4 int4 columns per lane, i.e.
// four rows at degree 256, sixteen at degree 64) for ONE slice of the signers; its WAVES waves take the slice's
// signers round-robin and never talk to each other until the end.  Per signer a lane loads ONE int4 of alpha (its
// position j is the same for its four columns) and four int4 of sigma, the next signer's five loads in flight
// meanwhile.  Arithmetic: alpha = hi * 2^16 + lo (hi in [-2^15, 2^15), lo in [0, 2^16)), so x * hi and x * lo are
// below 2^47 for any int32 x and sixteen signers accumulate EXACTLY in fp64 (< 2^51) with two FMAs per coefficient
// and no reduction at all; every kAggFold signers the two sums fold back below q (fz_fold, exact).  The split of
// alpha costs 16 integer/convert operations per signer and lane, shared by the lane's 16 coefficients.
// End of the workgroup: the WAVES partials meet in LDS; with one slice per aggregate the result is written directly;
// with several, each workgroup adds its 1024 exact integers into the tile's 64-bit accumulator words with returning
// agent-scope atomics (performed at the memory side: coherent across the 8 XCD L2s without any fence); a word also
// counts its arrivals, so the thread whose add comes last holds the finished sum and writes the output (see below).
// Only sigma and alpha are ever read: no scratch round trip, no second launch.
// Workgroup -> XCD placement (speed only): workgroups b and b + 8 share an XCD, so all column blocks of one
// (aggregate, slice) pair get block ids equal mod 8 and each alpha row is fetched into ONE L2.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kAggR = 4;          // int4 columns per lane (the default; 3 and 2 are instantiated too: the launcher picks what fills the chip)
constexpr int kAggDepth = 3;      // signers in flight per wave
constexpr int kAggFold = 16;      // signers between folds: 16 * 2^31 * 2^16 = 2^51 < 2^53
constexpr int kAggTile = 64 * kAggR * 4;   // coefficients per workgroup (1024)

struct AggDiv {                   // host-side arithmetic of the tile mapping (see the kernel)
    unsigned m_ncb, m_nsl;        // ceil(2^32 / ncb), ceil(2^32 / nsl): __umulhi(x, m) == x / d exactly for x * d < 2^32
    unsigned per_xcd;             // ceil(tiles / 8)
    unsigned base, extra;         // N / nsl, N % nsl
};

// RAG = FzNoRag: `groups` aggregates of N signers each, aggregate g's rows at g * N; RAG = FzRagged: aggregates of DIFFERENT
// sizes in one launch (fz_aggregate_*_ragged: many independent aggregate() calls, fusion.py:655, batched) -- aggregate g's
// signers are rows [off[g], off[g+1]) of the concatenated arrays, its slices from the table's base / extra
// SIGN (round 4): the signatures do not exist yet -- sigma_i = L_i * c_i + R_i (fusion.py:557) is computed from the secret key
// rows (sk_hat [N][2][l][degree]) and the challenge c [N][degree] as the signer comes up, written to sig_out, and aggregated
// from registers: what sign_core + this kernel read and write in two launches ((3l + 1) + (l + 1) rows per signature) becomes
// 3l + 2 rows in one -- the l rows of sigma are never read back.  Two signers in flight per wave instead of three (the key
// halves are two loads per column instead of one).
struct FzNoRag {};
template <int WAVES, typename RAG, bool SIGN = false, int AR = kAggR>
__global__ __launch_bounds__(64 * WAVES) void aggregate_onepass(const int32_t *sig, const int32_t *alpha, const int32_t *vkL,
                                                                const int32_t *vkR, const int32_t *c, size_t N, int l, int d4,
                                                                int ncb_a, int ncb, int nsl, int pairs, AggDiv dv,
                                                                unsigned long long *accum, int64_t *out64, size_t pstride,
                                                                int64_t *tout64, size_t tstride, int32_t *out32, FzMod m, RAG rag,
                                                                const int32_t *sk_hat = nullptr, int32_t *sig_out = nullptr) {
    constexpr bool RAGGED = !__is_same(RAG, FzNoRag);
    constexpr int DEPTH = SIGN ? 2 : kAggDepth;
    constexpr int TILE = 64 * AR * 4;                 // coefficients per workgroup: AR int4 columns per lane (1024 at AR = 4)
    __shared__ __attribute__((aligned(16))) double red[WAVES * TILE];      // 64 KiB at 8 waves
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;      // the wave index is uniform: say so (scalar address arithmetic)
    // block id -> tile t = p * ncb + cb (p = g * nsl + sb: the (aggregate, slice) pair, cb the column block).  Workgroups
    // b and b + 8 share an XCD, so XCD x = b % 8 gets the CONTIGUOUS run of tiles [x * per_xcd, (x + 1) * per_xcd): equal
    // load on every XCD (a pair-per-XCD mapping left XCDs with 21 and others with 42 workgroups at 12 slices: 30.7 us
    // instead of 21.9), and an alpha row is still fetched into one L2, two where a run of tiles ends inside a pair
    // (no integer division on the device: a uniform 64-bit divide is ~100 instructions in front of the first load;
    // quotients by ncb and nsl come from host-computed reciprocals, exact for these ranges -- AggDiv)
    const unsigned tiles = (unsigned)pairs * (unsigned)ncb, per_xcd = dv.per_xcd;
    const unsigned tl = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= per_xcd || tl >= tiles) return;
    const unsigned pu = ncb > 1 ? __umulhi(tl, dv.m_ncb) : tl;
    const int cb = (int)(tl - pu * (unsigned)ncb);
    const unsigned gu = nsl > 1 ? __umulhi(pu, dv.m_nsl) : pu;
    const size_t g = (size_t)gu;
    const int sb = (int)(pu - gu * (unsigned)nsl);
    size_t base = dv.base, extra = dv.extra, first_row = g * N;
    if constexpr (RAGGED) {
        base = rag.base[gu];
        extra = rag.extra[gu];
        first_row = rag.off[gu];
    }
    const size_t i0 = (size_t)sb * base + ((size_t)sb < extra ? (size_t)sb : extra);
    const size_t i1 = i0 + base + ((size_t)sb < extra ? 1 : 0);
    const bool tgt = cb >= ncb_a;
    const size_t cols_a = (size_t)l * d4;
    const int4 *alpha4 = reinterpret_cast<const int4 *>(alpha) + first_row * (size_t)d4;

    double lo[AR][4];
#pragma unroll
    for (int r = 0; r < AR; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) lo[r][k] = 0.0;

    if (!tgt) {
        double hi[AR][4];
#pragma unroll
        for (int r = 0; r < AR; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k) hi[r][k] = 0.0;
        const int j4 = lane & (d4 - 1);                   // d4 (a power of two) divides 64: the same position for the lane's four columns
        size_t col[AR];
#pragma unroll
        for (int r = 0; r < AR; ++r) {
            const size_t cr = (size_t)cb * (64 * AR) + (size_t)r * 64 + lane;
            col[r] = cr < cols_a ? cr : cols_a - 1;       // clamped lanes compute garbage that is never written
        }
        const int4 *sig4 = reinterpret_cast<const int4 *>(sig) + first_row * cols_a;
        const int4 *sk4 = reinterpret_cast<const int4 *>(sk_hat) + first_row * 2 * cols_a;        // SIGN: [signer][L | R][l * d4]
        const int4 *c4 = reinterpret_cast<const int4 *>(c) + first_row * (size_t)d4;
        int4 *so4 = reinterpret_cast<int4 *>(sig_out) + first_row * cols_a;
        bool live[AR];                                 // SIGN: this lane's column exists (a clamped lane must not store)
#pragma unroll
        for (int r = 0; r < AR; ++r) live[r] = (size_t)cb * (64 * AR) + (size_t)r * 64 + lane < cols_a;
        // DEPTH signers in flight per wave (5 loads of 16 bytes per lane each; 10 with SIGN): a wave's signers are a SEQUENTIAL
        // chain of memory latencies otherwise (first version, one signer ahead: N = 1024 took 29 us, 11 signers per wave
        // at ~2 us each)
        int4 a_q[DEPTH], x_q[DEPTH][AR], c_q[SIGN ? DEPTH : 1], r_q[SIGN ? DEPTH : 1][AR];
        auto load = [&](int s, size_t i) {
            a_q[s] = alpha4[i * d4 + j4];
            if constexpr (SIGN) {
                c_q[s] = c4[i * d4 + j4];
#pragma unroll
                for (int r = 0; r < AR; ++r) {
                    x_q[s][r] = ld_stream4(sk4 + i * 2 * cols_a + col[r]);                  // L
                    r_q[s][r] = ld_stream4(sk4 + (i * 2 + 1) * cols_a + col[r]);            // R
                }
            } else {
#pragma unroll
                for (int r = 0; r < AR; ++r) x_q[s][r] = ld_stream4(sig4 + i * cols_a + col[r]);      // read once (alpha: by every column block)
            }
        };
        const size_t first = i0 + wave;
        int since = 0;
        // one signer from slot s: alpha split, (SIGN: the signature made and stored,) the AR x 4 products accumulated
        auto consume = [&](int s, size_t is) __attribute__((always_inline)) {
            const int av[4] = {a_q[s].x, a_q[s].y, a_q[s].z, a_q[s].w};
            double ah[4], al[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ah[k] = (double)(av[k] >> 16);
                al[k] = (double)(av[k] & 0xffff);
            }
#pragma unroll
            for (int r = 0; r < AR; ++r) {
                double xv[4] = {(double)x_q[s][r].x, (double)x_q[s][r].y, (double)x_q[s][r].z, (double)x_q[s][r].w};
                if constexpr (SIGN) {                 // sigma = cent(cent(L * c) + R), exactly sign_kernel's arithmetic
                    const double cv[4] = {(double)c_q[s].x, (double)c_q[s].y, (double)c_q[s].z, (double)c_q[s].w};
                    const double rv[4] = {(double)r_q[s][r].x, (double)r_q[s][r].y, (double)r_q[s][r].z, (double)r_q[s][r].w};
                    int sv[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        sv[k] = cent_i32(fz_mulmod(xv[k], cv[k], m) + rv[k], m);
                        xv[k] = (double)sv[k];
                    }
                    if (live[r]) st_stream4(so4 + is * cols_a + col[r], make_int4(sv[0], sv[1], sv[2], sv[3]));     // never read back here
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    hi[r][k] = __builtin_fma(xv[k], ah[k], hi[r][k]);
                    lo[r][k] = __builtin_fma(xv[k], al[k], lo[r][k]);
                }
            }
            if (++since == kAggFold) {
                since = 0;
#pragma unroll
                for (int r = 0; r < AR; ++r)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        lo[r][k] = fz_fold(lo[r][k], m) + fz_fold(hi[r][k] * 65536.0, m);
                        hi[r][k] = 0.0;
                    }
            }
        };
        // The wave's signers first, first + WAVES, ... in a rolling window of DEPTH slots.  The loop that carries the bulk has NO
        // condition around its loads: gfx9 counts outstanding loads in ONE in-order counter, and the compiler, which must assume
        // the fewest loads any path may have issued, answered a run-time `if (next < i1) load(...)` with a wait for ALL loads at
        // the loop header -- the window drained once per DEPTH signers (one whole memory latency each time; rounds 3-4).
        // Steady loop: every slot full, every refill valid.  Then ONE pass whose refills may run out, then the drain.
        size_t i = first;
        if (first + (size_t)(DEPTH - 1) * WAVES < i1) {                          // wave-uniform: at least DEPTH signers
#pragma unroll
            for (int s = 0; s < DEPTH; ++s) load(s, first + (size_t)s * WAVES);
            for (; i + (size_t)(2 * DEPTH - 1) * WAVES < i1; i += (size_t)DEPTH * WAVES) {
#pragma unroll
                for (int s = 0; s < DEPTH; ++s) {
                    const size_t is = i + (size_t)s * WAVES;
                    consume(s, is);
                    load(s, is + (size_t)DEPTH * WAVES);
                }
            }
#pragma unroll
            for (int s = 0; s < DEPTH; ++s) {
                const size_t is = i + (size_t)s * WAVES, in = is + (size_t)DEPTH * WAVES;
                consume(s, is);
                if (in < i1) load(s, in);
            }
            i += (size_t)DEPTH * WAVES;
        } else {
#pragma unroll
            for (int s = 0; s < DEPTH; ++s)
                if (first + (size_t)s * WAVES < i1) load(s, first + (size_t)s * WAVES);
        }
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) {                                        // the drain: fewer than DEPTH signers are left
            const size_t is = i + (size_t)s * WAVES;
            if (is < i1) consume(s, is);
        }
#pragma unroll
        for (int r = 0; r < AR; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k) lo[r][k] = fz_fold(lo[r][k], m) + fz_fold(hi[r][k] * 65536.0, m);
    } else {
        // verification target: sum_i (vkL_i * c_i + vkR_i) * alpha_i, degree/4 int4 columns in lo[0]
        const size_t tcol = (size_t)(cb - ncb_a) * 64 + lane;
        const size_t tc = tcol < (size_t)d4 ? tcol : (size_t)d4 - 1;
        const int4 *L4 = reinterpret_cast<const int4 *>(vkL) + first_row * (size_t)d4;
        const int4 *R4 = reinterpret_cast<const int4 *>(vkR) + first_row * (size_t)d4;
        const int4 *C4 = reinterpret_cast<const int4 *>(c) + first_row * (size_t)d4;
#pragma unroll 2
        for (size_t i = i0 + wave; i < i1; i += WAVES) {
            const size_t o = i * d4 + tc;
            const int4 L = L4[o], R = R4[o], ch = C4[o], a = alpha4[o];
            lo[0][0] += fz_mulmod(fz_mulmod((double)L.x, (double)ch.x, m) + (double)R.x, (double)a.x, m);   // |inner| < 2^32
            lo[0][1] += fz_mulmod(fz_mulmod((double)L.y, (double)ch.y, m) + (double)R.y, (double)a.y, m);
            lo[0][2] += fz_mulmod(fz_mulmod((double)L.z, (double)ch.z, m) + (double)R.z, (double)a.z, m);
            lo[0][3] += fz_mulmod(fz_mulmod((double)L.w, (double)ch.w, m) + (double)R.w, (double)a.w, m);
        }
    }

    // the WAVES partials meet in LDS: element e of the tile = coefficient cb * 1024 + e of the aggregate
    {
        double *mine = red + wave * TILE + lane * 4;
#pragma unroll
        for (int r = 0; r < AR; ++r) {
            *reinterpret_cast<double2 *>(mine + r * 256) = make_double2(lo[r][0], lo[r][1]);
            *reinterpret_cast<double2 *>(mine + r * 256 + 2) = make_double2(lo[r][2], lo[r][3]);
        }
    }
    __syncthreads();
    constexpr int PER = (TILE + 64 * WAVES - 1) / (64 * WAVES);          // elements per thread in the combine steps (AR = 3: the last half is idle)
    const size_t limit = tgt ? (size_t)d4 * 4 : cols_a * 4;
    const size_t k0 = tgt ? (size_t)(cb - ncb_a) * 256 : (size_t)cb * TILE;
    const int span = tgt ? 256 : TILE;
    double sum[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int e = threadIdx.x + u * 64 * WAVES;
        double s = 0.0;
        if (e < TILE) {
#pragma unroll
            for (int w = 0; w < WAVES; ++w) s += red[w * TILE + e];
        }
        sum[u] = s;
    }
    auto emit = [&](int e, long long v) {
        const size_t k = k0 + (size_t)e;
        if (tgt) tout64[g * tstride + k] = v;
        else if (out64) out64[g * pstride + k] = v;
        else out32[g * cols_a * 4 + k] = (int)fz_cent_i64(v, m);
    };
    if (nsl == 1) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int e = threadIdx.x + u * 64 * WAVES;
            if (e < span && k0 + (size_t)e < limit) emit(e, (long long)sum[u]);
        }
        return;
    }
    // Several slices per aggregate: every workgroup ADDS its exact integers into the tile's accumulator words with one
    // returning agent-scope atomic per coefficient (executed at the memory side: coherent across the 8 XCD L2s, no
    // fence).  A word carries the running sum in its low 54 bits (two's complement, |sum| < 2^53) and the number of
    // arrivals above them (each add also adds 2^54), so the value an add returns tells its thread whether it was the
    // LAST of the nsl adders of that coefficient: that thread then holds the complete sum, writes the output and
    // re-arms the word for the next launch by subtracting what it read.  No ticket, no barrier, no read-back pass:
    // the tail of the kernel is one atomic round trip (the first version -- fp64 adds, a ticket per tile, the last
    // workgroup swapping the sums out -- spent three: 8 us of fixed cost per launch, of which this removes ~2.5).
    unsigned long long *acc = accum + (g * (size_t)ncb + (size_t)cb) * TILE;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int e = threadIdx.x + u * 64 * WAVES;
        if (e < span && k0 + (size_t)e < limit) {
            const unsigned long long add = (unsigned long long)(long long)sum[u] + (1ull << 54);
            const unsigned long long now = __hip_atomic_fetch_add(acc + e, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + add;
            if ((unsigned)((now + (1ull << 53)) >> 54) == (unsigned)nsl) {
                emit(e, (long long)(now - ((unsigned long long)nsl << 54)));
                __hip_atomic_fetch_add(acc + e, 0ull - now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // back to zero
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Aggregation WITHOUT signer slices, for aggregates of a few hundred signers (round 4).  aggregate_onepass fills the chip
// by cutting the SIGNERS of an aggregate into slices, and pays for it at the end: every workgroup adds its tile into shared
// accumulator words (one returning atomic round trip to the memory side) and the last adder writes the output -- about
// 2 of the ~6 us a launch costs whatever N is (N = 256: 9.1 us = 30 % of the HBM roofline).  Here the chip is filled by
// cutting the COEFFICIENTS finer instead: a workgroup owns 16 int4 columns (256 contiguous bytes of a row) of R rows of the
// aggregate for ALL signers, so nothing is shared between workgroups, the result is written directly, and the only
// synchronisation is one workgroup barrier.  A wave's 64 lanes are 16 columns x 4 signers; its WAVES waves take the signers
// round-robin, DEPTH signers (R + 1 loads of 16 bytes each) in flight per lane; the 4 * WAVES partial sums of a coefficient meet
// first inside the wave (v_permlane32_swap / v_permlane16_swap: lanes 0..15 collect the other three quarters, no LDS) and
// then in 64 R doubles of LDS per wave.  R rows of a tile share ONE alpha load per signer (R = 4: 84 tiles per aggregate at
// rank 83, R = 2: 168; one row per tile and whole-row tiles were measured and dropped: every CU of an XCD then asks the L2 for
// the same alpha lines at the same time -- 256 x 4 signers 30 us against 20.7 -- profiles/r04_aggregate_direct_ab.txt).
// Same arithmetic as aggregate_onepass (alpha = hi * 2^16 + lo, exact fp64 sums folded every kAggFold signers), same outputs
// (int64 partial sums or centred int32; the verification target as extra tiles), uniform and ragged groups.
// ---------------------------------------------------------------------------------------------------------------

__device__ __forceinline__ double dir_from_upper(double x) {        // lanes 0..31: the value lanes 32..63 hold
    unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x), lo2 = lo, hi2 = hi;
    auto r = __builtin_amdgcn_permlane32_swap(lo, lo2, false, false);
    lo2 = r[1];
    r = __builtin_amdgcn_permlane32_swap(hi, hi2, false, false);
    hi2 = r[1];
    return __hiloint2double((int)hi2, (int)lo2);
}
__device__ __forceinline__ double dir_from_odd_row(double x) {      // lanes of even 16-lane rows: the value the next row holds
    unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x), lo2 = lo, hi2 = hi;
    auto r = __builtin_amdgcn_permlane16_swap(lo, lo2, false, false);
    lo2 = r[1];
    r = __builtin_amdgcn_permlane16_swap(hi, hi2, false, false);
    hi2 = r[1];
    return __hiloint2double((int)hi2, (int)lo2);
}

// CW: int4 columns per tile (16: a quarter row of degree 256, four signers side by side in a wave; 64: a whole row, one signer)
template <int CW, int R, int DEPTH, int WAVES, typename RAG>
__global__ __launch_bounds__(64 * WAVES) void aggregate_direct(const int32_t *sig, const int32_t *alpha, const int32_t *vkL,
                                                               const int32_t *vkR, const int32_t *c, size_t N, int l, int d4,
                                                               int ntile_a, int64_t *out64, size_t pstride, int64_t *tout64,
                                                               size_t tstride, int32_t *out32, FzMod m, RAG rag) {
    constexpr bool RAGGED = !__is_same(RAG, FzNoRag);
    constexpr int kDirCW = CW, kDirSub = 64 / CW;
    constexpr int STEP = kDirSub * WAVES;                          // signers the workgroup takes per round
    __shared__ __attribute__((aligned(16))) double red[WAVES * CW * 4 * R];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int colw = lane & (kDirCW - 1), sub = lane / kDirCW;
    const size_t g = blockIdx.y;
    const int t = blockIdx.x;
    size_t first_row = g * N, n = N;
    if constexpr (RAGGED) {
        first_row = rag.off[g];
        n = rag.off[g + 1] - rag.off[g];
    }
    const bool tgt = t >= ntile_a;
    const int ncg = d4 / kDirCW;                                   // column groups per row (4 at degree 256, 1 at degree 64)
    const int tt = tgt ? t - ntile_a : t;
    const int cg = tt % ncg, rg = tt / ncg;
    const int j4 = cg * kDirCW + colw;
    const size_t cols_a = (size_t)l * d4;
    const int4 *alpha4 = reinterpret_cast<const int4 *>(alpha) + first_row * (size_t)d4 + j4;
    double lo[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) lo[r][k] = 0.0;
    const size_t i_first = (size_t)wave * kDirSub + sub;

    if (!tgt) {
        double hi[R][4];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k) hi[r][k] = 0.0;
        size_t rowoff[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int k = rg * R + r;
            rowoff[r] = (size_t)(k < l ? k : l - 1) * d4 + j4;       // clamped rows compute garbage that is never written
        }
        const int4 *sig4 = reinterpret_cast<const int4 *>(sig) + first_row * cols_a;
        int4 a_q[DEPTH], x_q[DEPTH][R];
        // Rounds of STEP signers (a lane group takes signer `sub` of its wave's share of the round).  The round count is
        // wave-uniform; a lane whose signer of a round does not exist requests the aggregate's LAST signer instead and leaves it
        // out of the sums.  Round 5: requests are issued by the whole wave on every path -- with the lane-dependent
        // `if (i < n) load(...)` of round 4 the compiler could not count what is outstanding (one in-order counter for all
        // loads) and waited for EVERYTHING before every signer: the DEPTH-deep window was drained at each step, a whole memory
        // latency per round (N = 256: eight of them in a 9 us kernel).  Same scheme as aggregate_onepass: a steady loop whose
        // refills all exist, one pass whose refills run out, the drain.
        const size_t w0 = (size_t)wave * kDirSub;
        const int rounds = n > w0 ? (int)((n - w0 + STEP - 1) / STEP) : 0;
        auto load = [&](int s, int round) __attribute__((always_inline)) {
            const size_t i = i_first + (size_t)round * STEP, ic = i < n ? i : n - 1;
            a_q[s] = alpha4[ic * d4];
#pragma unroll
            for (int r = 0; r < R; ++r) x_q[s][r] = ld_stream4(sig4 + ic * cols_a + rowoff[r]);
        };
        auto consume = [&](int s, int round) __attribute__((always_inline)) {
            if (i_first + (size_t)round * STEP < n) {
                const int av[4] = {a_q[s].x, a_q[s].y, a_q[s].z, a_q[s].w};
                double ah[4], al[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    ah[k] = (double)(av[k] >> 16);
                    al[k] = (double)(av[k] & 0xffff);
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const double xv[4] = {(double)x_q[s][r].x, (double)x_q[s][r].y, (double)x_q[s][r].z, (double)x_q[s][r].w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        hi[r][k] = __builtin_fma(xv[k], ah[k], hi[r][k]);
                        lo[r][k] = __builtin_fma(xv[k], al[k], lo[r][k]);
                    }
                }
            }
        };
        int since = 0;
        auto fold_if_due = [&]() __attribute__((always_inline)) {
            since += DEPTH;
            if (since + DEPTH > kAggFold) {                        // lanes of a wave differ by at most one signer: fold together
                since = 0;
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        lo[r][k] = fz_fold(lo[r][k], m) + fz_fold(hi[r][k] * 65536.0, m);
                        hi[r][k] = 0.0;
                    }
            }
        };
        int k0 = 0;
        if (rounds >= DEPTH) {                                       // wave-uniform
#pragma unroll
            for (int s = 0; s < DEPTH; ++s) load(s, s);
            for (; k0 + 2 * DEPTH - 1 < rounds; k0 += DEPTH) {
#pragma unroll
                for (int s = 0; s < DEPTH; ++s) {
                    consume(s, k0 + s);
                    load(s, k0 + s + DEPTH);
                }
                fold_if_due();
            }
#pragma unroll
            for (int s = 0; s < DEPTH; ++s) {
                consume(s, k0 + s);
                if (k0 + s + DEPTH < rounds) load(s, k0 + s + DEPTH);
            }
            fold_if_due();
            k0 += DEPTH;
        } else {
#pragma unroll
            for (int s = 0; s < DEPTH; ++s)
                if (s < rounds) load(s, s);
        }
#pragma unroll
        for (int s = 0; s < DEPTH; ++s)                              // the drain: fewer than DEPTH rounds are left
            if (k0 + s < rounds) consume(s, k0 + s);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k) lo[r][k] = fz_fold(lo[r][k], m) + fz_fold(hi[r][k] * 65536.0, m);
    } else {
        // verification target sum_i (vkL_i * c_i + vkR_i) * alpha_i (fusion.py:706-714): 16 columns of the one target row
        const int4 *L4 = reinterpret_cast<const int4 *>(vkL) + first_row * (size_t)d4 + j4;
        const int4 *R4 = reinterpret_cast<const int4 *>(vkR) + first_row * (size_t)d4 + j4;
        const int4 *C4 = reinterpret_cast<const int4 *>(c) + first_row * (size_t)d4 + j4;
#pragma unroll 2
        for (size_t i = i_first; i < n; i += STEP) {
            const size_t o = i * d4;
            const int4 L = L4[o], Rv = R4[o], ch = C4[o], a = alpha4[o];
            lo[0][0] += fz_mulmod(fz_mulmod((double)L.x, (double)ch.x, m) + (double)Rv.x, (double)a.x, m);   // |inner| < 2^32
            lo[0][1] += fz_mulmod(fz_mulmod((double)L.y, (double)ch.y, m) + (double)Rv.y, (double)a.y, m);
            lo[0][2] += fz_mulmod(fz_mulmod((double)L.z, (double)ch.z, m) + (double)Rv.z, (double)a.z, m);
            lo[0][3] += fz_mulmod(fz_mulmod((double)L.w, (double)ch.w, m) + (double)Rv.w, (double)a.w, m);
        }
    }
    // the four signer quarters of the wave meet in lanes 0..15, then the waves in LDS: red[wave][r][column][k]
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double v = lo[r][k];
            if constexpr (CW == 16) {
                v += dir_from_upper(v);
                v += dir_from_odd_row(v);
            }
            lo[r][k] = v;
        }
    if (lane < kDirCW) {
        double *mine = red + (wave * R) * (CW * 4) + lane * 4;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            *reinterpret_cast<double2 *>(mine + r * (CW * 4)) = make_double2(lo[r][0], lo[r][1]);
            *reinterpret_cast<double2 *>(mine + r * (CW * 4) + 2) = make_double2(lo[r][2], lo[r][3]);
        }
    }
    __syncthreads();
    const int rows_here = tgt ? 1 : R;
    for (int e = threadIdx.x; e < rows_here * (CW * 4); e += 64 * WAVES) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) s += red[w * R * (CW * 4) + e];
        const int r = e / (CW * 4), k = tgt ? 0 : rg * R + r;
        if (k >= l) continue;
        const size_t coef = (size_t)k * d4 * 4 + (size_t)cg * (kDirCW * 4) + (e % (CW * 4));
        const long long v = (long long)s;
        if (tgt) tout64[g * tstride + coef] = v;
        else if (out64) out64[g * pstride + coef] = v;
        else out32[g * cols_a * 4 + coef] = (int)fz_cent_i64(v, m);
    }
}

// any degree, any N: one thread per coefficient of the aggregate (and of the target), all signers in sequence.
// Only for parameter sets the one-pass kernel does not cover (degree not a power of two <= 256).
__global__ __launch_bounds__(kBlock) void aggregate_generic_kernel(const int32_t *sig, const int32_t *alpha, const int32_t *vkL,
                                                                   const int32_t *vkR, const int32_t *c, size_t N, int l, int degree,
                                                                   int64_t *out64, size_t pstride, int64_t *tout64, size_t tstride,
                                                                   int32_t *out32, FzMod m) {
    const size_t g = blockIdx.z, per = (size_t)l * degree, total = per + (vkL ? (size_t)degree : 0);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const size_t j = e % (size_t)degree;
        double s = 0.0;
        if (e < per) {
            for (size_t i = 0; i < N; ++i)
                s += fz_mulmod((double)sig[(g * N + i) * per + e], (double)alpha[(g * N + i) * degree + j], m);
            if (out64) out64[g * pstride + e] = (int64_t)s;
            else out32[g * per + e] = (int)fz_cent_wide(s, m);
        } else {
            for (size_t i = 0; i < N; ++i) {
                const size_t o = (g * N + i) * degree + j;
                s += fz_mulmod(fz_mulmod((double)vkL[o], (double)c[o], m) + (double)vkR[o], (double)alpha[o], m);
            }
            tout64[g * tstride + j] = (int64_t)s;
        }
    }
}

__device__ __forceinline__ void atomic_add_i64(int64_t *p, double v) {
    atomicAdd(reinterpret_cast<unsigned long long *>(p), (unsigned long long)(long long)v);
}

// partial[g][j] += sum_i ((vkL_i*c_i + vkR_i) * alpha_i)[j]; one thread per coefficient, grid.y splits N
__global__ __launch_bounds__(kBlock) void target_kernel(const int32_t *vkL, const int32_t *vkR, const int32_t *c,
                                                        const int32_t *alpha, int64_t *partial, size_t pstride,
                                                        size_t N, int degree, FzMod m) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= degree) return;
    const size_t g = blockIdx.z, goff = g * N * (size_t)degree;
    const size_t per = (N + gridDim.y - 1) / gridDim.y;
    const size_t i0 = (size_t)blockIdx.y * per;
    const size_t i1 = (i0 + per < N) ? i0 + per : N;
    double s = 0;
#pragma unroll 4
    for (size_t i = i0; i < i1; ++i) {
        const size_t o = goff + i * (size_t)degree + j;
        double t = fz_mulmod((double)vkL[o], (double)c[o], m) + (double)vkR[o];   // |t| < 2^32
        s += fz_mulmod(t, (double)alpha[o], m);
    }
    if (i1 > i0) atomic_add_i64(partial + g * pstride + j, s);
}

// zero `count` int64 at partial + g*pstride for every group (one launch instead of G memsets)
__global__ __launch_bounds__(kBlock) void zero_i64_kernel(int64_t *partial, size_t pstride, size_t count) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    int64_t *p = partial + (size_t)blockIdx.z * pstride;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) p[i] = 0;
}

__global__ __launch_bounds__(kBlock) void reduce_i64_kernel(const int64_t *in, int32_t *out, size_t count, FzMod m) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride)
        out[i] = (int)fz_cent_i64(in[i], m);         // exact for ANY int64 (sums that crossed an all-reduce)
}

// one wave per row: max |x| over stored values, #{x : x mod q != 0}
__global__ __launch_bounds__(64) void norm_weight_kernel(const int32_t *coef, size_t batch, int degree, uint32_t q,
                                                         int64_t *max_abs, int32_t *weight) {
    for (size_t row = blockIdx.x; row < batch; row += gridDim.x) {
        long long mx = 0;
        int w = 0;
        for (int j = threadIdx.x; j < degree; j += 64) {
            long long x = coef[row * (size_t)degree + j];
            long long ax = x < 0 ? -x : x;
            mx = ax > mx ? ax : mx;
            // |x| <= 2^31 < 2q, so x mod q == 0 iff x in {0, q, -q}
            w += (ax != 0 && ax != (long long)q) ? 1 : 0;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            long long o = __shfl_xor(mx, off);
            mx = o > mx ? o : mx;
            w += __shfl_xor(w, off);
        }
        if (threadIdx.x == 0) {
            max_abs[row] = mx;
            weight[row] = w;
        }
    }
}

// verdict of fusion/fusion.py:718-728, evaluated in the reference's order; one block per aggregate
__global__ __launch_bounds__(256) void verdict_kernel(const int32_t *target, const int32_t *observed, int degree,
                                                      const int64_t *max_abs, const int32_t *weight, int l,
                                                      int64_t beta, int64_t omega, int *verdict) {
    __shared__ int s_mis, s_norm, s_wt;
    const size_t g = blockIdx.x;
    target += g * degree;
    observed += g * degree;
    max_abs += g * l;
    weight += g * l;
    if (threadIdx.x == 0) { s_mis = 0; s_norm = 0; s_wt = 0; }
    __syncthreads();
    for (int j = threadIdx.x; j < degree; j += blockDim.x)
        if (target[j] != observed[j]) atomicOr(&s_mis, 1);      // both centred: equal ints <=> equal mod q
    for (int k = threadIdx.x; k < l; k += blockDim.x) {
        if (max_abs[k] > beta) atomicOr(&s_norm, 1);
        if ((int64_t)weight[k] > omega) atomicOr(&s_wt, 1);
    }
    __syncthreads();
    if (threadIdx.x == 0)
        verdict[g] = s_mis ? FZ_VERDICT_TARGET_MISMATCH : (s_norm ? FZ_VERDICT_NORM : (s_wt ? FZ_VERDICT_WEIGHT : FZ_VERDICT_OK));
}

unsigned grid_for(fz_ctx *ctx, size_t work_items, int per_cu = -1) {
    size_t blocks = (work_items + kBlock - 1) / kBlock;
    // default: a flat grid, one item per thread (45.0 us per 1024 signatures for sign_core against 49.7 with the grid capped at
    // 8 workgroups per CU, cold: round 2)
    size_t cap = per_cu > 0 ? (size_t)ctx->num_cu * per_cu : (size_t)0x7fffffff;
    if (blocks < 1) blocks = 1;
    return (unsigned)(blocks < cap ? blocks : cap);
}

}  // namespace

int fz_launch_pw(fz_ctx *ctx, int op, const int32_t *a, const int32_t *b, int32_t *out, size_t count) {
    if (count == 0) return FZ_OK;
    const int vec = ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0) ? 1 : 0;
    const unsigned grid = grid_for(ctx, vec ? count / 4 + 4 : count);
    switch (op) {
        case FZ_OP_MUL: hipLaunchKernelGGL(pw_kernel<FZ_OP_MUL>, dim3(grid), dim3(kBlock), 0, ctx->stream, a, b, out, count, vec, ctx->mod); break;
        case FZ_OP_ADD: hipLaunchKernelGGL(pw_kernel<FZ_OP_ADD>, dim3(grid), dim3(kBlock), 0, ctx->stream, a, b, out, count, vec, ctx->mod); break;
        case FZ_OP_SUB: hipLaunchKernelGGL(pw_kernel<FZ_OP_SUB>, dim3(grid), dim3(kBlock), 0, ctx->stream, a, b, out, count, vec, ctx->mod); break;
        case FZ_OP_NEG: hipLaunchKernelGGL(pw_kernel<FZ_OP_NEG>, dim3(grid), dim3(kBlock), 0, ctx->stream, a, a, out, count, vec, ctx->mod); break;
        case FZ_OP_MULACC: hipLaunchKernelGGL(pw_kernel<FZ_OP_MULACC>, dim3(grid), dim3(kBlock), 0, ctx->stream, a, b, out, count, vec, ctx->mod); break;
        default: return fz_set_error(FZ_E_BADARG, "unknown pointwise op %d", op);
    }
    return fz_check_hip(hipGetLastError(), "pointwise launch");
}

int fz_launch_pw_bcast(fz_ctx *ctx, const int32_t *a, const int32_t *s, int32_t *out, size_t rows) {
    if (rows == 0) return FZ_OK;
    const size_t total = rows * (size_t)ctx->degree;
    hipLaunchKernelGGL(pw_bcast_kernel, dim3(grid_for(ctx, total)), dim3(kBlock), 0, ctx->stream, a, s, out, total,
                       ctx->degree, ctx->mod);
    return fz_check_hip(hipGetLastError(), "pw_bcast launch");
}

int fz_launch_matvec(fz_ctx *ctx, const int32_t *A, const int32_t *S, int32_t *out, size_t batch, int l) {
    if (batch == 0) return FZ_OK;
    const bool vec = ctx->degree % 4 == 0 && ((((uintptr_t)A | (uintptr_t)S | (uintptr_t)out) & 15) == 0);   // int4 rows
    if (vec && ctx->degree >= 16 && ctx->degree <= 4096 && (ctx->degree & (ctx->degree - 1)) == 0 &&
        (ctx->knob_matvec_slices < 0 ? batch * (size_t)(ctx->degree / 4) < (size_t)ctx->num_cu * 256
                                     : ctx->knob_matvec_slices == 0 && batch <= (size_t)ctx->num_cu * 2))
        // one 1024-thread workgroup per product while all of them are resident at once (two per CU); measured on cold operands
        // (profiles/r03_matvec_ab.txt): 512 products 11.1 us against 12.0 for the sliced kernel at degree 256, 8.7 against
        // 10.5 at degree 64; 1024 products 19.3 (sliced) against 26.2 / 12.5 against 14.0
        hipLaunchKernelGGL(matvec_split_kernel, dim3((unsigned)batch), dim3(1024), 0, ctx->stream, A, S, out, l,
                           ctx->degree, ctx->mod);
    else if (vec && l <= 32768 && ctx->knob_matvec_slices >= 0) {
        // slices of the k range per column: waves for every SIMD several times over, at least 8 rows per slice
        const size_t cols = batch * (size_t)(ctx->degree / 4), waves1 = (cols + 63) / 64;
        int ks = ctx->knob_matvec_slices;                                  // FZ_MATVEC_SLICES = 1 | 2 | 4 | 8 | 16 (A/B runs), -1: the fp64 kernels
        if (ks != 1 && ks != 2 && ks != 4 && ks != 8 && ks != 16) {
            ks = 1;
            while (ks < 16 && waves1 * ks < (size_t)ctx->num_cu * 8) ks <<= 1;
        }
        while (ks > 1 && l / ks < 8) ks >>= 1;
        const int cx = ks <= 4 ? 256 / ks : 64;
        const size_t blocks = (cols + cx - 1) / cx;
        if (blocks > 0x7fffffffull) return fz_set_error(FZ_E_UNSUPPORTED, "batch too large for one matvec launch");
        const dim3 grid((unsigned)blocks), block(cx, ks);
#define FZ_MVS(K) hipLaunchKernelGGL(matvec_sliced_kernel<K>, grid, block, 0, ctx->stream, A, S, out, batch, l, ctx->degree, ctx->mod)
        if (ks == 16) FZ_MVS(16);
        else if (ks == 8) FZ_MVS(8);
        else if (ks == 4) FZ_MVS(4);
        else if (ks == 2) FZ_MVS(2);
        else FZ_MVS(1);
#undef FZ_MVS
    }
    else if (vec)
        hipLaunchKernelGGL(matvec_kernel, dim3(grid_for(ctx, batch * (size_t)(ctx->degree / 4))), dim3(kBlock), 0,
                           ctx->stream, A, S, out, batch, l, ctx->degree, ctx->mod);
    else
        hipLaunchKernelGGL(matvec_scalar_kernel, dim3(grid_for(ctx, batch * (size_t)ctx->degree)), dim3(kBlock), 0,
                           ctx->stream, A, S, out, batch, l, ctx->degree, ctx->mod);
    return fz_check_hip(hipGetLastError(), "matvec launch");
}

int fz_launch_sign(fz_ctx *ctx, const int32_t *sk_hat, const int32_t *c_hat, int32_t *sig, size_t batch, int l) {
    if (batch == 0) return FZ_OK;
    if (ctx->degree % 4 != 0 || ((((uintptr_t)sk_hat | (uintptr_t)c_hat | (uintptr_t)sig) & 15) != 0)) {
        const size_t n = batch * (size_t)l * ctx->degree;
        hipLaunchKernelGGL(sign_scalar_kernel, dim3(grid_for(ctx, n)), dim3(kBlock), 0, ctx->stream, sk_hat, c_hat, sig, batch, l,
                           ctx->degree, ctx->mod);
        return fz_check_hip(hipGetLastError(), "sign (scalar) launch");
    }
    const size_t total = batch * (size_t)l * (ctx->degree / 4);
    hipLaunchKernelGGL(sign_kernel, dim3(grid_for(ctx, total)), dim3(kBlock), 0, ctx->stream, sk_hat, c_hat, sig, batch,
                       l, ctx->degree, ctx->mod);
    return fz_check_hip(hipGetLastError(), "sign launch");
}

static unsigned split_count(fz_ctx *ctx, size_t N, unsigned gx, size_t groups, size_t min_per_thread) {
    // enough blocks to fill the chip (~8 per CU over all groups) but at least `min_per_thread` items each
    size_t want = ((size_t)ctx->num_cu * 8 + gx * groups - 1) / (gx * groups);
    const size_t most = (N + min_per_thread - 1) / min_per_thread;
    if (want > most) want = most;
    if (want < 1) want = 1;
    if (want > 65535) want = 65535;
    return (unsigned)want;
}

template <int WAVES, int AR>
static void launch_onepass(fz_ctx *ctx, unsigned grid, const int32_t *sig, const int32_t *alpha, const int32_t *vkL,
                           const int32_t *vkR, const int32_t *c, size_t N, int l, int d4, int ncb_a, int ncb, int nsl, int pairs,
                           unsigned long long *acc, int64_t *out64, size_t pstride, int64_t *tout64, size_t tstride,
                           int32_t *out32, const FzRagged *rag, const int32_t *sk_hat = nullptr, int32_t *sig_out = nullptr) {
    AggDiv dv;
    dv.m_ncb = ncb > 1 ? (unsigned)((0x100000000ull + (unsigned)ncb - 1) / (unsigned)ncb) : 0u;      // unused for a divisor of 1
    dv.m_nsl = nsl > 1 ? (unsigned)((0x100000000ull + (unsigned)nsl - 1) / (unsigned)nsl) : 0u;
    dv.per_xcd = (unsigned)(((size_t)pairs * ncb + 7) / 8);
    dv.base = (unsigned)(N / (size_t)nsl);
    dv.extra = (unsigned)(N % (size_t)nsl);
    if (sk_hat)
        hipLaunchKernelGGL((aggregate_onepass<WAVES, FzNoRag, true, AR>), dim3(grid), dim3(64 * WAVES), 0, ctx->stream, sig, alpha, vkL, vkR, c, N,
                           l, d4, ncb_a, ncb, nsl, pairs, dv, acc, out64, pstride, tout64, tstride, out32, ctx->mod, FzNoRag(), sk_hat, sig_out);
    else if (rag)
        hipLaunchKernelGGL((aggregate_onepass<WAVES, FzRagged, false, AR>), dim3(grid), dim3(64 * WAVES), 0, ctx->stream, sig, alpha, vkL, vkR, c, N,
                           l, d4, ncb_a, ncb, nsl, pairs, dv, acc, out64, pstride, tout64, tstride, out32, ctx->mod, *rag, nullptr, nullptr);
    else
        hipLaunchKernelGGL((aggregate_onepass<WAVES, FzNoRag, false, AR>), dim3(grid), dim3(64 * WAVES), 0, ctx->stream, sig, alpha, vkL, vkR, c, N,
                           l, d4, ncb_a, ncb, nsl, pairs, dv, acc, out64, pstride, tout64, tstride, out32, ctx->mod, FzNoRag(), nullptr, nullptr);
}


template <int CW, int R, int DEPTH>
static void launch_direct(fz_ctx *ctx, dim3 grid, const int32_t *sig, const int32_t *alpha, const int32_t *vkL, const int32_t *vkR,
                          const int32_t *c, size_t N, int l, int d4, int ntile_a, int64_t *out64, size_t pstride, int64_t *tout64,
                          size_t tstride, int32_t *out32, const FzRagged *rag) {
    constexpr int WAVES = 8;
    if (rag)
        hipLaunchKernelGGL((aggregate_direct<CW, R, DEPTH, WAVES, FzRagged>), grid, dim3(64 * WAVES), 0, ctx->stream, sig, alpha, vkL, vkR, c, N, l,
                           d4, ntile_a, out64, pstride, tout64, tstride, out32, ctx->mod, *rag);
    else
        hipLaunchKernelGGL((aggregate_direct<CW, R, DEPTH, WAVES, FzNoRag>), grid, dim3(64 * WAVES), 0, ctx->stream, sig, alpha, vkL, vkR, c, N, l,
                           d4, ntile_a, out64, pstride, tout64, tstride, out32, ctx->mod, FzNoRag());
}

// out64 != nullptr: int64 partial sums at out64 + g*pstride; else centred int32 at out32 + g*l*degree.
// vkL != nullptr: the verification target's int64 partial sums go to tout64 + g*tstride in the same launch.
// h_offsets != nullptr (ragged): aggregate g's signers are rows [h_offsets[g], h_offsets[g+1]) of the arrays, groups <=
// kFzRaggedMax, N = the largest group; sig == nullptr then means "verification targets only".  One-pass kernel only.
// sk_hat != nullptr (fused signing, one-pass kernel only): sig is the OUTPUT sig_out, c the signers' challenges (required).
int fz_launch_aggregate(fz_ctx *ctx, const int32_t *sig, const int32_t *alpha, int64_t *out64, size_t pstride,
                        int32_t *out32, size_t groups, size_t N, int l, const int32_t *vkL, const int32_t *vkR,
                        const int32_t *c, int64_t *tout64, size_t tstride, const size_t *h_offsets, const int32_t *sk_hat,
                        int32_t *sig_out) {
    if (groups == 0) return FZ_OK;
    const int d = ctx->degree;
    const uintptr_t align = (uintptr_t)sig | (uintptr_t)alpha | (uintptr_t)vkL | (uintptr_t)vkR | (uintptr_t)c | (uintptr_t)sk_hat | (uintptr_t)sig_out;
    const bool vec = (d % 4 == 0) && (align & 15) == 0;
    if (sk_hat && !(vec && (d & (d - 1)) == 0 && d <= 256 && !h_offsets && c && sig_out))
        return fz_set_error(FZ_E_UNSUPPORTED, "fused signing + aggregation: power-of-two degree <= 256, 16-byte aligned rows, equal-size aggregates");
    if (sk_hat) sig = sig_out;                        // "signatures present" for the tile arithmetic below
    if (h_offsets && !(vec && (d & (d - 1)) == 0 && d <= 256 && groups <= (size_t)kFzRaggedMax))
        return fz_set_error(FZ_E_UNSUPPORTED, "ragged aggregation: power-of-two degree <= 256, 16-byte aligned rows, <= %d aggregates per launch", kFzRaggedMax);
    if (vec && (d & (d - 1)) == 0 && d <= 256) {
        const int d4 = d / 4;
        const size_t cols_a = (size_t)l * d4;
        // Few signers in the whole launch: no signer slices, no shared accumulators (aggregate_direct).  Measured on cold
        // operands, both forms forced (profiles/r05_aggregate_direct_ab.txt; round 5, after the sliced kernel's streaming loads and
        // this kernel's exact wait counts): one aggregate of 8 / 32 / 96 signers 3.4 / 3.9 / 5.0 us against 4.6 / 4.8 / 5.2 us for
        // the sliced kernel, 128: 5.6 either way, 192 / 256: 6.7 / 8.0 against 6.3 / 7.8; two of 64: 5.6 against 5.8, two of
        // 128: 8.1 against 7.3 -- so up to 128 signers per launch (round 4: 256); with the verification target in the same
        // launch the sliced kernel leads throughout.  FZ_AGG_DIRECT = -1: never; 2 | 4: always, with that many rows per tile
        // (the tests force both forms through every output mode).
        const bool direct_auto = !vkL && groups * N <= 128;
        if (!sk_hat && d4 >= 16 && ctx->knob_agg_direct >= 0 && (ctx->knob_agg_direct > 0 || direct_auto) && groups <= 65535) {
            int R = ctx->knob_agg_direct;
            if (R != 2 && R != 4) R = groups <= 2 ? 2 : 4;          // rows of a tile share one alpha load; 168 / 84 tiles per aggregate at rank 83
            const int ncg = d4 / 16;
            const int ntile_a = sig ? ((l + R - 1) / R) * ncg : 0;
            const int ntile = ntile_a + (vkL ? ncg : 0);
            FzRagged rag;
            if (h_offsets) {
                for (size_t g = 0; g <= groups; ++g) rag.off[g] = (unsigned)h_offsets[g];
                for (size_t g = 0; g < groups; ++g) rag.base[g] = rag.extra[g] = 0;
            }
            const FzRagged *rp = h_offsets ? &rag : nullptr;
            const dim3 grid((unsigned)ntile, (unsigned)groups);
            if (R == 4) launch_direct<16, 4, 3>(ctx, grid, sig, alpha, vkL, vkR, c, N, l, d4, ntile_a, out64, pstride, tout64, tstride, out32, rp);
            else launch_direct<16, 2, 5>(ctx, grid, sig, alpha, vkL, vkR, c, N, l, d4, ntile_a, out64, pstride, tout64, tstride, out32, rp);
            return fz_check_hip(hipGetLastError(), "aggregate (direct) launch");
        }
        // slices of the signers per aggregate: as many tiles (column block x aggregate x slice) as the chip holds at once --
        // one 8-wave workgroup per CU (180-220 VGPRs: two waves per SIMD) -- so that every CU streams from the first moment
        // and nothing waits for a second round (a 257th workgroup costs a third more: 264 tiles 28.1 us against 20.8 for 176);
        // at least kAggDepth signers per wave.  Rows per column block (AR = 4 | 3 | 2 int4 columns per lane): whichever leaves the
        // fewest CUs idle -- 4 aggregates of 256 signers + targets are 88 tiles per slice at AR = 4 (2 slices: 176 of 256 CUs)
        // and 116 at AR = 3 (2 slices: 232).
        constexpr int waves = 8;                       // (4-wave workgroups, two per CU, measured no better at any size: profiles/r03_aggregate_shapes.txt)
        const size_t capacity = (size_t)ctx->num_cu;
        int ar = 0, ncb_a = 0, ncb = 0;
        size_t nsl = 1, best_tiles = 0;
        for (int cand = kAggR; cand >= 2; --cand) {
            const int na = sig ? (int)((cols_a + 64 * cand - 1) / (64 * cand)) : 0;
            const int nb = na + (vkL ? 1 : 0);                       // d4 <= 64: the target fits one column block
            const size_t blocks_min = (size_t)nb * groups;
            size_t ns = capacity / blocks_min;
            const size_t most = N / ((size_t)waves * kAggDepth);
            if (ns > most) ns = most;
            if (ns < 1) ns = 1;
            if (ns > 512) ns = 512;                    // the arrival count shares a 64-bit word with the sum (10 bits)
            if (ns > N && N > 0) ns = N;
            if (N == 0) ns = 1;
            const size_t tiles = blocks_min * ns;
            // the widest block stands unless a narrower one puts at least 1/16 more workgroups into the ONE round
            // (launches of more than one round keep the widest block: fewest workgroups)
            const bool take = ar == 0 || (best_tiles <= capacity && tiles <= capacity && tiles * 16 >= best_tiles * 17);
            if (take) { ar = cand; ncb_a = na; ncb = nb; nsl = ns; best_tiles = tiles; }
        }
        // More tiles than CUs even at the widest block (many small aggregates in one launch: the batch queue's ragged launches):
        // the launch runs in several rounds, and at AR = 4 (190 VGPRs: one 8-wave workgroup per CU) a CU's next workgroup cannot
        // start before the previous one has drained its LDS reduction and stores.  AR = 2 (114 VGPRs: TWO workgroups per CU)
        // overlaps one workgroup's tail with the other's loads: 64 aggregates of 64 signers + targets 80.4 -> 74.0 us (0.57 ->
        // 0.62 of the HBM peak by bytes moved), 128 x 32: 95.8 -> 80.6, 256 x 16: 121 -> 99.5, 16 x 64: 23.9 -> 20.0; launches
        // of one round lose with it (8 x 128: 20.3 -> 23.5 us) and keep the rule above (profiles/r05_queue_aggregates.txt).
        {
            const int na4 = sig ? (int)((cols_a + 64 * kAggR - 1) / (64 * kAggR)) : 0;
            if ((size_t)(na4 + (vkL ? 1 : 0)) * groups > capacity) {
                ar = 2;
                ncb_a = sig ? (int)((cols_a + 64 * 2 - 1) / (64 * 2)) : 0;
                ncb = ncb_a + (vkL ? 1 : 0);
                nsl = 1;
            }
        }
        const size_t pairs = groups * nsl;
        // the kernel divides tile numbers by ncb and nsl with 32-bit reciprocals: exact while tiles * divisor < 2^32
        if (pairs * (size_t)ncb > 0x3fffffffull || pairs * (size_t)ncb * (size_t)(ncb > (int)nsl ? ncb : (int)nsl) >= 0x100000000ull)
            return fz_set_error(FZ_E_UNSUPPORTED, "aggregate: grid too large (%zu aggregates x %zu slices x %d column blocks)", groups, nsl, ncb);
        unsigned long long *acc = nullptr;
        if (nsl > 1) {
            int rc = fz_agg_scratch(ctx, groups * (size_t)ncb, (size_t)kAggTile, &acc);       // (sized for the widest tile)
            if (rc != FZ_OK) return rc;
        }
        const unsigned grid = (unsigned)(8 * ((pairs * (size_t)ncb + 7) / 8));
        FzRagged rag;
        if (h_offsets) {
            for (size_t g = 0; g < groups; ++g) {
                const size_t ng = h_offsets[g + 1] - h_offsets[g];
                rag.off[g] = (unsigned)h_offsets[g];
                rag.base[g] = (unsigned)(ng / nsl);
                rag.extra[g] = (unsigned)(ng % nsl);
            }
            rag.off[groups] = (unsigned)h_offsets[groups];
        }
        const FzRagged *rp = h_offsets ? &rag : nullptr;
#define FZ_ONEPASS(AR_) launch_onepass<8, AR_>(ctx, grid, sig, alpha, vkL, vkR, c, N, l, d4, ncb_a, ncb, (int)nsl, (int)pairs, acc, out64, \
                                                pstride, tout64, tstride, out32, rp, sk_hat, sig_out)
        if (ar == 4) FZ_ONEPASS(4);
        else if (ar == 3) FZ_ONEPASS(3);
        else FZ_ONEPASS(2);
#undef FZ_ONEPASS
        const int rc = fz_check_hip(hipGetLastError(), "aggregate launch");
        if (rc != FZ_OK && nsl > 1) ctx->agg_dirty = 1;              // accumulators / tickets may no longer be zero
        return rc;
    }
    // degrees the one-pass kernels do not cover (not a power of two, > 256, or unaligned rows)
    const size_t total = (size_t)l * d + (vkL ? (size_t)d : 0);
    hipLaunchKernelGGL(aggregate_generic_kernel, dim3(grid_for(ctx, total), 1, (unsigned)groups), dim3(kBlock), 0, ctx->stream,
                       sig, alpha, vkL, vkR, c, N, l, d, out64, pstride, tout64, tstride, out32, ctx->mod);
    return fz_check_hip(hipGetLastError(), "aggregate (generic) launch");
}

// out[seg][k][:] = in[seg][:] for k < l (generic-degree path of fz_keygen_core_bcast)
__global__ __launch_bounds__(kBlock) void bcast_rows_kernel(const int32_t *in, int32_t *out, size_t segments, int l, int degree) {
    const size_t total = segments * (size_t)l * degree, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride)
        out[i] = in[(i / ((size_t)l * degree)) * degree + i % degree];
}

int fz_launch_bcast_rows(fz_ctx *ctx, const int32_t *in, int32_t *out, size_t segments, int l) {
    const size_t total = segments * (size_t)l * ctx->degree;
    if (total == 0) return FZ_OK;
    hipLaunchKernelGGL(bcast_rows_kernel, dim3(grid_for(ctx, total)), dim3(kBlock), 0, ctx->stream, in, out, segments, l, ctx->degree);
    return fz_check_hip(hipGetLastError(), "bcast_rows launch");
}

// synthetic input generator ("device-side sampling for synthetic batches"): value i of the stream is
// SplitMix64(seed + (i + 1) * golden) mod q, centred -- the same sequence as the host generator the tests and the
// bench use (oracle.splitmix_centered), so device-filled and host-filled batches are interchangeable
__global__ __launch_bounds__(kBlock) void fill_synthetic_kernel(int32_t *out, size_t count, unsigned long long seed, uint32_t q) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        unsigned long long z = seed + (unsigned long long)(i + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        out[i] = (int32_t)((long long)(z % q) - (long long)((q - 1) / 2));
    }
}

int fz_launch_fill_synthetic(fz_ctx *ctx, int32_t *out, size_t count, unsigned long long seed) {
    if (count == 0) return FZ_OK;
    hipLaunchKernelGGL(fill_synthetic_kernel, dim3(grid_for(ctx, count)), dim3(kBlock), 0, ctx->stream, out, count, seed, ctx->q);
    return fz_check_hip(hipGetLastError(), "fill_synthetic launch");
}

int fz_launch_target_partial(fz_ctx *ctx, const int32_t *vkL, const int32_t *vkR, const int32_t *c, const int32_t *alpha,
                             int64_t *partial, size_t pstride, size_t groups, size_t N) {
    if (groups == 0) return FZ_OK;
    hipLaunchKernelGGL(zero_i64_kernel, dim3(1, 1, (unsigned)groups), dim3(kBlock), 0, ctx->stream, partial, pstride,
                       (size_t)ctx->degree);
    int rc = fz_check_hip(hipGetLastError(), "zero target");
    if (rc != FZ_OK || N == 0) return rc;
    const unsigned gx = (unsigned)((ctx->degree + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(target_kernel, dim3(gx, split_count(ctx, N, gx, groups, 8), (unsigned)groups), dim3(kBlock), 0,
                       ctx->stream, vkL, vkR, c, alpha, partial, pstride, N, ctx->degree, ctx->mod);
    return fz_check_hip(hipGetLastError(), "target launch");
}

int fz_launch_reduce_i64(fz_ctx *ctx, const int64_t *in, int32_t *out, size_t count) {
    if (count == 0) return FZ_OK;
    hipLaunchKernelGGL(reduce_i64_kernel, dim3(grid_for(ctx, count)), dim3(kBlock), 0, ctx->stream, in, out, count, ctx->mod);
    return fz_check_hip(hipGetLastError(), "reduce_i64 launch");
}

int fz_launch_norm_weight(fz_ctx *ctx, const int32_t *coef, size_t batch, int64_t *max_abs, int32_t *weight) {
    if (batch == 0) return FZ_OK;
    size_t cap = (size_t)ctx->num_cu * 16;
    const unsigned grid = (unsigned)(batch < cap ? batch : cap);
    hipLaunchKernelGGL(norm_weight_kernel, dim3(grid), dim3(64), 0, ctx->stream, coef, batch, ctx->degree, ctx->q, max_abs, weight);
    return fz_check_hip(hipGetLastError(), "norm_weight launch");
}

int fz_launch_verdict(fz_ctx *ctx, const int32_t *target, const int32_t *observed, const int64_t *max_abs,
                      const int32_t *weight, size_t groups, int l, int64_t beta, int64_t omega, int *d_verdict) {
    hipLaunchKernelGGL(verdict_kernel, dim3((unsigned)groups), dim3(256), 0, ctx->stream, target, observed, ctx->degree, max_abs, weight,
                       l, beta, omega, d_verdict);
    return fz_check_hip(hipGetLastError(), "verdict launch");
}
