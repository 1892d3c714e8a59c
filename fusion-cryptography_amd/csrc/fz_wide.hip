// fz_wide.hip -- the GENERIC-parameter path: any odd modulus 3 <= q < 2^63, any power-of-two length, 64-bit rows.
//
// The reference transforms any power-of-two length over any odd modulus with whatever twiddle table it is handed
// (algebra/ntt.py:239-290, :345-377; Python integers, no size limit), and its polynomial classes add, negate and multiply
// over any modulus (algebra/polynomials.py:140-216, :272-333).  The int32 path (fz_ntt.hip, fz_pointwise.hip) covers q < 2^32 and
// lengths up to 4096 -- every parameter set the scheme defines -- at full speed; this file is what stands behind the drop-in
// packages for everything else, so that no parameter the reference accepts below 2^63 is refused and none is computed on the
// CPU.  It is written for correctness, not bandwidth: int64 rows of centred residues in and out, canonical residues in
// [0, q) inside, Montgomery multiplication with R = 2^64 (two 64 x 64 -> 128-bit products and one conditional subtraction;
// twiddles are handed over in Montgomery form, so a butterfly's product needs no conversion), one workgroup per polynomial
// walking the reference's own loop nest stage by stage in global memory (a workgroup barrier per stage: any length).
// Bit-identical to the reference by construction (exact integer arithmetic, one cent() where the reference has one).
// tests/test_gpu_wide.py compares with the pure-Python loops of oracle/oracle.py on Python integers.
#include "fz_internal.h"
#include "../../include/fusion_hip_generic.h"
#include <algorithm>
#include <memory>
#include <new>

namespace {
#define FZW_TRY(x) do { int rc_ = (x); if (rc_ != FZ_OK) return rc_; } while (0)
#define FZW_HIP(call, what) FZW_TRY(fz_check_hip((call), what))

struct WMod {
    unsigned long long q, qneg_inv;     // -q^{-1} mod 2^64
    unsigned long long r2;              // 2^128 mod q (Montgomery form of 2^64: mont(x, r2) = x * 2^64 mod q)
    unsigned long long half;            // (q - 1) / 2
};

// a * b * 2^-64 mod q for a, b in [0, q), q odd < 2^63
__device__ __forceinline__ unsigned long long wmont(unsigned long long a, unsigned long long b, const WMod &m) {
    const unsigned long long lo = a * b, hi = __umul64hi(a, b);
    const unsigned long long k = lo * m.qneg_inv;
    // lo + (k * q mod 2^64) == 0 (mod 2^64): the carry into the high word is 1 unless lo == 0
    unsigned long long u = hi + __umul64hi(k, m.q) + (lo != 0ull);
    return u >= m.q ? u - m.q : u;
}
__device__ __forceinline__ unsigned long long wadd(unsigned long long a, unsigned long long b, const WMod &m) {
    const unsigned long long s = a + b;                   // < 2^64: a, b < q < 2^63
    return s >= m.q ? s - m.q : s;
}
__device__ __forceinline__ unsigned long long wsub(unsigned long long a, unsigned long long b, const WMod &m) {
    return a >= b ? a - b : a + m.q - b;
}
// ANY int64 -> its residue in [0, q) (the int32 entry points accept unreduced rows too; q < 2^63 fits the signed type)
__device__ __forceinline__ unsigned long long wcanon(long long x, const WMod &m) {
    const long long r = x % (long long)m.q;
    return (unsigned long long)(r < 0 ? r + (long long)m.q : r);
}
__device__ __forceinline__ long long wcent(unsigned long long v, const WMod &m) {          // the reference's cent(): ntt.py:93-123
    return v > m.half ? (long long)v - (long long)m.q : (long long)v;
}

// forward: cooley_tukey_ntt's loop nest (ntt.py:274-290) -- stage with m blocks of 2t: butterfly (j, j + t) of block i uses table[m + i];
// inverse: gentleman_sande_intt's (ntt.py:354-376) -- stage with h = m/2 blocks: table[h + i], then the scaling by n^{-1}.
// tab: the caller's table, Montgomery form.  One workgroup per row; the row lives in `out` (canonical residues reinterpreted).
__global__ __launch_bounds__(256) void wide_ntt_kernel(const long long *in, long long *out, size_t batch, int n, const unsigned long long *tab,
                                                       unsigned long long n_inv_mont, int inverse, WMod m) {
  for (size_t rowi = blockIdx.x; rowi < batch; rowi += gridDim.x) {       // (a launch of more than 2^24 workgroups of 256 is refused by HIP)
    const long long *src = in + rowi * (size_t)n;
    unsigned long long *v = reinterpret_cast<unsigned long long *>(out + rowi * (size_t)n);
    for (int j = threadIdx.x; j < n; j += blockDim.x) v[j] = wcanon(src[j], m);
    __syncthreads();
    const int half = n / 2;
    if (!inverse) {
        for (int mm = 1, t = half; mm < n; mm *= 2, t /= 2) {
            for (int b = threadIdx.x; b < half; b += blockDim.x) {
                const int i = b / t, jj = b % t, j = 2 * i * t + jj;
                const unsigned long long U = v[j], V = wmont(v[j + t], tab[mm + i], m);
                v[j] = wadd(U, V, m);
                v[j + t] = wsub(U, V, m);
            }
            __syncthreads();
        }
    } else {
        for (int mm = n, t = 1; mm > 1; mm /= 2, t *= 2) {
            const int h = mm / 2;
            for (int b = threadIdx.x; b < half; b += blockDim.x) {
                const int i = b / t, jj = b % t, j = 2 * i * t + jj;
                const unsigned long long U = v[j], V = v[j + t];
                v[j] = wadd(U, V, m);
                v[j + t] = wmont(wsub(U, V, m), tab[h + i], m);
            }
            __syncthreads();
        }
    }
    long long *dst = out + rowi * (size_t)n;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        unsigned long long x = v[j];
        if (inverse) x = wmont(x, n_inv_mont, m);
        dst[j] = wcent(x, m);
    }
    __syncthreads();
  }
}

// op: FZ_OP_MUL / ADD / SUB / NEG as in fusion_hip.h.  NEG is the reference's -(x mod q) in [-(q-1), 0] (polynomials.py:155-163)
__global__ __launch_bounds__(256) void wide_pw_kernel(int op, const long long *a, const long long *b, long long *out, size_t count, WMod m) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const unsigned long long x = wcanon(a[i], m);
        if (op == FZ_OP_NEG) { out[i] = -(long long)x; continue; }
        const unsigned long long y = wcanon(b[i], m);
        unsigned long long r;
        if (op == FZ_OP_ADD) r = wadd(x, y, m);
        else if (op == FZ_OP_SUB) r = wsub(x, y, m);
        else r = wmont(wmont(x, y, m), m.r2, m);           // x * y * 2^-64, then * 2^64
        out[i] = wcent(r, m);
    }
}

// out[b][j] = cent(sum_k A[k][j] * S[b][k][j]): the (1 x l) . (l x 1) product of matrices.py:143-181
__global__ __launch_bounds__(256) void wide_matvec_kernel(const long long *A, const long long *S, long long *out, size_t batch, int l, int d, WMod m) {
    const size_t total = batch * (size_t)d, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const size_t b = i / d;
        const int j = (int)(i % d);
        unsigned long long acc = 0;
        for (int k = 0; k < l; ++k)
            acc = wadd(acc, wmont(wcanon(A[(size_t)k * d + j], m), wcanon(S[(b * l + k) * (size_t)d + j], m), m), m);
        out[i] = wcent(wmont(acc, m.r2, m), m);
    }
}

// per row: max |x| over the stored values (unsigned: |INT64_MIN| = 2^63) and the number of non-zero ones (polynomials.py:221-229)
__global__ __launch_bounds__(256) void wide_norm_weight_kernel(const long long *rows, size_t batch, int d, unsigned long long *mx, int *wt) {
    __shared__ unsigned long long s_mx[256];
    __shared__ int s_wt[256];
    for (size_t rowi = blockIdx.x; rowi < batch; rowi += gridDim.x) {
        const long long *r = rows + rowi * (size_t)d;
        unsigned long long best = 0;
        int cnt = 0;
        for (int j = threadIdx.x; j < d; j += blockDim.x) {
            const long long x = r[j];
            const unsigned long long ax = x < 0 ? 0ull - (unsigned long long)x : (unsigned long long)x;
            best = ax > best ? ax : best;
            cnt += x != 0;
        }
        s_mx[threadIdx.x] = best;
        s_wt[threadIdx.x] = cnt;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) {
                s_mx[threadIdx.x] = s_mx[threadIdx.x + s] > s_mx[threadIdx.x] ? s_mx[threadIdx.x + s] : s_mx[threadIdx.x];
                s_wt[threadIdx.x] += s_wt[threadIdx.x + s];
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) { mx[rowi] = s_mx[0]; wt[rowi] = s_wt[0]; }
        __syncthreads();
    }
}

constexpr size_t kWideMaxGrid = (size_t)1 << 20;          // workgroups per launch; rows beyond are walked by the same workgroups

int make_mod(uint64_t q, WMod *m) {
    if (q < 3 || (q & 1) == 0 || q >= (1ull << 63))
        return fz_set_error(FZ_E_UNSUPPORTED, "wide path: the modulus must be odd, 3 <= q < 2^63 (got %llu)", (unsigned long long)q);
    unsigned long long inv = q;                            // Newton: inv * q == 1 (mod 2^64); correct to 3 bits at the start (q odd)
    for (int i = 0; i < 6; ++i) inv *= 2 - q * inv;
    m->q = q;
    m->qneg_inv = 0ull - inv;
    const unsigned __int128 r = ((unsigned __int128)1 << 64) % q;
    m->r2 = (unsigned long long)((r * r) % q);
    m->half = (q - 1) / 2;
    return FZ_OK;
}

int use_device(int device) {                               // the same answer fz_ctx_create gives on a machine without a GPU
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fz_set_error(FZ_E_NODEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return fz_set_error(FZ_E_BADARG, "device %d out of range (%d devices)", device, ndev);
    return fz_check_hip(hipSetDevice(device), "hipSetDevice");
}

struct DevBuf {                                            // a device allocation for the duration of one call
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) { return fz_check_hip(hipMalloc(&p, bytes ? bytes : 1), "wide path: hipMalloc"); }
};
}  // namespace

extern "C" {

FZ_API int fz_wide_ntt_host(int device, uint64_t q, int degree, const uint64_t *h_table, uint64_t n_inv, int inverse,
                            const int64_t *h_in, int64_t *h_out, size_t batch) {
    if (degree < 2 || (degree & (degree - 1)) != 0) return fz_set_error(FZ_E_BADARG, "wide path: the length must be a power of two >= 2 (got %d)", degree);
    if (!h_table || (batch && (!h_in || !h_out))) return fz_set_error(FZ_E_BADARG, "NULL argument");
    WMod m;
    FZW_TRY(make_mod(q, &m));
    if (batch == 0) return FZ_OK;
    if (batch > 0x7fffffffull) return fz_set_error(FZ_E_BADARG, "wide path: at most 2^31 - 1 rows per call");
    FZW_TRY(use_device(device));
    // the table in Montgomery form: w * 2^64 mod q (host 128-bit arithmetic: a parameter conversion, once per call)
    std::unique_ptr<unsigned long long[]> tab(new (std::nothrow) unsigned long long[degree]);      // (no exception leaves the C ABI)
    if (!tab) return fz_set_error(FZ_E_HIP, "wide path: out of host memory for a table of %d entries", degree);
    for (int i = 0; i < degree; ++i) tab[i] = (unsigned long long)((((unsigned __int128)(h_table[i] % q)) << 64) % q);
    const unsigned long long n_inv_mont = (unsigned long long)((((unsigned __int128)(n_inv % q)) << 64) % q);
    const size_t bytes = batch * (size_t)degree * sizeof(int64_t);
    DevBuf din, dout, dtab;
    FZW_TRY(din.alloc(bytes));
    FZW_TRY(dout.alloc(bytes));
    FZW_TRY(dtab.alloc((size_t)degree * 8));
    FZW_HIP(hipMemcpy(din.p, h_in, bytes, hipMemcpyHostToDevice), "wide path: copy in");
    FZW_HIP(hipMemcpy(dtab.p, tab.get(), (size_t)degree * 8, hipMemcpyHostToDevice), "wide path: table");
    hipLaunchKernelGGL(wide_ntt_kernel, dim3((unsigned)std::min<size_t>(batch, kWideMaxGrid)), dim3(256), 0, 0, (const long long *)din.p, (long long *)dout.p, batch, degree,
                       (const unsigned long long *)dtab.p, n_inv_mont, inverse ? 1 : 0, m);
    FZW_HIP(hipGetLastError(), "wide transform launch");
    return fz_check_hip(hipMemcpy(h_out, dout.p, bytes, hipMemcpyDeviceToHost), "wide path: copy out");
}

FZ_API int fz_wide_pw_host(int device, uint64_t q, int op, const int64_t *h_a, const int64_t *h_b, int64_t *h_out, size_t count) {
    if (op < FZ_OP_MUL || op > FZ_OP_NEG) return fz_set_error(FZ_E_BADARG, "bad op %d", op);
    if (count && (!h_a || !h_out || (op != FZ_OP_NEG && !h_b))) return fz_set_error(FZ_E_BADARG, "NULL argument");
    WMod m;
    FZW_TRY(make_mod(q, &m));
    if (count == 0) return FZ_OK;
    FZW_TRY(use_device(device));
    const size_t bytes = count * sizeof(int64_t);
    DevBuf da, db, dout;
    FZW_TRY(da.alloc(bytes));
    FZW_TRY(db.alloc(bytes));
    FZW_TRY(dout.alloc(bytes));
    FZW_HIP(hipMemcpy(da.p, h_a, bytes, hipMemcpyHostToDevice), "wide path: copy in");
    if (op != FZ_OP_NEG) FZW_HIP(hipMemcpy(db.p, h_b, bytes, hipMemcpyHostToDevice), "wide path: copy in");
    const unsigned grid = (unsigned)std::min<size_t>((count + 255) / 256, 65535);
    hipLaunchKernelGGL(wide_pw_kernel, dim3(grid), dim3(256), 0, 0, op, (const long long *)da.p, (const long long *)db.p, (long long *)dout.p, count, m);
    FZW_HIP(hipGetLastError(), "wide pointwise launch");
    return fz_check_hip(hipMemcpy(h_out, dout.p, bytes, hipMemcpyDeviceToHost), "wide path: copy out");
}

FZ_API int fz_wide_matvec_host(int device, uint64_t q, int degree, const int64_t *h_A, const int64_t *h_S, int64_t *h_out, size_t batch, int l) {
    if (degree < 1 || l < 1) return fz_set_error(FZ_E_BADARG, "wide path: degree and l must be positive");
    if (batch && (!h_A || !h_S || !h_out)) return fz_set_error(FZ_E_BADARG, "NULL argument");
    WMod m;
    FZW_TRY(make_mod(q, &m));
    if (batch == 0) return FZ_OK;
    FZW_TRY(use_device(device));
    const size_t row = (size_t)degree * sizeof(int64_t);
    DevBuf dA, dS, dout;
    FZW_TRY(dA.alloc((size_t)l * row));
    FZW_TRY(dS.alloc(batch * (size_t)l * row));
    FZW_TRY(dout.alloc(batch * row));
    FZW_HIP(hipMemcpy(dA.p, h_A, (size_t)l * row, hipMemcpyHostToDevice), "wide path: copy in");
    FZW_HIP(hipMemcpy(dS.p, h_S, batch * (size_t)l * row, hipMemcpyHostToDevice), "wide path: copy in");
    const size_t total = batch * (size_t)degree;
    const unsigned grid = (unsigned)std::min<size_t>((total + 255) / 256, 65535);
    hipLaunchKernelGGL(wide_matvec_kernel, dim3(grid), dim3(256), 0, 0, (const long long *)dA.p, (const long long *)dS.p, (long long *)dout.p, batch, l, degree, m);
    FZW_HIP(hipGetLastError(), "wide matvec launch");
    return fz_check_hip(hipMemcpy(h_out, dout.p, batch * row, hipMemcpyDeviceToHost), "wide path: copy out");
}

FZ_API int fz_wide_norm_weight_host(int device, const int64_t *h_rows, size_t batch, int degree, uint64_t *h_max_abs, int32_t *h_weight) {
    if (degree < 1) return fz_set_error(FZ_E_BADARG, "wide path: degree must be positive");
    if (batch && (!h_rows || !h_max_abs || !h_weight)) return fz_set_error(FZ_E_BADARG, "NULL argument");
    if (batch == 0) return FZ_OK;
    if (batch > 0x7fffffffull) return fz_set_error(FZ_E_BADARG, "wide path: at most 2^31 - 1 rows per call");
    FZW_TRY(use_device(device));
    DevBuf drows, dmx, dwt;
    FZW_TRY(drows.alloc(batch * (size_t)degree * 8));
    FZW_TRY(dmx.alloc(batch * 8));
    FZW_TRY(dwt.alloc(batch * 4));
    FZW_HIP(hipMemcpy(drows.p, h_rows, batch * (size_t)degree * 8, hipMemcpyHostToDevice), "wide path: copy in");
    hipLaunchKernelGGL(wide_norm_weight_kernel, dim3((unsigned)std::min<size_t>(batch, kWideMaxGrid)), dim3(256), 0, 0, (const long long *)drows.p, batch, degree,
                       (unsigned long long *)dmx.p, (int *)dwt.p);
    FZW_HIP(hipGetLastError(), "wide norm / weight launch");
    FZW_HIP(hipMemcpy(h_max_abs, dmx.p, batch * 8, hipMemcpyDeviceToHost), "wide path: copy out");
    return fz_check_hip(hipMemcpy(h_weight, dwt.p, batch * 4, hipMemcpyDeviceToHost), "wide path: copy out");
}

}  // extern "C"
