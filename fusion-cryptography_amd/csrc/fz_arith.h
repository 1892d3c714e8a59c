// fz_arith.h -- exact modular arithmetic on integers carried in IEEE doubles.
//
// Why doubles: measured on gfx950 (profiles/r01_instr_rate_microbench.txt) v_mul_lo_u32,
// v_mul_hi_u32, v_min_u32 and every fp64 op issue at the same ~4 cycles per wave64, so
// a Shoup/Montgomery int32 butterfly (3 multiplies + 3 range corrections + 5 adds) is not
// cheaper than an fp64 butterfly (6-op FMA-Barrett multiply + add + sub), and the fp64
// form needs NO range corrections: with a 31-bit modulus, 53-bit significands leave room
// for lazy accumulation across all eight butterfly stages and for arbitrary int32 inputs.
//
// Every function is exact (no rounding error reaches a result) under the stated operand
// bounds; results are therefore bit-identical to the reference's Python-integer arithmetic
// (algebra/ntt.py:93-123 `cent`, :276-290, :356-376).  The same header compiles for the host
// (tests/test_arith_host.py builds it with g++) so the bounds are exercised on the CPU.
#ifndef FZ_ARITH_H
#define FZ_ARITH_H

#if defined(__HIPCC__)
#define FZ_HD __host__ __device__ __forceinline__
#else
#define FZ_HD inline
#endif

struct FzMod {
    double q;      // modulus, odd, < 2^32 (centred residues are int32 whatever q is; the 4-op multiply only below 2^31)
    double qinv;   // 1.0 / q rounded to nearest
    // pseudo-Mersenne form q = K - delta, K = 2^k the smallest power of two >= q
    double K;      // 2^k
    double kq;     // K / q rounded to nearest (w2 = w * kq is the "quotient twiddle" of fz_mulmod4)
    double kappa;  // delta / K, exact (delta < 2^k, power-of-two scaling)
    double magic;  // 1.5 * 2^52 * K: adding it rounds to the nearest multiple of K
    int fast;      // 1 when delta < 2^15, i.e. fz_mulmod4 is exact for |a| <= 2^38
    double r32;    // 2^32 mod q (fz_cent_i64)
};

FZ_HD FzMod fz_make_mod(unsigned q) {
    FzMod m;
    m.q = (double)q;
    m.qinv = 1.0 / (double)q;
    unsigned long long K = 1;
    while (K < q) K <<= 1;
    m.K = (double)K;
    m.kq = (double)K / (double)q;
    m.kappa = (double)(K - q) / (double)K;
    m.magic = 6755399441055744.0 * (double)K;      // 1.5 * 2^52 * K
    m.fast = ((K - q) < 32768ull && q < 0x80000000u) ? 1 : 0;      // (fz_mulmod4's bounds are stated for K <= 2^31)
    m.r32 = (double)(4294967296ull % q);
    return m;
}

// r = a*b - c*q exactly, with c = rint(fl(fl(a*b) * qinv)).
// Exact when a, b are integers with |a*b| < 2^83 (so that |low part| < 2^30) -- in this
// library |a| < 2^40 and |b| < 2^32 (twiddles of a modulus below 2^32; operands are centred, < 2^31).
// |r| <= q/2 + q*|a*b/q|*2^-51  (i.e. "almost centred").
//   h = fl(a*b); l = a*b - h exactly (FMA);  h - c*q is an integer below 2^53 -> exact.
FZ_HD double fz_mulmod(double a, double b, const FzMod m) {
    double h = a * b;
    double l = __builtin_fma(a, b, -h);
    double c = __builtin_rint(h * m.qinv);
    double d = __builtin_fma(-c, m.q, h);
    return d + l;
}

// 4-op variant for pseudo-Mersenne moduli (q = K - delta, delta < 2^15; the scheme's prime is
// 2^31 - 17919).  w2 = fl(w * K / q) is precomputed per twiddle.
//   u  = fma(a, w2, magic)      -> magic + K*c,  c = round(a*w2/K) ~ round(a*w/q)   (ulp there is K)
//   cK = u - magic              -> K*c exactly
//   t  = fma(a, w, -cK)         -> a*w - c*K = (a*w - c*q) - c*delta: an integer below 2^53, exact
//   r  = fma(cK, kappa, t)      -> t + c*delta = a*w - c*q, exact
// Exact for integers |a| <= 2^38, 0 <= w < q (then |c|*delta + q < 2^53).  |r| <= q/2 + q*2^-13.
FZ_HD double fz_mulmod4(double a, double w, double w2, const FzMod m) {
    double u = __builtin_fma(a, w2, m.magic);
    double cK = u - m.magic;
    double t = __builtin_fma(a, w, -cK);
    return __builtin_fma(cK, m.kappa, t);
}

// Canonical centred residue of an integer-valued x, |x| < 2^19 * q (q < 2^31; |x| < 2^18 * q for q < 2^32: the margin to a
// half-integer is 1/(2q)):
// the unique r == x (mod q) with |r| <= (q-1)/2, i.e. the reference's cent(x).
// fl(x*qinv) is within 2^-33.. of x/q, and x/q is at least 1/(2q) > 2^-32 away from any
// half-integer (q odd), so rint picks the true nearest integer.
FZ_HD double fz_cent(double x, const FzMod m) {
    double c = __builtin_rint(x * m.qinv);
    return __builtin_fma(-c, m.q, x);
}

// cent() of an integer-valued double beyond fz_cent's 2^19 * q bound (lazily accumulated fp64 sums, |x| < 2^53):
// two steps -- the first brings |x| below ~q (exact FMA as above), the second makes it canonical.
FZ_HD double fz_cent_wide(double x, const FzMod m) {
    double c = __builtin_rint(x * m.qinv);
    double r = __builtin_fma(-c, m.q, x);
    return fz_cent(r, m);
}

// x - q * rint(x / q) for any integer-valued |x| < 2^80: NOT canonical (|r| <= q/2 + |x| * 2^-51: the quotient estimate is
// off by at most |x| / q * 2^-51; below 2^67, as in the aggregation kernel, that is q/2 + 2^16), but
// exact -- the FMA's true result is an integer below 2^33, hence representable.  Folds lazily accumulated sums.
FZ_HD double fz_fold(double x, const FzMod m) {
    return __builtin_fma(-__builtin_rint(x * m.qinv), m.q, x);
}

// cent() of ANY int64 (partial sums that crossed an all-reduce): converting v itself to double is exact only below
// 2^53, so split v = hi * 2^32 + lo (hi signed, lo unsigned 32-bit halves, both exact as doubles) and reduce
// hi * 2^32 with the exact FMA multiply: |hi * r32| < 2^62 is inside fz_mulmod's bound.
FZ_HD double fz_cent_i64(long long v, const FzMod m) {
    const double hi = (double)(int)(v >> 32);
    const double lo = (double)(unsigned)(v & 0xffffffffll);
    return fz_cent(fz_mulmod(hi, m.r32, m) + lo, m);                                  // |.| < q + 2^32 < 2^19 * q
}

// cent(a*b) for int32-range a, b: canonical.
FZ_HD double fz_mulmod_cent(double a, double b, const FzMod m) {
    return fz_cent(fz_mulmod(a, b, m), m);
}

// Integer multiply-add accumulation of A (.) y (v_mad_i64_i32): with A split into 16-bit halves, hi += y * (A >> 16) and
// lo += y * (A & 0xffff) are exact in int64 for up to 2^15 products of any int32 operands (|y * half| <= 2^47).
// fz_imad_total: hi * 2^16 + lo (mod q) as a double with |result| <= q + 2^18.  `small` (wave-uniform): at most 32 products went into each
// sum, so |hi|, |lo| < 2^52 convert to fp64 exactly and two folds do (12 operations); otherwise both sums are centred exactly
// as arbitrary int64 first.
FZ_HD double fz_imad_total(long long hi, long long lo, bool small, const FzMod m) {
    if (small) return fz_fold((double)hi * 65536.0, m) + fz_fold((double)lo, m);
    return fz_fold(fz_cent_i64(hi, m) * 65536.0, m) + fz_cent_i64(lo, m);
}

#endif  // FZ_ARITH_H
