// fz_challenge.hip -- the per-signer challenge pipeline ON THE DEVICE (device half).
//
// hash_ch (fusion/fusion.py:511-531) for N independent (key, message) pairs:
//     x  = sign_hash_dst + "," + str(vk) + "," + str(int(prehash))                      fusion.py:412-419
//     b  = SHAKE-256(x), as many bytes as the decoder consumes                          fusion.py:515-524
//     c  = decode_bytes_to_polynomial_coefficients(b, ...)                              fusion.py:422-481
//     c^ = NTT(c)                                                                       fusion.py:499-507
// The text of str(vk) is the one the reference hashes (fusion.py:328-329 -> algebra/matrices.py:40-41 ->
// algebra/polynomials.py:257-258).  Verification keys never leave the device as text: three kernels,
//   1. vk_text_kernel      one wave per signer: the exact ASCII text (the decimal of the 256-bit pre-hashed message
//                          included), padded (SHAKE suffix 0x1f ... 0x80) to whole 136-byte blocks, written to a scratch
//                          row; integers are formatted where they are stored
//   2. shake_kernel        TWO lanes per signer, one holding the low and one the high 32 bits of every Keccak lane:
//                          bitwise steps are independent 32-bit operations, a 64-bit rotation is one v_alignbit of
//                          (own half, partner's half) with the partner's half fetched by a DPP quad swap -- the same
//                          instruction stream for both lanes.  119 vector operations per round and lane instead of
//                          ~200 for a whole state per lane: the chain of ~108 permutations per signer (47 absorbed
//                          blocks + 61 squeezed) is the latency of the whole pipeline, and Keccak offers no more
//                          parallelism inside a permutation without bit-slicing overheads that cancel it.
//                          Output words leave transposed ([word][signer]) so that every access is coalesced.
//   3. decode_kernel       one lane per signer: bit-string of signs, then the partial Fisher-Yates shuffle driven by
//                          33-byte big-endian integers reduced mod (i + 1) -- as byte dot products with a weight table
//                          (v_dot4_u32_u8), all indices first (independent, loads three iterations ahead), then the
//                          swaps; stream positions are the same for every signer (scalar registers), shuffle state
//                          and indices in LDS ([position][lane]).
// then the ordinary forward transform (fz_ntt.hip).  Supported: the scheme's parameter sets (norm bound 1, i.e. ternary
// challenges; degree <= 256); anything else returns FZ_E_UNSUPPORTED and callers use the host pipeline (fz_host.cpp).
#include "fz_internal.h"
#include "../../include/fusion_hip.h"
#include "fz_keccak_wave.h"
#include <algorithm>

namespace {

constexpr int kRate = 136;                   // SHAKE-256 rate in bytes (17 lanes of 64 bits)

// ---------------------------------------------------------------------------------------------------------------
// 1. text
// ---------------------------------------------------------------------------------------------------------------
struct VkTextParts {                         // the fixed pieces of the text, built on the host once per call
    char s0[384], s1[384], s2[16];           // before the left values / between left and right / after the right values
    int n0, n1, n2;
};

__device__ __forceinline__ int dec_len(int v) {          // len(str(v)) for an int32
    unsigned u = v < 0 ? 0u - (unsigned)v : (unsigned)v;
    int n = (v < 0) ? 2 : 1;
    n += u >= 10u; n += u >= 100u; n += u >= 1000u; n += u >= 10000u; n += u >= 100000u; n += u >= 1000000u;
    n += u >= 10000000u; n += u >= 100000000u; n += u >= 1000000000u;
    return n;
}

// str(int) of a 256-bit integer (eight 32-bit limbs, least significant first) in base 10^9: chunks out[0..n-1], least
// significant first; returns n (1..9).  Nine rounds of an eight-limb short division: a sequential chain, one LANE's work.
__device__ __forceinline__ int u256_to_base1e9(uint32_t (&limb)[8], uint32_t *out) {
    int n = 0;
    bool nz = true;
    while (nz && n < 9) {
        unsigned long long rem = 0;
        nz = false;
#pragma unroll
        for (int t = 7; t >= 0; --t) {
            const unsigned long long cur = (rem << 32) | limb[t];
            limb[t] = (uint32_t)(cur / 1000000000ull);
            rem = cur % 1000000000ull;
            nz |= limb[t] != 0;
        }
        out[n++] = (uint32_t)rem;
    }
    return n;
}
constexpr int kDecStride = 16;               // uint32 per signer in the decimal scratch: 9 chunks, the chunk count at [9]

// The text of ONE signer by ONE wave, into `buf` (LDS, `cap` bytes, a multiple of 16): the exact characters, the SHAKE
// suffix (0x1f ... 0x80) and zeros up to the end of the last 136-byte block; returns the number of blocks (all lanes).
// aux[0..9]: the pre-hashed integer in base 10^9 (chunks, least significant first) and the chunk count at [9].
// What the text of one signer needs from memory, per lane: its values of the key row (2 * degree of them over the wave, at
// most 8 per lane) and its bytes of the three fixed pieces.  Requested by the caller as early as it can: a load where the byte
// is written is a round trip per loop iteration (most of the 9 us the text took for one signer in round 5's first form).
struct VkTextIn {
    int32_t vals[8];
    uint8_t f0[6], f1[6], f2;
};
__device__ __forceinline__ void vk_text_load(VkTextIn &in, const int32_t *row, int degree, const VkTextParts &T, int lane) {
    const int nvals = 2 * degree;
    const int vpl = nvals >= 64 ? nvals / 64 : 1;
    const int k0 = lane * vpl;
#pragma unroll
    for (int t = 0; t < 8; ++t) in.vals[t] = (t < vpl && k0 + t < nvals) ? row[k0 + t] : 0;
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        in.f0[t] = lane + 64 * t < T.n0 ? (uint8_t)T.s0[lane + 64 * t] : (uint8_t)0;
        in.f1[t] = lane + 64 * t < T.n1 ? (uint8_t)T.s1[lane + 64 * t] : (uint8_t)0;
    }
    in.f2 = lane < T.n2 ? (uint8_t)T.s2[lane] : (uint8_t)0;
}

__device__ __forceinline__ int vk_text_wave(uint8_t *buf, size_t cap, const uint32_t *aux, const VkTextIn &in, int degree, const VkTextParts &T,
                                            int lane) {
    const int nvals = 2 * degree;
    const int vpl = nvals >= 64 ? nvals / 64 : 1;               // values per lane, at most 8 (a lane never straddles the two halves)
    const int k0 = lane * vpl;
    const int32_t (&vals)[8] = in.vals;
    const uint8_t (&f0)[6] = in.f0, (&f1)[6] = in.f1;
    const uint8_t f2 = in.f2;
    for (size_t o = (size_t)lane * 16; o < cap; o += 64 * 16) *reinterpret_cast<int4 *>(buf + o) = make_int4(0, 0, 0, 0);
    // pass 1: lengths (digits + ", " unless last of its half)
    int mine = 0;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int k = k0 + t;
        if (t < vpl && k < nvals) mine += dec_len(vals[t]) + (((k + 1) % degree) ? 2 : 0);
    }
    int incl = mine;                                            // inclusive scan over the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
    }
    const int total_vals = __shfl(incl, 63);
    const int left_total = (nvals >= 64) ? __shfl(incl, 31) : __shfl(incl, degree - 1);
    int pos = T.n0 + (incl - mine) + ((k0 >= degree) ? T.n1 : 0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // the zero fill above before the byte writes below
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // pass 2: characters, last digit first
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int k = k0 + t;
        if (t < vpl && k < nvals) {
            const int v = vals[t];
            const int n = dec_len(v);
            unsigned u = v < 0 ? 0u - (unsigned)v : (unsigned)v;
            int p = pos + n;
            do {
                buf[--p] = (uint8_t)('0' + u % 10u);
                u /= 10u;
            } while (u);
            if (v < 0) buf[--p] = '-';
            pos += n;
            if ((k + 1) % degree) { buf[pos] = ','; buf[pos + 1] = ' '; pos += 2; }
        }
    }
    // fixed pieces and the decimal of the pre-hashed message (lane c writes chunk c: 9 digits, the leading chunk as many
    // as it has)
    const int nch = (int)aux[9];
    const int top = dec_len((int)aux[nch - 1]);                  // < 10^9: fits an int
    const int tl = top + 9 * (nch - 1);
    const int at1 = T.n0 + left_total, at2 = T.n0 + T.n1 + total_vals, at3 = at2 + T.n2;
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const int c = lane + 64 * t;
        if (c < T.n0) buf[c] = f0[t];
        if (c < T.n1) buf[at1 + c] = f1[t];
    }
    if (lane < T.n2) buf[at2 + lane] = f2;
    if (lane < nch) {
        unsigned u = aux[lane];
        const int nd = (lane == nch - 1) ? top : 9;
        int p = at3 + tl - 9 * lane;                             // one past this chunk's last digit
        for (int k = 0; k < nd; ++k) {
            buf[--p] = (uint8_t)('0' + u % 10u);
            u /= 10u;
        }
    }
    const int len = at3 + tl;
    const int nb = len / kRate + 1;                             // pad10*1 always adds at least one byte
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (lane == 0) {
        buf[len] ^= 0x1f;                                       // SHAKE domain bits + first pad bit (FIPS 202)
        buf[nb * kRate - 1] ^= 0x80;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return nb;
}

// one wave per signer, kTextWaves signers per workgroup.  text row i: blocks * 136 bytes, *nblocks = blocks.
// dec (optional): the pre-hashed integers already in base 10^9 (prehash_kernel converts them with a lane per message; done
// here it is one lane of the signer's wave while 63 wait: a third of this kernel's time)
constexpr int kTextWaves = 4;
__global__ __launch_bounds__(64 * kTextWaves) void vk_text_kernel(const int32_t *vk, size_t vk_stride, const uint8_t *pre,
                                                                  const uint32_t *dec, size_t N, int degree, VkTextParts T,
                                                                  uint8_t *text, size_t text_stride, int *nblocks) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t i = (size_t)blockIdx.x * kTextWaves + wave;
    if (i >= N) return;
    uint8_t *buf = smem + (size_t)wave * text_stride;
    uint32_t *aux = reinterpret_cast<uint32_t *>(smem + (size_t)kTextWaves * text_stride) + wave * 16;
    // str(int.from_bytes(prehash, "little")) (fusion.py:405-409, :416-418): the 256-bit integer in base 10^9, least
    // significant chunk first (one lane: 9 rounds of an 8-limb short division), digits written by lanes 0..8
    if (dec) {
        if (lane < 10) aux[lane] = dec[i * kDecStride + lane];
    } else if (lane == 0) {
        uint32_t limb[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const uint8_t *b = pre + i * 32 + 4 * t;
            limb[t] = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
        }
        aux[9] = (uint32_t)u256_to_base1e9(limb, aux);
    }
    VkTextIn tin;
    vk_text_load(tin, vk + i * vk_stride, degree, T, lane);
    const int nb = vk_text_wave(buf, text_stride, aux, tin, degree, T, lane);
    if (lane == 0) nblocks[i] = nb;
    uint8_t *dst = text + i * text_stride;
    for (int o = lane * 8; o < nb * kRate; o += 64 * 8) *reinterpret_cast<uint2 *>(dst + o) = *reinterpret_cast<const uint2 *>(buf + o);
}

// ---------------------------------------------------------------------------------------------------------------
// 2. Keccak-f[1600] on two lanes per state (low halves in the even lane, high halves in the odd lane)
// ---------------------------------------------------------------------------------------------------------------
__constant__ uint32_t kRC[24][2] = {
    {0x00000001u, 0x00000000u}, {0x00008082u, 0x00000000u}, {0x0000808au, 0x80000000u}, {0x80008000u, 0x80000000u},
    {0x0000808bu, 0x00000000u}, {0x80000001u, 0x00000000u}, {0x80008081u, 0x80000000u}, {0x00008009u, 0x80000000u},
    {0x0000008au, 0x00000000u}, {0x00000088u, 0x00000000u}, {0x80008009u, 0x00000000u}, {0x8000000au, 0x00000000u},
    {0x8000808bu, 0x00000000u}, {0x0000008bu, 0x80000000u}, {0x00008089u, 0x80000000u}, {0x00008003u, 0x80000000u},
    {0x00008002u, 0x80000000u}, {0x00000080u, 0x80000000u}, {0x0000800au, 0x00000000u}, {0x8000000au, 0x80000000u},
    {0x80008081u, 0x80000000u}, {0x00008080u, 0x80000000u}, {0x80000001u, 0x00000000u}, {0x80008008u, 0x80000000u}};

__device__ __forceinline__ uint32_t partner(uint32_t x) {       // the other half of the same 64-bit lane: quad_perm [1,0,3,2]
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);
}

// this lane's half of rotl64(X, R), X = (own half, partner's half):  R < 32: (x << R) | (px >> (32 - R)) for BOTH lanes
// (low lane: lo' = lo << R | hi >> (32-R); high lane: hi' = hi << R | lo >> (32-R)); R >= 32 swaps the roles first
template <int R>
__device__ __forceinline__ uint32_t rotl64_half(uint32_t x, uint32_t px) {
    if (R == 0) return x;
    if (R == 32) return px;
    if (R < 32) return __builtin_amdgcn_alignbit(x, px, 32 - R);
    return __builtin_amdgcn_alignbit(px, x, 64 - R);
}

// three-input XOR in ONE instruction (v_bitop3_b32, truth table 0x96; the compiler keeps two v_xor otherwise)
#define FZ_X3(a, b, c) ((uint32_t)__builtin_amdgcn_bitop3_b32((a), (b), (c), 0x96))
// five rotated lanes of one output row: the five inputs first, then their five partner fetches, then the five
// rotations -- a DPP read of a register written by the previous instruction costs two wait states, and the compiler fills
// them with s_nop (36 per round pair in the first version) unless independent work sits in between
#define FZ_ROW(A0_, R0, A1_, R1, A2_, R2, A3_, R3, A4_, R4) \
    { const uint32_t t0 = (A0_), t1 = (A1_), t2 = (A2_), t3 = (A3_), t4 = (A4_); \
      const uint32_t p0 = partner(t0), p1 = partner(t1), p2 = partner(t2), p3 = partner(t3), p4 = partner(t4); \
      b0 = rotl64_half<R0>(t0, p0); b1 = rotl64_half<R1>(t1, p1); b2 = rotl64_half<R2>(t2, p2); \
      b3 = rotl64_half<R3>(t3, p3); b4 = rotl64_half<R4>(t4, p4); }
// one round A -> E (theta, rho + pi, chi, iota); every index a compile-time constant.  theta's column parities take two
// 3-input XORs each, and A ^ D = A ^ C[x-1] ^ rot(C[x+1], 1) is one more, so D is never formed: 119 vector operations per
// round and lane (10 parities, 10 for the five rotated parities, 25 + 48 for rho and pi, 25 chi, 1 iota)
#define FZ_KROUND32(A, E, rc) { \
    const uint32_t c0 = FZ_X3(FZ_X3(A##0, A##5, A##10), A##15, A##20), c1 = FZ_X3(FZ_X3(A##1, A##6, A##11), A##16, A##21), \
                   c2 = FZ_X3(FZ_X3(A##2, A##7, A##12), A##17, A##22), c3 = FZ_X3(FZ_X3(A##3, A##8, A##13), A##18, A##23), \
                   c4 = FZ_X3(FZ_X3(A##4, A##9, A##14), A##19, A##24); \
    uint32_t r0, r1, r2, r3, r4; \
    { const uint32_t q0 = partner(c0), q1 = partner(c1), q2 = partner(c2), q3 = partner(c3), q4 = partner(c4); \
      r0 = rotl64_half<1>(c0, q0); r1 = rotl64_half<1>(c1, q1); r2 = rotl64_half<1>(c2, q2); r3 = rotl64_half<1>(c3, q3); \
      r4 = rotl64_half<1>(c4, q4); } \
    /* d0 = c4 ^ r1, d1 = c0 ^ r2, d2 = c1 ^ r3, d3 = c2 ^ r4, d4 = c3 ^ r0 */ \
    uint32_t b0, b1, b2, b3, b4; \
    FZ_ROW(FZ_X3(A##0, c4, r1), 0, FZ_X3(A##6, c0, r2), 44, FZ_X3(A##12, c1, r3), 43, FZ_X3(A##18, c2, r4), 21, FZ_X3(A##24, c3, r0), 14) \
    E##0 = b0 ^ (~b1 & b2) ^ (rc); E##1 = b1 ^ (~b2 & b3); E##2 = b2 ^ (~b3 & b4); E##3 = b3 ^ (~b4 & b0); E##4 = b4 ^ (~b0 & b1); \
    FZ_ROW(FZ_X3(A##3, c2, r4), 28, FZ_X3(A##9, c3, r0), 20, FZ_X3(A##10, c4, r1), 3, FZ_X3(A##16, c0, r2), 45, FZ_X3(A##22, c1, r3), 61) \
    E##5 = b0 ^ (~b1 & b2); E##6 = b1 ^ (~b2 & b3); E##7 = b2 ^ (~b3 & b4); E##8 = b3 ^ (~b4 & b0); E##9 = b4 ^ (~b0 & b1); \
    FZ_ROW(FZ_X3(A##1, c0, r2), 1, FZ_X3(A##7, c1, r3), 6, FZ_X3(A##13, c2, r4), 25, FZ_X3(A##19, c3, r0), 8, FZ_X3(A##20, c4, r1), 18) \
    E##10 = b0 ^ (~b1 & b2); E##11 = b1 ^ (~b2 & b3); E##12 = b2 ^ (~b3 & b4); E##13 = b3 ^ (~b4 & b0); E##14 = b4 ^ (~b0 & b1); \
    FZ_ROW(FZ_X3(A##4, c3, r0), 27, FZ_X3(A##5, c4, r1), 36, FZ_X3(A##11, c0, r2), 10, FZ_X3(A##17, c1, r3), 15, FZ_X3(A##23, c2, r4), 56) \
    E##15 = b0 ^ (~b1 & b2); E##16 = b1 ^ (~b2 & b3); E##17 = b2 ^ (~b3 & b4); E##18 = b3 ^ (~b4 & b0); E##19 = b4 ^ (~b0 & b1); \
    FZ_ROW(FZ_X3(A##2, c1, r3), 62, FZ_X3(A##8, c2, r4), 55, FZ_X3(A##14, c3, r0), 39, FZ_X3(A##15, c4, r1), 41, FZ_X3(A##21, c0, r2), 2) \
    E##20 = b0 ^ (~b1 & b2); E##21 = b1 ^ (~b2 & b3); E##22 = b2 ^ (~b3 & b4); E##23 = b3 ^ (~b4 & b0); E##24 = b4 ^ (~b0 & b1); }

struct KState {
    uint32_t a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13, a14, a15, a16, a17, a18, a19, a20, a21, a22, a23, a24;
};

__device__ __forceinline__ void keccak_f_half(KState &S, int half) {
    uint32_t a0 = S.a0, a1 = S.a1, a2 = S.a2, a3 = S.a3, a4 = S.a4, a5 = S.a5, a6 = S.a6, a7 = S.a7, a8 = S.a8, a9 = S.a9,
             a10 = S.a10, a11 = S.a11, a12 = S.a12, a13 = S.a13, a14 = S.a14, a15 = S.a15, a16 = S.a16, a17 = S.a17,
             a18 = S.a18, a19 = S.a19, a20 = S.a20, a21 = S.a21, a22 = S.a22, a23 = S.a23, a24 = S.a24;
    uint32_t e0, e1, e2, e3, e4, e5, e6, e7, e8, e9, e10, e11, e12, e13, e14, e15, e16, e17, e18, e19, e20, e21, e22, e23, e24;
    // both halves of a round constant by SCALAR loads (the round index is uniform), the lane's half by a bitwise select (a
    // per-lane load put a vector-memory wait inside every round pair); the next pair's constants are requested before
    // this pair's rounds, so their latency is covered
    const uint32_t hmask = 0u - (uint32_t)half;
    uint32_t lo0 = kRC[0][0], hi0 = kRC[0][1], lo1 = kRC[1][0], hi1 = kRC[1][1];
#pragma unroll 1
    for (int r = 0; r < 24; r += 2) {
        const uint32_t rc0 = lo0 ^ ((lo0 ^ hi0) & hmask), rc1 = lo1 ^ ((lo1 ^ hi1) & hmask);
        const int rn = (r + 2) % 24;
        lo0 = kRC[rn][0]; hi0 = kRC[rn][1]; lo1 = kRC[rn + 1][0]; hi1 = kRC[rn + 1][1];
        FZ_KROUND32(a, e, rc0)
        FZ_KROUND32(e, a, rc1)
    }
    S.a0 = a0; S.a1 = a1; S.a2 = a2; S.a3 = a3; S.a4 = a4; S.a5 = a5; S.a6 = a6; S.a7 = a7; S.a8 = a8; S.a9 = a9;
    S.a10 = a10; S.a11 = a11; S.a12 = a12; S.a13 = a13; S.a14 = a14; S.a15 = a15; S.a16 = a16; S.a17 = a17; S.a18 = a18;
    S.a19 = a19; S.a20 = a20; S.a21 = a21; S.a22 = a22; S.a23 = a23; S.a24 = a24;
}

#define FZ_FOR17(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15) M(16)

// text rows [N][text_stride] (padded blocks), nblocks [N]  ->  xof words, transposed: word k (32 bits: 64-bit lane k/2 of
// the stream, half k%2) of signer s at xof[k * xstride + s]
// Workgroups of WAVES independent waves.  More waves than CUs: four per workgroup -- they go to the four SIMDs of a CU,
// whereas four one-wave workgroups on a CU may share a SIMD (1024 of them ran at twice the time of 256: two chains on one
// issue port).  Fewer: one per workgroup, so that every chain has a CU (and its instruction fetch) to itself (669 us
// against 693 us).
template <int kShakeWaves>
__global__ __launch_bounds__(64 * kShakeWaves) void shake_kernel(const uint8_t *text, size_t text_stride, const int *nblocks, size_t N,
                                                                 int max_blocks, int out_blocks, uint32_t *xof, size_t xstride) {
    const int lane = threadIdx.x & 63, half = lane & 1;
    const size_t s_raw = ((size_t)blockIdx.x * kShakeWaves + (threadIdx.x >> 6)) * 32 + (lane >> 1);
    if (s_raw - (lane >> 1) >= N) return;                        // a whole wave past the end (the waves never synchronise)
    const bool live = s_raw < N;
    const size_t s = live ? s_raw : N - 1;                       // idle pairs shadow the last signer and store nothing
    const uint32_t *row = reinterpret_cast<const uint32_t *>(text + s * text_stride) + half;
    const int nb = nblocks[s];
    KState S = {};
    // the next block's 17 words are requested before this block's permutation: an absorbed block is otherwise one memory
    // round trip (the rows are 7 KB apart: nothing coalesces) in front of every one of the ~47 permutations
    uint32_t m0, m1, m2, m3, m4, m5, m6, m7, m8, m9, m10, m11, m12, m13, m14, m15, m16;
#define FZ_LD(i) m##i = row[2 * i];
    FZ_FOR17(FZ_LD)
#undef FZ_LD
#pragma unroll 1
    for (int b = 0; b < max_blocks; ++b) {
        if (b < nb) {                                            // both lanes of a pair agree; pairs of a wave may differ by a block
#define FZ_ABS(i) S.a##i ^= m##i;
            FZ_FOR17(FZ_ABS)
#undef FZ_ABS
            const uint32_t *w = row + (b + 1 < nb ? b + 1 : b) * (kRate / 4);
#define FZ_LD(i) m##i = w[2 * i];
            FZ_FOR17(FZ_LD)
#undef FZ_LD
            keccak_f_half(S, half);
        }
    }
    uint32_t *out = xof + s + (size_t)half * xstride;
#pragma unroll 1
    for (int m = 0; m < out_blocks; ++m) {
        if (live) {
            uint32_t *o = out + (size_t)m * 34 * xstride;
#define FZ_SQ(i) o[(size_t)(2 * i) * xstride] = S.a##i;
            FZ_FOR17(FZ_SQ)
#undef FZ_SQ
        }
        if (m + 1 < out_blocks) keccak_f_half(S, half);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 2b. Keccak-f[1600] with a WHOLE state per lane (both 32-bit halves of every 64-bit lane in registers), for batches that
// fill the chip: a 64-bit rotation is two v_alignbit of the lane's own halves -- no partner fetch -- so a state costs
// ~180 vector operations per round instead of 2 x 119.  A single state takes 1.5x as long as on a lane pair, so this form
// pays only when the pair form would need more than one wave per SIMD (> 32 768 signers per pass): 65 536 signers are
// 1024 waves, one per SIMD.
// ---------------------------------------------------------------------------------------------------------------
struct W2 {
    uint32_t l, h;
};
__device__ __forceinline__ W2 w2_x3(W2 a, W2 b, W2 c) { return W2{FZ_X3(a.l, b.l, c.l), FZ_X3(a.h, b.h, c.h)}; }
__device__ __forceinline__ W2 w2_chi(W2 a, W2 b, W2 c) { return W2{a.l ^ (~b.l & c.l), a.h ^ (~b.h & c.h)}; }
template <int R>
__device__ __forceinline__ W2 w2_rotl(W2 x) {
    if (R == 0) return x;
    if (R == 32) return W2{x.h, x.l};
    if (R < 32) return W2{__builtin_amdgcn_alignbit(x.l, x.h, 32 - R), __builtin_amdgcn_alignbit(x.h, x.l, 32 - R)};
    return W2{__builtin_amdgcn_alignbit(x.h, x.l, 64 - R), __builtin_amdgcn_alignbit(x.l, x.h, 64 - R)};
}
// one output row y' of rho + pi + chi: B[x'] = rot(A[x][y] ^ D[x]) for the five (x, y) that land in row y' (x' = y,
// y' = 2x + 3y), the same (index, rotation) table as FZ_KROUND32 above
#define FZ_ROW64(E, O, I0, R0, I1, R1, I2, R2, I3, R3, I4, R4) \
    { const W2 b0 = w2_rotl<R0>(w2_x3(A[I0], C[(I0 + 4) % 5], Q[(I0 + 1) % 5])), b1 = w2_rotl<R1>(w2_x3(A[I1], C[(I1 + 4) % 5], Q[(I1 + 1) % 5])), \
               b2 = w2_rotl<R2>(w2_x3(A[I2], C[(I2 + 4) % 5], Q[(I2 + 1) % 5])), b3 = w2_rotl<R3>(w2_x3(A[I3], C[(I3 + 4) % 5], Q[(I3 + 1) % 5])), \
               b4 = w2_rotl<R4>(w2_x3(A[I4], C[(I4 + 4) % 5], Q[(I4 + 1) % 5])); \
      E[O] = w2_chi(b0, b1, b2); E[O + 1] = w2_chi(b1, b2, b3); E[O + 2] = w2_chi(b2, b3, b4); E[O + 3] = w2_chi(b3, b4, b0); \
      E[O + 4] = w2_chi(b4, b0, b1); }
__device__ __forceinline__ void keccak_round_full(const W2 (&A)[25], W2 (&E)[25], uint32_t rcl, uint32_t rch) {
    W2 C[5], Q[5];
#pragma unroll
    for (int x = 0; x < 5; ++x) C[x] = w2_x3(w2_x3(A[x], A[x + 5], A[x + 10]), A[x + 15], A[x + 20]);
#pragma unroll
    for (int x = 0; x < 5; ++x) Q[x] = w2_rotl<1>(C[x]);
    FZ_ROW64(E, 0, 0, 0, 6, 44, 12, 43, 18, 21, 24, 14)
    FZ_ROW64(E, 5, 3, 28, 9, 20, 10, 3, 16, 45, 22, 61)
    FZ_ROW64(E, 10, 1, 1, 7, 6, 13, 25, 19, 8, 20, 18)
    FZ_ROW64(E, 15, 4, 27, 5, 36, 11, 10, 17, 15, 23, 56)
    FZ_ROW64(E, 20, 2, 62, 8, 55, 14, 39, 15, 41, 21, 2)
    E[0].l ^= rcl;
    E[0].h ^= rch;
}
__device__ __forceinline__ void keccak_f_full(W2 (&A)[25]) {
    W2 E[25];
    uint32_t lo0 = kRC[0][0], hi0 = kRC[0][1], lo1 = kRC[1][0], hi1 = kRC[1][1];
#pragma unroll 1
    for (int r = 0; r < 24; r += 2) {
        const uint32_t a0 = lo0, b0 = hi0, a1 = lo1, b1 = hi1;
        const int rn = (r + 2) % 24;
        lo0 = kRC[rn][0]; hi0 = kRC[rn][1]; lo1 = kRC[rn + 1][0]; hi1 = kRC[rn + 1][1];
        keccak_round_full(A, E, a0, b0);
        keccak_round_full(E, A, a1, b1);
    }
}

// the same contract as shake_kernel, one lane per signer
template <int kShakeWaves>
__global__ __launch_bounds__(64 * kShakeWaves) void shake_full_kernel(const uint8_t *text, size_t text_stride, const int *nblocks,
                                                                      size_t N, int max_blocks, int out_blocks, uint32_t *xof,
                                                                      size_t xstride) {
    const int lane = threadIdx.x & 63;
    const size_t s_raw = ((size_t)blockIdx.x * kShakeWaves + (threadIdx.x >> 6)) * 64 + lane;
    if (s_raw - lane >= N) return;                               // a whole wave past the end (the waves never synchronise)
    const bool live = s_raw < N;
    const size_t s = live ? s_raw : N - 1;                       // idle lanes shadow the last signer and store nothing
    const uint2 *row = reinterpret_cast<const uint2 *>(text + s * text_stride);
    const int nb = nblocks[s];
    W2 A[25];
#pragma unroll
    for (int i = 0; i < 25; ++i) A[i] = W2{0u, 0u};
    uint2 m[17];
#pragma unroll
    for (int i = 0; i < 17; ++i) m[i] = row[i];
#pragma unroll 1
    for (int b = 0; b < max_blocks; ++b) {
        if (b < nb) {
#pragma unroll
            for (int i = 0; i < 17; ++i) { A[i].l ^= m[i].x; A[i].h ^= m[i].y; }
            const uint2 *w = row + (size_t)(b + 1 < nb ? b + 1 : b) * (kRate / 8);
#pragma unroll
            for (int i = 0; i < 17; ++i) m[i] = w[i];
            keccak_f_full(A);
        }
    }
    uint32_t *out = xof + s;
#pragma unroll 1
    for (int q = 0; q < out_blocks; ++q) {
        if (live) {
            uint32_t *o = out + (size_t)q * 34 * xstride;
#pragma unroll
            for (int i = 0; i < 17; ++i) {
                o[(size_t)(2 * i) * xstride] = A[i].l;
                o[(size_t)(2 * i + 1) * xstride] = A[i].h;
            }
        }
        if (q + 1 < out_blocks) keccak_f_full(A);
    }
}

// hash_message_to_int (fusion.py:405-409): SHA3-256 of dst + "," + message, one lane pair per message as above.  The
// stream is read byte by byte where it is absorbed -- messages are short (one block below 133 bytes) and start at any
// byte offset; its padding (0x06 ... 0x80) is part of the same byte function.  pre [N][32]: the digests, i.e. the
// pre-hashed integers, little-endian, exactly what vk_text_kernel prints in decimal.
__global__ __launch_bounds__(64) void prehash_kernel(const uint8_t *msgs, const unsigned long long *off, size_t N, uint32_t dst0,
                                                     uint32_t dst1, uint8_t *pre, uint32_t *dec) {
    const int lane = threadIdx.x & 63, half = lane & 1;
    const size_t s_raw = (size_t)blockIdx.x * 32 + (lane >> 1);
    const bool live = s_raw < N;
    const size_t s = live ? s_raw : N - 1;                       // idle pairs shadow the last message and store nothing
    const unsigned long long len = off[s + 1] - off[s];
    const uint8_t *m = msgs + (off[s] - off[0]);
    const unsigned long long nb = (len + 4 + kRate - 1) / kRate;        // 3 prefix bytes + message + the suffix byte
    const unsigned long long last = nb * kRate - 1;
    unsigned long long nbmax = nb;                               // the wave's loop bound
#pragma unroll
    for (int w = 2; w < 64; w <<= 1) {
        const unsigned long long o = __shfl_xor(nbmax, w);
        nbmax = o > nbmax ? o : nbmax;
    }
    auto byte_at = [&](unsigned long long p) -> uint32_t {
        uint32_t v;
        if (p < 3) {
            v = p == 0 ? dst0 : (p == 1 ? dst1 : 0x2cu);
        } else {
            const unsigned long long k = p - 3;
            v = k < len ? (uint32_t)m[k] : (k == len ? 0x06u : 0u);
        }
        return p == last ? (v | 0x80u) : v;
    };
    KState S = {};
#pragma unroll 1
    for (unsigned long long b = 0; b < nbmax; ++b) {
        if (b < nb) {                                            // both lanes of a pair agree
            const unsigned long long base = b * kRate + 4 * half;
#define FZ_ABSB(i) { const unsigned long long q_ = base + 8 * i; \
                     S.a##i ^= byte_at(q_) | (byte_at(q_ + 1) << 8) | (byte_at(q_ + 2) << 16) | (byte_at(q_ + 3) << 24); }
            FZ_FOR17(FZ_ABSB)
#undef FZ_ABSB
            keccak_f_half(S, half);
        }
    }
    if (live) {
        uint32_t *o = reinterpret_cast<uint32_t *>(pre + s * 32) + half;
        o[0] = S.a0; o[2] = S.a1; o[4] = S.a2; o[6] = S.a3;
    }
    // the integer's decimal chunks for vk_text_kernel, by the even lane of the pair (limb 2k = its own half of lane k, limb
    // 2k + 1 = the partner's)
    const uint32_t p0 = partner(S.a0), p1 = partner(S.a1), p2 = partner(S.a2), p3 = partner(S.a3);
    if (dec && live && half == 0) {
        uint32_t limb[8] = {S.a0, p0, S.a1, p1, S.a2, p2, S.a3, p3};
        uint32_t *d = dec + s * kDecStride;
        d[9] = (uint32_t)u256_to_base1e9(limb, d);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 3. decoder (fusion.py:422-481) for norm bound 1: signs, then the partial Fisher-Yates shuffle
// ---------------------------------------------------------------------------------------------------------------
struct DecodeShapeDev {
    int degree, weight, sign_bytes, coef_bytes, index_bytes;
};

// stream byte `pos` (the same for every lane) of this lane's signer
__device__ __forceinline__ uint32_t stream_byte(const uint32_t *x, size_t xstride, int pos) {
    return (x[(size_t)(pos >> 2) * xstride] >> (8 * (pos & 3))) & 0xffu;
}

constexpr int kChunkWords = 12;              // 32-bit stream words that can hold one index chunk at any alignment (<= 44 bytes)
constexpr int kTabStride = 16;               // uint32 per modulus in the weight table: 12 packed weight words, the reciprocal

// int.from_bytes(chunk, "big") % m for m <= 256 is a DOT PRODUCT: sum_k byte_k * (256^(ib-1-k) mod m) < 44 * 255^2 < 2^22,
// reduced once.  The weights of modulus m sit packed four to a word in tab[m] in the chunk's own byte order, so every
// four stream bytes cost one v_dot4_u32_u8; tab[m][12] = ceil(2^32 / m) gives the exact quotient of the final
// reduction by one multiply-high (sum * (ceil(2^32/m) * m - 2^32) < 2^22 * 2^8 < 2^32).
//
// Two phases.  (1) The shuffle INDICES j_n do not depend on the shuffle state, so all of them are computed first, as
// independent iterations whose loads are requested three iterations ahead (three chunk buffers with fixed roles: a
// register that a load is still writing is never moved; every load unconditional, indices clamped; the table rows come
// through VECTOR loads of a lane-invariant address, because scalar loads share one out-of-order counter with LDS and
// cannot be waited for individually), and parked in LDS as bytes.  (2) The swaps, the only sequential part: three
// dependent LDS accesses each.  Versions of this kernel, per launch: nine dependent fp64 steps per chunk with loads where
// they were used 582 us; loads ahead 185; dot products 131; branch-free loads 91; two phases with cheap addressing ~70.
template <int NW>       // stream words per chunk: ceil((3 + index_bytes) / 4)
__global__ __launch_bounds__(64) void decode_kernel(const uint32_t *xof, size_t xstride, size_t N, DecodeShapeDev D,
                                                    const uint32_t *tab, int kmax, int32_t *coefs) {
    extern __shared__ __attribute__((aligned(16))) signed char lds_dec[];
    const int lane = threadIdx.x & 63;
    const size_t s_raw = (size_t)blockIdx.x * 64 + lane;
    const bool live = s_raw < N;
    const uint32_t *x = xof + (live ? s_raw : N - 1);
    const int d = D.degree;
    signed char *out = lds_dec;                                  // [degree][64]: coefficients are -1, 0, 1 (norm bound 1)
    unsigned char *jl = reinterpret_cast<unsigned char *>(lds_dec) + (size_t)d * 64;      // [draws][64]: shuffle indices
    const int draws = d - 1 - D.weight > 0 ? d - 1 - D.weight : 0;
    for (int j = D.weight; j < d; ++j) out[j * 64 + lane] = 0;
    // sign i = bit i (LSB first) of the big-endian integer in the leading sign_bytes bytes (fusion.py:447-453);
    // magnitudes are 1 + (chunk mod 1) = 1.  Up to 64 signs: the integer is two byte-swapped words, loaded once.
    if (D.sign_bytes <= 8) {
        const uint32_t w0 = x[0], w1 = x[(size_t)(1 < kmax ? 1 : kmax) * xstride];
        unsigned long long v = ((unsigned long long)__builtin_bswap32(w0) << 32) | __builtin_bswap32(w1);    // bytes 0..7, big-endian
        v >>= 8 * (8 - D.sign_bytes);
        for (int i = 0; i < D.weight; ++i)
            if (i < d) out[i * 64 + lane] = ((v >> i) & 1ull) ? 1 : -1;
    } else {
        for (int i = 0; i < D.weight; ++i) {
            const uint32_t byte = stream_byte(x, xstride, D.sign_bytes - 1 - (i >> 3));
            if (i < d) out[i * 64 + lane] = ((byte >> (i & 7)) & 1) ? 1 : -1;
        }
    }
    // ---- phase 1: j_n = int.from_bytes(chunk_n, "big") % (d - n) for n = 0 .. draws-1   (fusion.py:472-480) ----
    const int ib = D.index_bytes, pos = D.sign_bytes + D.coef_bytes * D.weight;
    unsigned vzero = 0;
    asm volatile("" : "+v"(vzero));                              // an opaque zero in a VGPR: tab + vzero is still a GLOBAL pointer
    const uint32_t *tabv = tab + vzero;                          // (a laundered pointer would become flat loads, which force
                                                                 // vmcnt(0) waits), but its loads are vector loads
    struct Buf { uint32_t w[NW]; uint4 t0, t1, t2; uint32_t t12; };
    auto fetch = [&](Buf &b, int n) {                            // chunk n's words and its modulus' table row
        // past the last chunk the last one is fetched again (never used): every word index is then valid without a
        // per-word clamp (the stream buffer has one spare word row), and a chunk's NW words are ONE 64-bit base plus
        // loop-invariant multiples of the row stride -- per-word 64-bit index arithmetic was 2/3 of this loop's instructions
        const int nn = n < draws ? n : draws - 1;
        const uint32_t *pw = x + (size_t)((pos + nn * ib) >> 2) * xstride;
#pragma unroll
        for (int t = 0; t < NW; ++t) b.w[t] = pw[(size_t)t * xstride];
        const uint32_t *T = tabv + (size_t)(d - nn) * kTabStride;       // modulus i + 1 = d - n
        b.t0 = *reinterpret_cast<const uint4 *>(T);
        b.t1 = *reinterpret_cast<const uint4 *>(T + 4);
        b.t2 = *reinterpret_cast<const uint4 *>(T + 8);
        b.t12 = T[12];
    };
    auto step = [&](Buf &b, int n) {
        const int i = n < draws ? d - 1 - n : 0;
        const int sh = (pos + n * ib) & 3;
        const uint32_t tw[12] = {b.t0.x, b.t0.y, b.t0.z, b.t0.w, b.t1.x, b.t1.y, b.t1.z, b.t1.w, b.t2.x, b.t2.y, b.t2.z, b.t2.w};
        uint32_t sum = 0;
#pragma unroll
        for (int g = 0; g < NW; ++g) {
            // chunk bytes 4g .. 4g+3 in stream order (bytes past the chunk meet zero weights)
            const uint32_t hi = g + 1 < NW ? b.w[g + 1 < NW ? g + 1 : g] : 0u;
            sum = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(hi, b.w[g], sh), tw[g], sum, false);
        }
        const uint32_t j = sum - __umulhi(sum, b.t12) * (uint32_t)(i + 1);       // sum mod (i + 1), exact
        if (n < draws) jl[n * 64 + lane] = (unsigned char)j;
        fetch(b, n + 3);                                         // refill: three iterations ahead
    };
    Buf ba, bb, bc;
    fetch(ba, 0);
    fetch(bb, 1);
    fetch(bc, 2);
    for (int n = 0; n < draws; n += 3) {
        step(ba, n);
        step(bb, n + 1);
        step(bc, n + 2);
    }
    // ---- phase 2: for i = d-1 down to weight+1: swap(out[i], out[j]) ----
    for (int n = 0; n < draws; ++n) {
        const int i = d - 1 - n, j = jl[n * 64 + lane];
        const signed char vi = out[i * 64 + lane], vj = out[j * 64 + lane];
        out[i * 64 + lane] = vj;
        out[j * 64 + lane] = vi;
    }
    if (live) {
        int32_t *dst = coefs + s_raw * (size_t)d;
        for (int j = 0; j < d; j += 4)
            *reinterpret_cast<int4 *>(dst + j) = make_int4(out[j * 64 + lane], out[(j + 1) * 64 + lane], out[(j + 2) * 64 + lane],
                                                           out[(j + 3) * 64 + lane]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 4. the whole pipeline of ONE signer on ONE wave (fz_keccak_wave.h): pre-hash, text, absorb, squeeze, decode
// ---------------------------------------------------------------------------------------------------------------
// With a batch of BASELINE configs[2]'s size (1024 signers) the three kernels above run 32 (lane pairs) or 16 (decoder)
// waves on a chip with 1024 SIMDs and the call is the latency of one signer's chain.  Here a signer has a wave to itself:
//   * SHA3-256 of the message and SHAKE-256 of the key text run on the wave-wide Keccak state (24 instructions + 4 gathers
//     per round instead of 119 instructions): the ~108 permutations take ~0.25 ms instead of 0.64;
//   * the text is written to LDS and absorbed from there, the stream is squeezed into the same LDS bytes and decoded from
//     there: nothing but the key row, the message and the 4 d bytes of the challenge touches memory;
//   * the decoder's two phases are the wave's: the shuffle indices 64 at a time (a lane per draw, the byte dot product of
//     decode_kernel with the table row of the lane's own modulus), then the swaps -- not as swaps at all: the k-th non-zero
//     coefficient (weight <= 64) ends where the FIRST draw that names its original position sends it (see phase 2 below),
//     one LDS minimum per draw instead of 195 dependent steps.
// LDS per wave: max(text row, stream, 4 d) + 64 bytes.
constexpr int kWaveAux = 64;
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <int W, int NW>      // signers (waves) per workgroup; stream words per index chunk
__global__ __launch_bounds__(64 * W) void challenge_wave_kernel(const int32_t *vk, size_t vk_stride, const uint8_t *pre, const uint8_t *msgs,
                                                                const unsigned long long *off, uint8_t *pre_out, uint32_t dst0, uint32_t dst1,
                                                                size_t N, VkTextParts T, DecodeShapeDev D, const uint32_t *tab, int out_blocks,
                                                                size_t region, int32_t *coefs) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63;
    const size_t s = (size_t)blockIdx.x * W + (threadIdx.x >> 6);
    if (s >= N) return;                                          // waves of a workgroup never synchronise with each other
    uint8_t *buf = smem + (size_t)(threadIdx.x >> 6) * (region + kWaveAux);
    uint32_t *aux = reinterpret_cast<uint32_t *>(buf + region);
    VkTextIn tin;                                                // the key row and the fixed pieces: requested now, used after the message's
    vk_text_load(tin, vk + s * vk_stride, D.degree, T, lane);    // digest (a memory latency of ~2.5 us under the ~8 us before they are needed)
    fzkw::Wave K;
    K.init(lane);
    const bool ab = K.word < 17;                                 // the rate's 17 words (their owners and the halos)
    // ---- the pre-hashed message: SHA3-256(dst + "," + message) (fusion.py:405-409), or the caller's digest ----
    if (pre) {
        if (lane < 8) aux[lane] = reinterpret_cast<const uint32_t *>(pre + s * 32)[lane];
    } else {
        const unsigned long long len = off[s + 1] - off[s];
        const uint8_t *m = msgs + (off[s] - off[0]);
        const unsigned long long nb = (len + 4 + kRate - 1) / kRate, last = nb * kRate - 1;      // 3 prefix bytes + message + the suffix byte
        // a block's 136 bytes are put together in LDS by the whole wave -- byte p of the stream by lane p mod 64, so that the
        // message is read in three coalesced requests per block (it may sit in pinned HOST memory: fz_capi.hip) -- and taken
        // from there as the rate's 17 words
#pragma unroll 1
        for (unsigned long long b = 0; b < nb; ++b) {
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int idx = lane + 64 * t;
                if (idx < kRate) {
                    const unsigned long long p = b * kRate + (unsigned)idx;
                    uint32_t v;
                    if (p < 3) {
                        v = p == 0 ? dst0 : (p == 1 ? dst1 : 0x2cu);
                    } else {
                        const unsigned long long k = p - 3;
                        v = k < len ? (uint32_t)m[k] : (k == len ? 0x06u : 0u);
                    }
                    buf[idx] = (uint8_t)(p == last ? (v | 0x80u) : v);
                }
            }
            wave_sync();
            if (ab) {
                const uint2 w = *reinterpret_cast<const uint2 *>(buf + 8 * K.word);
                K.lo ^= w.x;
                K.hi ^= w.y;
            }
            wave_sync();
            K.permute();
        }
        if (K.main && K.word < 4) {
            aux[2 * K.word] = K.lo;
            aux[2 * K.word + 1] = K.hi;
            if (pre_out) *reinterpret_cast<uint2 *>(pre_out + s * 32 + 8 * K.word) = make_uint2(K.lo, K.hi);
        }
        K.lo = K.hi = 0u;
    }
    wave_sync();
    // str(int.from_bytes(digest, "little")): base 10^9 chunks.  The nine rounds of the eight-limb short division as a
    // systolic array: lane t owns limb t, the remainder travels from lane t + 1 to lane t (one DPP read), so lane t runs
    // round r at step r + 7 - t and the 72 dependent divisions of one lane become 16 steps of the wave (4.6 -> ~1 us).
    {
        uint32_t limb = lane < 8 ? aux[lane] : 0u, rem = 0u;
        wave_sync();                                             // every limb has been read before aux is rewritten
#pragma unroll 1
        for (int step = 0; step < 16; ++step) {
            const int r = step - (7 - lane);                     // this lane's round
            const uint32_t rin = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rem, 0x101, 0xf, 0xf, true);       // lane t + 1's remainder
            const unsigned long long cur = ((unsigned long long)rin << 32) | limb;
            const uint32_t q = (uint32_t)(cur / 1000000000ull);
            const bool on = lane < 8 && r >= 0 && r < 9;
            rem = on ? (uint32_t)(cur - (unsigned long long)q * 1000000000ull) : 0u;
            if (on) limb = q;
            if (on && lane == 0) aux[r] = rem;
        }
        wave_sync();
        const uint32_t mine = lane < 9 ? aux[lane] : 0u;
        const unsigned long long nz = __ballot(mine != 0u);
        if (lane == 0) aux[9] = nz ? (uint32_t)(64 - __builtin_clzll(nz)) : 1u;
    }
    wave_sync();
    // ---- the text, absorbed from LDS ----
    const int nb = vk_text_wave(buf, region, aux, tin, D.degree, T, lane);
    {
        const uint8_t *src = buf + 8 * (ab ? K.word : 0);
        uint2 m = ab ? *reinterpret_cast<const uint2 *>(src) : make_uint2(0u, 0u);
#pragma unroll 1
        for (int b = 0; b < nb; ++b) {
            K.lo ^= m.x;
            K.hi ^= m.y;
            if (ab) m = *reinterpret_cast<const uint2 *>(src + (size_t)(b + 1 < nb ? b + 1 : b) * kRate);        // the next block, under the permutation
            K.permute();
        }
    }
    wave_sync();
    // ---- the stream, squeezed into the same bytes ----
#pragma unroll 1
    for (int q = 0; q < out_blocks; ++q) {
        if (ab && K.main) *reinterpret_cast<uint2 *>(buf + (size_t)q * kRate + 8 * K.word) = make_uint2(K.lo, K.hi);
        if (q + 1 < out_blocks) K.permute();
    }
    wave_sync();
    // ---- decoder (fusion.py:422-481), norm bound 1 ----
    const int d = D.degree, wt = D.weight < d ? D.weight : d;
    const int draws = d - 1 - D.weight > 0 ? d - 1 - D.weight : 0;
    // sign k = bit k (LSB first) of the big-endian integer in the leading sign_bytes bytes; magnitudes are 1 + (chunk mod 1) = 1
    int sgn = 0, pos = lane;
    if (lane < wt) sgn = ((buf[D.sign_bytes - 1 - (lane >> 3)] >> (lane & 7)) & 1) ? 1 : -1;
    // phase 1: j_n = int.from_bytes(chunk_n, "big") % (d - n), a lane per draw
    const int ib = D.index_bytes, pos0 = D.sign_bytes + D.coef_bytes * D.weight;
    const uint32_t *xs = reinterpret_cast<const uint32_t *>(buf);
    uint32_t jv[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c * 64 < draws) {                                    // (uniform)
            const int n = c * 64 + lane, nn = n < draws ? n : draws - 1;
            const int o = pos0 + nn * ib, sh = o & 3;
            const uint32_t *pw = xs + (o >> 2);
            uint32_t w[NW];
#pragma unroll
            for (int t = 0; t < NW; ++t) w[t] = pw[t];
            const uint32_t mod = (uint32_t)(d - nn);
            const uint32_t *Tm = tab + (size_t)mod * kTabStride;
            const uint4 t0 = *reinterpret_cast<const uint4 *>(Tm), t1 = *reinterpret_cast<const uint4 *>(Tm + 4), t2 = *reinterpret_cast<const uint4 *>(Tm + 8);
            const uint32_t recip = Tm[12];
            const uint32_t tw[12] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w, t2.x, t2.y, t2.z, t2.w};
            uint32_t sum = 0;
#pragma unroll
            for (int g = 0; g < NW; ++g) {
                const uint32_t hi = g + 1 < NW ? w[g + 1 < NW ? g + 1 : g] : 0u;
                sum = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(hi, w[g], sh), tw[g], sum, false);
            }
            jv[c] = sum - __umulhi(sum, recip) * mod;            // sum mod (d - n), exact (fz_challenge_weight_table)
        }
    }
    // phase 2: for i = d-1 down to weight+1: swap(c[i], c[j]).  Before step i the positions weight .. i hold zeros (induction
    // from the top: a step writes position i only), so a step either exchanges two zeros or lifts the non-zero coefficient k = j
    // from its ORIGINAL position (< weight) to i, where no later step reaches it (their i and j are smaller) -- and leaves a
    // zero at k, so later draws of k do nothing.  The k-th coefficient therefore ends at the i of the FIRST draw with j = k, or
    // stays: 195 dependent swaps (6.2 us of the kernel: a compare pair and two selects each) become one LDS minimum per draw.
    wave_sync();                                                 // every lane has read its index chunks: the stream's first words become scratch
    uint32_t *first = reinterpret_cast<uint32_t *>(buf);         // first[k], k < weight: the smallest draw number n with j_n = k
    if (lane < wt) first[lane] = 0xffffffffu;                    // (4 * weight <= 4 * degree bytes: inside the region whatever the parameters)
    wave_sync();
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int n = c * 64 + lane;
        if (n < draws && jv[c] < (uint32_t)wt) atomicMin(&first[jv[c]], (uint32_t)n);
    }
    wave_sync();
    if (lane < wt) {
        const uint32_t n1 = first[lane];
        pos = n1 != 0xffffffffu ? d - 1 - (int)n1 : lane;
    }
    wave_sync();                                                 // the stream has been read: its bytes become the coefficient row
    int32_t *out = reinterpret_cast<int32_t *>(buf);
    for (int j = lane; j < d; j += 64) out[j] = 0;
    wave_sync();
    if (lane < wt) out[pos] = sgn;
    wave_sync();
    int32_t *dst = coefs + s * (size_t)d;
    for (int j = lane * 4; j < d; j += 256) *reinterpret_cast<int4 *>(dst + j) = *reinterpret_cast<const int4 *>(out + j);
}

}  // namespace

// the fixed pieces of str(OneTimeVerificationKey) come from the host serialiser, so both pipelines share one definition
void fz_host_vk_text_parts(const fz_scheme_params *P, char *s0, int *n0, char *s1, int *n1, char *s2, int *n2, int cap);
size_t fz_host_challenge_needed_bytes(const fz_scheme_params *P, int *sign_bytes, int *coef_bytes, int *index_bytes);

// d_vk [N][2][degree] int32 (left row, right row), d_pre [N][32] (SHA3-256 digests of the messages = the pre-hashed
// integers, little-endian), d_text / d_nblocks / d_xof scratch; d_coefs [N][degree] out
int fz_launch_challenge(fz_ctx *ctx, const fz_scheme_params *P, const int32_t *d_vk, const uint8_t *d_pre, const uint32_t *d_dec, size_t N,
                        uint8_t *d_text, size_t text_stride, int *d_nblocks, uint32_t *d_xof, size_t xstride, int out_blocks,
                        const uint32_t *d_tab, int32_t *d_coefs) {
    VkTextParts T;
    fz_host_vk_text_parts(P, T.s0, &T.n0, T.s1, &T.n1, T.s2, &T.n2, 384);
    if (T.n0 < 0) return fz_set_error(FZ_E_UNSUPPORTED, "verification-key text pieces do not fit");
    const int d = P->degree;
    const size_t lds_text = (size_t)kTextWaves * text_stride + kTextWaves * 64;
    if (lds_text > 160 * 1024) return fz_set_error(FZ_E_UNSUPPORTED, "text row too long for the serialiser");
    if (lds_text > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)vk_text_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_text);
        if (e != hipSuccess) return fz_check_hip(e, "text kernel LDS attribute");
    }
    hipLaunchKernelGGL(vk_text_kernel, dim3((unsigned)((N + kTextWaves - 1) / kTextWaves)), dim3(64 * kTextWaves), lds_text, ctx->stream,
                       d_vk, (size_t)2 * d, d_pre, d_dec, N, d, T, d_text, text_stride, d_nblocks);
    int rc = fz_check_hip(hipGetLastError(), "vk_text launch");
    if (rc != FZ_OK) return rc;
    const int max_blocks = (int)(text_stride / kRate);
    // lane pairs while they fit one wave per SIMD (the latency-optimal form), whole states per lane beyond
    const int shake_mode = ctx->knob_shake_full ? ctx->knob_shake_full : ((N + 31) / 32 > (size_t)ctx->num_cu * 4 ? 2 : 1);
    const size_t per_wave = shake_mode == 2 ? 64 : 32, waves = (N + per_wave - 1) / per_wave;
#define FZ_SHK(KERNEL, W) hipLaunchKernelGGL((KERNEL<W>), dim3((unsigned)((waves + W - 1) / W)), dim3(64 * W), 0, ctx->stream, d_text, \
                                             text_stride, d_nblocks, N, max_blocks, out_blocks, d_xof, xstride)
    if (shake_mode == 2) { if (waves > (size_t)ctx->num_cu) FZ_SHK(shake_full_kernel, 4); else FZ_SHK(shake_full_kernel, 1); }
    else { if (waves > (size_t)ctx->num_cu) FZ_SHK(shake_kernel, 4); else FZ_SHK(shake_kernel, 1); }
#undef FZ_SHK
    rc = fz_check_hip(hipGetLastError(), "shake launch");
    if (rc != FZ_OK) return rc;
    DecodeShapeDev D;
    D.degree = d;
    D.weight = P->omega_ch;
    (void)fz_host_challenge_needed_bytes(P, &D.sign_bytes, &D.coef_bytes, &D.index_bytes);
    if (D.index_bytes > 4 * (kChunkWords - 1)) return fz_set_error(FZ_E_UNSUPPORTED, "index chunks of %d bytes", D.index_bytes);
    const int nw = (3 + D.index_bytes + 3) / 4, kmax = out_blocks * 34 - 1;
    const dim3 dgrid((unsigned)((N + 63) / 64));
    const size_t dlds = (size_t)d * 64 + (size_t)(d > D.weight ? d - 1 - D.weight : 0) * 64 + 64;
    if (nw <= 5) hipLaunchKernelGGL(decode_kernel<5>, dgrid, dim3(64), dlds, ctx->stream, d_xof, xstride, N, D, d_tab, kmax, d_coefs);
    else if (nw <= 9) hipLaunchKernelGGL(decode_kernel<9>, dgrid, dim3(64), dlds, ctx->stream, d_xof, xstride, N, D, d_tab, kmax, d_coefs);
    else hipLaunchKernelGGL(decode_kernel<12>, dgrid, dim3(64), dlds, ctx->stream, d_xof, xstride, N, D, d_tab, kmax, d_coefs);
    return fz_check_hip(hipGetLastError(), "decode launch");
}

// the fused form: one wave per signer.  d_pre [N][32] (digests) or, when null, d_msgs / d_off (the messages back to back and
// their N + 1 offsets) with d_pre_out [N][32] (optional) receiving the digests; d_coefs [N][degree] out
bool fz_challenge_wave_ok(const fz_scheme_params *P) { return P->omega_ch <= 64 && P->degree <= 256 && P->degree >= 4; }

int fz_launch_challenge_wave(fz_ctx *ctx, const fz_scheme_params *P, const int32_t *d_vk, const uint8_t *d_pre, const uint8_t *d_msgs,
                             const unsigned long long *d_off, uint8_t *d_pre_out, size_t N, size_t text_stride, int out_blocks,
                             const uint32_t *d_tab, int32_t *d_coefs) {
    if (N == 0) return FZ_OK;
    VkTextParts T;
    fz_host_vk_text_parts(P, T.s0, &T.n0, T.s1, &T.n1, T.s2, &T.n2, 384);
    if (T.n0 < 0) return fz_set_error(FZ_E_UNSUPPORTED, "verification-key text pieces do not fit");
    DecodeShapeDev D;
    D.degree = P->degree;
    D.weight = P->omega_ch;
    (void)fz_host_challenge_needed_bytes(P, &D.sign_bytes, &D.coef_bytes, &D.index_bytes);
    if (D.index_bytes > 4 * (kChunkWords - 1)) return fz_set_error(FZ_E_UNSUPPORTED, "index chunks of %d bytes", D.index_bytes);
    if (!fz_challenge_wave_ok(P)) return fz_set_error(FZ_E_UNSUPPORTED, "wave form: weight <= 64 and degree 4..256 only");
    const int nw = (3 + D.index_bytes + 3) / 4;
    // one region for the text row, then the stream (a chunk's NW words may end up to 12 bytes past it), then the coefficients
    size_t region = text_stride;
    region = std::max(region, (size_t)out_blocks * kRate + 16);
    region = std::max(region, (size_t)P->degree * 4);
    region = (region + 15) & ~(size_t)15;
    constexpr int W = 4;
    const size_t lds = (size_t)W * (region + kWaveAux);
    if (lds > 160 * 1024) return fz_set_error(FZ_E_UNSUPPORTED, "text row too long for the fused challenge kernel");
    const dim3 grid((unsigned)((N + W - 1) / W)), block(64 * W);
#define FZ_CW(NWV) do { \
        if (lds > 64 * 1024) { \
            hipError_t e = hipFuncSetAttribute((const void *)challenge_wave_kernel<W, NWV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return fz_check_hip(e, "challenge kernel LDS attribute"); \
        } \
        hipLaunchKernelGGL((challenge_wave_kernel<W, NWV>), grid, block, lds, ctx->stream, d_vk, (size_t)2 * P->degree, d_pre, d_msgs, d_off, d_pre_out, \
                           (uint32_t)P->sign_pre_hash_dst[0], (uint32_t)P->sign_pre_hash_dst[1], N, T, D, d_tab, out_blocks, region, d_coefs); \
    } while (0)
    if (nw <= 5) FZ_CW(5);
    else if (nw <= 9) FZ_CW(9);
    else FZ_CW(12);
#undef FZ_CW
    return fz_check_hip(hipGetLastError(), "challenge (wave form) launch");
}

// SHA3-256 of dst + "," + message for N messages: d_msgs the message bytes back to back, d_off [N + 1] their offsets (any
// origin: off[0] is subtracted), d_pre [N][32] out
int fz_launch_prehash(fz_ctx *ctx, const fz_scheme_params *P, const uint8_t *d_msgs, const unsigned long long *d_off, size_t N,
                      uint8_t *d_pre, uint32_t *d_dec) {
    if (N == 0) return FZ_OK;
    hipLaunchKernelGGL(prehash_kernel, dim3((unsigned)((N + 31) / 32)), dim3(64), 0, ctx->stream, d_msgs, d_off, N,
                       (uint32_t)P->sign_pre_hash_dst[0], (uint32_t)P->sign_pre_hash_dst[1], d_pre, d_dec);
    return fz_check_hip(hipGetLastError(), "prehash launch");
}

// the decoder's weight table for (index_bytes, degree): tab[m][g] (g < 12) packs 256^(ib-1-(4g+t)) mod m for t = 0..3 (zero
// past the chunk), tab[m][12] = ceil(2^32 / m); m = 1 .. degree.  Host array of (degree + 1) * 16 words.
void fz_challenge_weight_table(int index_bytes, int degree, uint32_t *h_tab) {
    for (int m = 0; m <= degree; ++m) {
        uint32_t *T = h_tab + (size_t)m * kTabStride;
        for (int g = 0; g < kTabStride; ++g) T[g] = 0;
        if (m == 0) continue;
        uint32_t pw = 1u % (uint32_t)m;                          // 256^0 mod m, then upwards from the chunk's LAST byte
        for (int k = index_bytes - 1; k >= 0; --k) {
            T[k >> 2] |= pw << (8 * (k & 3));
            pw = (pw * 256u) % (uint32_t)m;
        }
        T[12] = (uint32_t)((0x100000000ull + (unsigned)m - 1) / (unsigned)m);      // ceil(2^32 / m); m = 1: 0 (2^32 wraps) ...
        if (m == 1) T[12] = 0xffffffffu;                         // ... any sum mod 1 = 0: with q = sum - 1 the weights are all 0 anyway
    }
}
