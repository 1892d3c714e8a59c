// fz_keccak_wave.h -- Keccak-f[1600] with ONE state spread over a 64-lane wave (device only, gfx950).
//
// The challenge pipeline of one signer (fusion/fusion.py:412-419, :511-531) is a chain of ~108 dependent permutations;
// with few signers per call (BASELINE configs[2]: 1024) the chain's latency is the call's latency.  A lone wave issues one
// vector instruction per ~4 cycles whatever the number of active lanes, so the time of a round is the number of
// instructions the wave executes for it and the length of their dependency chain: 119 on the lane-pair form of
// fz_challenge.hip (25 lanes of the state per lane), 24 + 4 cross-lane gathers here, in a chain of 14 dependent steps.
//
// Layout.  Keccak lane (x, y), both 32-bit halves (lo, hi), lives in wave lane 16 r + 8 h + p with y = 2 r + h and
// p = x + 1: a plane (fixed y) is a group of eight consecutive lanes, two planes share a DPP row, planes 0..4 take rows 0..2.
// Positions p = 0 and p = 6 are HALOS -- copies of x = 4 and x = 0 -- so that theta's x - 1 / x + 1 are plain DPP row shifts
// (row_shr:1 / row_shl:1) without a wrap-around; p = 7, the second group of row 2 and all of row 3 are idle and hold zero
// where theta reads them.
//   theta  column parities: an XOR all-reduce over y = over (h, r): h by one DPP row_ror:8 per half; the two row bits by
//          v_permlane16_swap / v_permlane32_swap, both halves travelling in ONE register between the steps (even rows carry
//          lo, odd rows hi): 8 instructions instead of 14.  D is never formed: a ^= C[x-1] (row_shr:1) ^ rot(C[x+1], 1) (row_shl:1).
//   rho    every lane rotates its own 64-bit word by its own amount: v_alignbit with a per-lane shift, halves swapped first
//          where the amount is >= 32.
//   pi+chi ONE level of ds_bpermute gathers: lane (x', y') fetches B[x'], B[x'+1] of its row straight from the lanes that
//          hold them before pi, B[x'+2] is the right neighbour's B[x'+1] (DPP row_shl:1; p = 7 gathers for it); halos
//          fetch what their twins fetch, so all seven positions are valid again.  chi is one v_bitop3 per half.
//   iota   deferred into the next round's theta as a per-lane constant of its three-input XOR (struct Wave): no instruction.
// Idle lanes stay zero where theta reads them without a masking instruction: row 3 is never written with anything but
// zero (theta's DPP reads carry row_mask 0x7) and every other idle lane gathers from it.
// Everything is checked bit for bit against hashlib (tests/test_gpu_challenge.py, tools/microbench/keccak_wave.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fzkw {

__device__ static const uint32_t kRoundConst[24][2] = {
    {0x00000001u, 0x00000000u}, {0x00008082u, 0x00000000u}, {0x0000808au, 0x80000000u}, {0x80008000u, 0x80000000u},
    {0x0000808bu, 0x00000000u}, {0x80000001u, 0x00000000u}, {0x80008081u, 0x80000000u}, {0x00008009u, 0x80000000u},
    {0x0000008au, 0x00000000u}, {0x00000088u, 0x00000000u}, {0x80008009u, 0x00000000u}, {0x8000000au, 0x00000000u},
    {0x8000808bu, 0x00000000u}, {0x0000008bu, 0x80000000u}, {0x00008089u, 0x80000000u}, {0x00008003u, 0x80000000u},
    {0x00008002u, 0x80000000u}, {0x00000080u, 0x80000000u}, {0x0000800au, 0x00000000u}, {0x8000000au, 0x80000000u},
    {0x80008081u, 0x80000000u}, {0x00008080u, 0x80000000u}, {0x80000001u, 0x00000000u}, {0x80008008u, 0x80000000u}};

// rotation offsets, index x + 5 y
__device__ static const unsigned char kRho[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};

__host__ __device__ constexpr int lane_of(int x, int y) { return 16 * (y >> 1) + 8 * (y & 1) + x + 1; }

// rot64 by one of a round constant, as (lo, hi)
__host__ __device__ constexpr uint32_t rot1_lo(uint32_t lo, uint32_t hi) { return (lo << 1) | (hi >> 31); }
__host__ __device__ constexpr uint32_t rot1_hi(uint32_t lo, uint32_t hi) { return (hi << 1) | (lo >> 31); }

struct Wave {
    // per-lane constants
    uint32_t idx0, idx1;             // byte addresses (4 * source lane) of the two gathers
    uint32_t shift;                  // v_alignbit amount: (32 - n mod 32) mod 32
    bool swap;                       // rotation amount >= 32 (or 0: see rho below): halves exchanged before the funnel shift
    bool main;                       // p in 1..5: the lane that owns (x, y)
    bool origin;                     // holds (0, 0) (the owner and its halo)
    int word;                        // x + 5 y (0..24) on state lanes, 25 on idle lanes
    // iota is DEFERRED into the next round's theta: with e = the state without the last round's constant rc, the parities
    // computed from e miss rc in C[0] only, which theta uses in column 1 (as C[x-1]) and column 4 (as rot(C[x+1], 1)); so the
    // lanes of (0, 0), of column 1 and of column 4 XOR a per-lane constant -- rc, rc, rot(rc, 1) -- into the same three-input
    // XOR that applies D, and the constant costs no instruction and sits on no dependency chain.  kl/kh[i] belong to round i + 1;
    // the last round's constant is applied by permute() itself.
    uint32_t kl[23], kh[23];
    // the state
    uint32_t lo, hi;

    __device__ __forceinline__ void init(int lane) {
        const int r = lane >> 4, h = (lane >> 3) & 1, p = lane & 7, y = 2 * r + h;
        const bool state = y <= 4 && p <= 6;
        const int x = (p + 4) % 5;
        main = state && p >= 1 && p <= 5;
        word = state ? x + 5 * y : 25;
        const int n = state ? kRho[x + 5 * y] : 1;
        // n = 0 (only (0, 0)): "swapped, shift 0" is the identity, because v_alignbit by 0 returns its SECOND operand
        swap = state && (n >= 32 || n == 0);
        shift = (32u - ((unsigned)n & 31u)) & 31u;
        // Idle lanes must read as zero where theta's parities take them in (the second group of row 2, row 3).  Row 3 never
        // becomes non-zero: theta's two DPP reads are masked off there (row_mask 0x7), everything else maps zero to zero, and
        // its lanes gather from themselves.  The other idle lanes gather from lane 63, i.e. zero, so chi leaves zero in them.
        const uint32_t zero_src = r == 3 ? 4u * (uint32_t)lane : 4u * 63u;
        uint32_t id[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int X = (x + k) % 5, Y = y;               // B[X][Y] = rot(A[x0][y0]) with x0 = X + 3 Y, y0 = X   (pi inverted)
            id[k] = state ? 4u * (uint32_t)lane_of((X + 3 * Y) % 5, X) : zero_src;
        }
        idx0 = id[0]; idx1 = id[1];
        if (y <= 4 && p == 7) idx1 = 4u * (uint32_t)lane_of((x + 1 + 3 * y) % 5, (x + 1) % 5);      // p = 7 stands in for x = 1: its b1 is p = 6's b2
        origin = state && x == 0 && y == 0;
        const bool c1 = state && x == 1, c4 = state && x == 4;
#pragma unroll
        for (int i = 0; i < 23; ++i) {
            const uint32_t l = kRoundConst[i][0], hh = kRoundConst[i][1];
            kl[i] = (origin || c1) ? l : (c4 ? rot1_lo(l, hh) : 0u);
            kh[i] = (origin || c1) ? hh : (c4 ? rot1_hi(l, hh) : 0u);
        }
        lo = hi = 0u;
    }

    __device__ __forceinline__ static uint32_t dpp_ror8(uint32_t v) {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);
    }
    // rows 0..2 only (row_mask 0x7): row 3 receives `old` = 0
    __device__ __forceinline__ static uint32_t dpp_shr1(uint32_t v) {      // lane i reads lane i - 1 of its row (0 at the row's start)
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0x7, 0xf, true);
    }
    __device__ __forceinline__ static uint32_t dpp_shl1(uint32_t v) {      // lane i reads lane i + 1 of its row (0 at the row's end)
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0x7, 0xf, true);
    }

    // one round without its iota; k_lo / k_hi: the previous round's deferred constant (0 in round 0)
    __device__ __forceinline__ void round(uint32_t k_lo, uint32_t k_hi) {
        // the state with the deferred constant: theta's target.  Computed first, so that the swaps below may consume lo / hi
        uint32_t xl = lo ^ k_lo, xh = hi ^ k_hi;
        asm volatile("" : "+v"(xl), "+v"(xh));                                        // (computed HERE, not re-associated into the XORs below)
        // theta: column parities in every lane of the column -- an XOR all-reduce over y = over (row pair, row, group).  Both
        // halves travel in ONE register after the first swap: even rows carry lo, odd rows hi.  A swap consumes two registers:
        // where both must hold the same value it is computed twice (two independent instructions) rather than copied (an
        // instruction that waits for the first); the second forms differ only in spelling, to keep the compiler from merging them.
        // (Each half reduced on its own -- row_ror:8, v_permlane16_swap and v_permlane32_swap of a value with its twin, no step
        // to take the halves apart again: five dependent steps instead of six, 30 instructions per round instead of 24 --
        // measured SLOWER, 256 / 275 us against 247 / 266: a lone wave pays for every instruction, not only for the chain.)
        const auto s1 = __builtin_amdgcn_permlane16_swap(lo, hi, false, false);       // [lo r0, hi r0, lo r2, hi r2] / [lo r1, hi r1, lo r3, hi r3]
        const uint32_t z0 = s1[0] ^ s1[1];                                            // rows r ^ (r xor 1)
        const uint32_t za = z0 ^ (uint32_t)__builtin_amdgcn_update_dpp(0, (int)z0, 0x128, 0xf, 0xf, false);       // the row's two groups
        const uint32_t zb = z0 ^ (uint32_t)__builtin_amdgcn_update_dpp(0, (int)z0, 0x128, 0xf, 0xf, true);
        const auto s2 = __builtin_amdgcn_permlane32_swap(za, zb, false, false);       // [z.L, z.L] / [z.H, z.H]
        const uint32_t wa = s2[0] ^ s2[1];                                            // even rows: C lo, odd rows: C hi
        const uint32_t wb = (uint32_t)__builtin_amdgcn_bitop3_b32(s2[0], s2[1], s2[1], 0x3C);
        const auto s3 = __builtin_amdgcn_permlane16_swap(wa, wb, false, false);
        const uint32_t cl = s3[0], ch = s3[1];
        // a ^= C[x-1] ^ rot(C[x+1], 1): two DPP reads folded into two XORs, in place (row 3 is masked off and keeps its zero);
        // the first does not wait for the rotation
        xl ^= dpp_shr1(cl);
        xh ^= dpp_shr1(ch);
        asm volatile("" : "+v"(xl), "+v"(xh));                                        // (keeps the two XORs apart: one DPP operand each)
        const uint32_t rl = __builtin_amdgcn_alignbit(cl, ch, 31), rh = __builtin_amdgcn_alignbit(ch, cl, 31);       // rot(C, 1)
        const uint32_t tl = xl ^ dpp_shl1(rl), th = xh ^ dpp_shl1(rh);
        // rho
        const uint32_t l2 = swap ? th : tl, h2 = swap ? tl : th;
        const uint32_t nh = __builtin_amdgcn_alignbit(h2, l2, shift), nl = __builtin_amdgcn_alignbit(l2, h2, shift);
        // pi + chi
        // B[x], B[x+1] of the lane's row by two gathers per half; B[x+2] is the right neighbour's B[x+1] (one DPP read).  Measured
        // (tools/microbench/keccak_wave.hip, 108 permutations): all three by gathers 265 us at 256 signers / 300 at 1024 (four
        // waves of a CU share one LDS crossbar); this form 247 / 266; B[x] alone gathered and both neighbours by DPP (theta then
        // needs a second masked read, because position 6 is no longer valid) 264 / 273.
        const uint32_t b0l = (uint32_t)__builtin_amdgcn_ds_bpermute((int)idx0, (int)nl), b1l = (uint32_t)__builtin_amdgcn_ds_bpermute((int)idx1, (int)nl);
        const uint32_t b0h = (uint32_t)__builtin_amdgcn_ds_bpermute((int)idx0, (int)nh), b1h = (uint32_t)__builtin_amdgcn_ds_bpermute((int)idx1, (int)nh);
        const uint32_t b2l = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b1l, 0x101, 0xf, 0xf, true);
        const uint32_t b2h = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b1h, 0x101, 0xf, 0xf, true);
        lo = (uint32_t)__builtin_amdgcn_bitop3_b32(b0l, b1l, b2l, 0xD2);              // b0 ^ (~b1 & b2)
        hi = (uint32_t)__builtin_amdgcn_bitop3_b32(b0h, b1h, b2h, 0xD2);
    }

    __device__ __forceinline__ void permute() {
        round(0u, 0u);
#pragma unroll
        for (int i = 0; i < 23; ++i) round(kl[i], kh[i]);
        lo ^= origin ? kRoundConst[23][0] : 0u;
        hi ^= origin ? kRoundConst[23][1] : 0u;
    }
};

}  // namespace fzkw
