// Dependent-chain latency of the cross-lane primitives a wave-wide Keccak state can be built from, one wave on an idle CU
// (gfx950): plain VALU, DPP, v_permlane*_swap, ds_bpermute, ds_swizzle, an LDS write + read.  Cycles are s_memtime ticks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int kIter = 2048;
template <int mode>
__global__ __launch_bounds__(64) void probe(uint32_t seed, uint64_t *out, uint32_t *sink) {
    __shared__ uint32_t lds[256];
    const int lane = threadIdx.x;
    uint32_t v = seed * 2654435761u + lane, w = v ^ 0x9e3779b9u, idx = 4u * ((lane * 5 + 3) & 63), idx2 = 4u * ((lane * 7 + 1) & 63), idx3 = 4u * ((lane * 3 + 2) & 63);
    lds[lane] = v;
    __syncthreads();
    const uint64_t t0 = __builtin_readcyclecounter();
    const uint64_t r0 = wall_clock64();
#pragma unroll 1
    for (int i = 0; i < kIter; i += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            switch (mode) {
                case 0: asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v) : "v"(w)); break;
                case 1: asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(v) : "v"(w)); break;
                case 2: v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false); break;
                case 3: { auto s = __builtin_amdgcn_permlane32_swap(v, v, false, false); v = s[0] ^ s[1] ^ w; } break;
                case 4: v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)idx, (int)v); break;
                case 5: { const uint32_t a = (uint32_t)__builtin_amdgcn_ds_bpermute((int)idx, (int)v), b = (uint32_t)__builtin_amdgcn_ds_bpermute((int)idx2, (int)v),
                                          c = (uint32_t)__builtin_amdgcn_ds_bpermute((int)idx3, (int)v);
                          v = (uint32_t)__builtin_amdgcn_bitop3_b32(a, b, c, 0xD2); } break;
                case 6: { const uint32_t a = (uint32_t)__builtin_amdgcn_ds_bpermute((int)idx, (int)v), b = (uint32_t)__builtin_amdgcn_ds_bpermute((int)idx2, (int)v),
                                          c = (uint32_t)__builtin_amdgcn_ds_bpermute((int)idx3, (int)v);
                          const uint32_t a2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)idx, (int)w), b2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)idx2, (int)w),
                                          c2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)idx3, (int)w);
                          v = (uint32_t)__builtin_amdgcn_bitop3_b32(a, b, c, 0xD2); w = (uint32_t)__builtin_amdgcn_bitop3_b32(a2, b2, c2, 0xD2); } break;
                case 7: v = (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x801f) ^ w; break;      // swap within 32
                case 8: { lds[lane] = v; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                          v = lds[idx >> 2] ^ w; __builtin_amdgcn_wave_barrier(); } break;
                case 9: { auto s = __builtin_amdgcn_permlane16_swap(v, w, false, false); v = s[0] ^ s[1]; } break;
                case 10: v = (uint32_t)__builtin_amdgcn_readlane((int)v, 5) ^ w; break;
                case 11: { const uint32_t a = (uint32_t)__builtin_amdgcn_ds_bpermute((int)idx, (int)v), a2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)idx, (int)w);
                           v = a ^ w; w = a2 ^ v; } break;
            }
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    const uint64_t r1 = wall_clock64();
    if (lane == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
    sink[lane] = v ^ w;
}
int main() {
    uint64_t *d_out; uint32_t *d_sink;
    CHECK(hipMalloc(&d_out, 16)); CHECK(hipMalloc(&d_sink, 256));
    const char *names[] = {"v_xor_b32 (plain VALU)", "v_alignbit_b32", "v_xor_b32_dpp row_ror:8", "mov + v_permlane32_swap + 2 xor", "ds_bpermute_b32",
                           "3 ds_bpermute + v_bitop3", "6 ds_bpermute + 2 v_bitop3", "ds_swizzle + xor", "ds_write + ds_read + xor", "v_permlane16_swap + xor",
                           "v_readlane + xor", "2 ds_bpermute + 2 xor"};
    int wall_khz = 0; CHECK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
    for (int mode = 0; mode < 12; ++mode) {
        uint64_t h[2] = {0, 0};
        for (int rep = 0; rep < 3; ++rep) {
            switch (mode) {
#define L(M) case M: hipLaunchKernelGGL(probe<M>, dim3(1), dim3(64), 0, 0, 12345u + rep, d_out, d_sink); break;
                L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11)
#undef L
            }
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost));
        }
        printf("%-36s %7.1f s_memtime ticks per step, %7.1f ns per step (wall clock %d kHz)\n", names[mode], (double)h[0] / kIter, (double)h[1] / kIter * 1e6 / wall_khz, wall_khz);
    }
    return 0;
}
