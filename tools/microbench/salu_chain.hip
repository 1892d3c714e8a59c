// Dependent-chain cost of the MT19937 seeding step (init_by_array, Modules/_randommodule.c) on the SCALAR unit against the vector
// unit, one wave on an idle CU: prev = (tab ^ ((prev ^ (prev >> 30)) * 1664525)) + key, 4096 steps, shader clock cycles per step.
//   salu_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(64) void chain(const uint32_t *tab, uint32_t key, int steps, uint32_t *out, unsigned long long *cyc) {
    const int lane = threadIdx.x;
    uint32_t prev = MODE == 0 ? key : key + (uint32_t)lane;          // 0: wave-uniform (the compiler keeps the chain in scalar registers)
    uint32_t w = 0;
    typedef const __attribute__((address_space(4))) uint32_t *CTab;      // constant address space: scalar loads
    CTab ctab = (CTab)tab;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < steps; i += 16) {
        uint32_t t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = ctab[(i + u) & 1023];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            prev = (t[u] ^ ((prev ^ (prev >> 30)) * 1664525u)) + key;
            if (MODE == 0) {                                             // the word goes to lane i of a vector register
                const uint32_t sel = (uint32_t)(i + u) & 63u;
                asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(w) : "s"(prev), "s"(sel) : "m0");
            }
            else w ^= prev;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[lane] = w ^ prev;
    if (lane == 0) *cyc = t1 - t0;
}

int main() {
    uint32_t h_tab[1024];
    for (int i = 0; i < 1024; ++i) h_tab[i] = 1812433253u * (uint32_t)i + 12345u;
    uint32_t *d_tab, *d_out;
    unsigned long long *d_cyc, h_cyc;
    CHECK(hipMalloc(&d_tab, sizeof(h_tab)));
    CHECK(hipMalloc(&d_out, 256));
    CHECK(hipMalloc(&d_cyc, 8));
    CHECK(hipMemcpy(d_tab, h_tab, sizeof(h_tab), hipMemcpyHostToDevice));
    const int steps = 4096;
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(chain<0>, dim3(1), dim3(64), 0, 0, d_tab, 777u, steps, d_out, d_cyc);
            else hipLaunchKernelGGL(chain<1>, dim3(1), dim3(64), 0, 0, d_tab, 777u, steps, d_out, d_cyc);
            CHECK(hipDeviceSynchronize());
        }
        CHECK(hipMemcpy(&h_cyc, d_cyc, 8, hipMemcpyDeviceToHost));
        printf("%s chain: %.1f counter ticks per step (s_memtime / readcyclecounter units)\n", mode == 0 ? "scalar (uniform)" : "vector (per lane)", (double)h_cyc / steps);
    }
    // wall-clock per step from events over a longer chain
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) {
        const int big = 1 << 20;
        CHECK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(chain<0>, dim3(1), dim3(64), 0, 0, d_tab, 777u, big, d_out, d_cyc);
        else hipLaunchKernelGGL(chain<1>, dim3(1), dim3(64), 0, 0, d_tab, 777u, big, d_out, d_cyc);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s chain: %.2f ns per step over %d steps\n", mode == 0 ? "scalar (uniform)" : "vector (per lane)", ms * 1e6 / big, big);
    }
    return 0;
}
