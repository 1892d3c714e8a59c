// Sustained clock and per-butterfly time of the fp64 (6-op, fz_mulmod4) and int32 (Shoup) butterflies under
// a long all-CU load, with the in-kernel clock read from s_memtime / s_memrealtime (100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr unsigned Q = 2147465729u;
constexpr int CH = 8;

__global__ void k_f64(unsigned long long* stamps, double* sink, int iters, double w, double w2) {
  const double K = 2147483648.0, kappa = 17919.0 / 2147483648.0, M = 6755399441055744.0 * K, q = (double)Q, qinv = 1.0 / q;
  double u[CH], v[CH];
  for (int i = 0; i < CH; ++i) { u[i] = (double)((threadIdx.x * 977u + i) % Q); v[i] = (double)((threadIdx.x * 131u + 7 * i) % Q); }
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      double uu = __builtin_fma(v[i], w2, M), cK = uu - M, t = __builtin_fma(v[i], w, -cK), r = __builtin_fma(cK, kappa, t);
      double a = u[i] + r, b = u[i] - r;
      u[i] = b; v[i] = a;
    }
    if ((it & 3) == 3) {
#pragma unroll
      for (int i = 0; i < CH; ++i) { double c = __builtin_rint(v[i] * qinv); v[i] = __builtin_fma(-c, q, v[i]); c = __builtin_rint(u[i] * qinv); u[i] = __builtin_fma(-c, q, u[i]); }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0; for (int i = 0; i < CH; ++i) s += u[i] + v[i];
  if (s == 0.123) sink[0] = s;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

__global__ void k_int(unsigned long long* stamps, unsigned* sink, int iters, unsigned w, unsigned wp) {
  unsigned u[CH], v[CH];
  for (int i = 0; i < CH; ++i) { u[i] = (threadIdx.x * 977u + i) % Q; v[i] = (threadIdx.x * 131u + 7 * i) % Q; }
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      unsigned qe = __umulhi(v[i], wp);
      unsigned r = v[i] * w - qe * Q;
      r = min(r, r - Q);
      unsigned a = u[i] + r; a = min(a, a - Q);
      unsigned b = u[i] - r; b = min(b, b + Q);
      u[i] = b; v[i] = a;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  unsigned s = 0; for (int i = 0; i < CH; ++i) s ^= u[i] ^ v[i];
  if (s == 0x12345678u) sink[0] = s;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int blocks = p.multiProcessorCount * 4, iters = 200000;     // 4 waves per SIMD, long enough for DVFS to settle
  unsigned long long* st; CHECK(hipMalloc(&st, blocks * 16)); void* sink; CHECK(hipMalloc(&sink, 64));
  unsigned long long* h = new unsigned long long[2 * blocks];
  for (int mode = 0; mode < 2; ++mode) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
      CHECK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL(k_f64, dim3(blocks), dim3(256), 0, 0, st, (double*)sink, iters, 123456789.0, 123456789.0 * (2147483648.0 / Q));
      else hipLaunchKernelGGL(k_int, dim3(blocks), dim3(256), 0, 0, st, (unsigned*)sink, iters, 123456789u, (unsigned)(((unsigned long long)123456789u << 32) / Q));
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    }
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipMemcpy(h, st, blocks * 16, hipMemcpyDeviceToHost));
    double clk = 0; for (int b = 0; b < blocks; ++b) clk += (double)h[2 * b] / (double)h[2 * b + 1] * 100.0; clk /= blocks;
    double waveops = (double)blocks * 4 * iters * CH, per_simd = waveops / (p.multiProcessorCount * 4.0);
    printf("%s butterfly: %.1f ms, %.2f ns per wave-butterfly per SIMD, in-kernel clock %.0f MHz -> %.1f cycles per butterfly\n",
           mode == 0 ? "fp64 6-op (+periodic reduce)" : "int32 Shoup", ms, ms * 1e6 / per_simd, clk, ms * 1e6 / per_simd * clk * 1e-3);
  }
  return 0;
}
