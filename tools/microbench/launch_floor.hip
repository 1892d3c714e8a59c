// Launch / latency floor on gfx950: back-to-back launches of (a) an empty kernel, (b) a 4 MiB int4 copy
// (the byte count of one B=4096, d=256 NTT launch), (c) the same copy with 64-thread blocks.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void k_empty() {}
__global__ void k_copy(const int4* in, int4* out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i];
}
__global__ void k_copy_stride(const int4* in, int4* out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
int main() {
  const size_t bytes = 4096ull * 256 * 4, n = bytes / 16;
  int4 *a, *b; CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes)); CHECK(hipMemset(a, 1, bytes));
  hipEvent_t t0, t1; CHECK(hipEventCreate(&t0)); CHECK(hipEventCreate(&t1));
  for (int mode = 0; mode < 3; ++mode) {
    const int reps = 200;
    for (int w = 0; w < 2; ++w) {
      if (w) CHECK(hipEventRecord(t0));
      for (int r = 0; r < reps; ++r) {
        if (mode == 0) hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, 0);
        else if (mode == 1) hipLaunchKernelGGL(k_copy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (r & 1) ? b : a, (r & 1) ? a : b, n);
        else hipLaunchKernelGGL(k_copy, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, (r & 1) ? b : a, (r & 1) ? a : b, n);
      }
      if (w) { CHECK(hipEventRecord(t1)); CHECK(hipEventSynchronize(t1)); }
      else CHECK(hipDeviceSynchronize());
    }
    float ms; CHECK(hipEventElapsedTime(&ms, t0, t1));
    printf("%s: %.2f us per launch (back-to-back, dependent)\n", mode == 0 ? "empty kernel 1024x256" : mode == 1 ? "copy 4 MiB, 256-thread blocks" : "copy 4 MiB, 64-thread blocks", ms * 1e3 / reps);
  }
  // large streaming copy: what this box sustains for a 1:1 read/write mix (the NTT's traffic shape)
  {
    const size_t big = 1ull << 30, nb = big / 16;
    int4 *c, *d2; CHECK(hipMalloc(&c, big)); CHECK(hipMalloc(&d2, big)); CHECK(hipMemset(c, 1, big));
    for (int blocks : {2048, 8192, 0}) {
      const unsigned g = blocks ? (unsigned)blocks : (unsigned)((nb + 255) / 256);
      for (int w = 0; w < 2; ++w) {
        if (w) CHECK(hipEventRecord(t0));
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_copy_stride, dim3(g), dim3(256), 0, 0, c, d2, nb);
        if (w) { CHECK(hipEventRecord(t1)); CHECK(hipEventSynchronize(t1)); } else CHECK(hipDeviceSynchronize());
      }
      float ms; CHECK(hipEventElapsedTime(&ms, t0, t1));
      printf("copy 1 GiB -> 1 GiB, %u blocks: %.1f us, %.2f TB/s (read+write)\n", g, ms * 1e3 / 5, 2.0 * big / (ms / 5 * 1e-3) / 1e12);
    }
  }
  return 0;
}
