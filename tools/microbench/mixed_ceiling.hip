// What does the chip sustain when a streaming kernel ALSO keeps the fp64 pipes busy?  The transform kernels read 4 B and
// write 4 B per coefficient and spend ~31 vector instructions on it (502 per wave and 4 KiB chunk: 432 fp64, 32 conversions);
// they plateau at 0.67-0.70 of 8 TB/s where a plain copy of the same buffers reaches 0.80, with the vector pipes ~68 % busy and
// the shader clock at 1.9 GHz (power) instead of 2.4.  Neither bound is reached -- so which ceiling is it?  This stand-alone
// kernel has the transforms' memory schedule exactly (resident grid, one 4 KiB chunk per wave and iteration, the next chunk's
// four 16-byte streaming loads issued first, four streaming stores last) and, in between, NOPS fp64 fused multiply-adds on the
// chunk's 16 values per lane (16 independent chains; no LDS, no waits other than the memory's own).  NOPS = 0 is the copy;
// NOPS = 464 is the transforms' fp64 + conversion count.  Printed per NOPS: time, GB/s, fraction of 8 TB/s, the shader clock
// (s_memtime ticks / s_memrealtime ticks x 100 MHz) and the share of issue slots the fp64 work fills.
// usage: mixed_ceiling [log2 rows = 18] [reps = 20]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int kChunk = 1024, kWaves = 4;

template <bool NT> __device__ __forceinline__ v4i ld(const int *p) {
  return NT ? __builtin_nontemporal_load(reinterpret_cast<const v4i *>(p)) : *reinterpret_cast<const v4i *>(p);
}
template <bool NT> __device__ __forceinline__ void st(int *p, v4i v) {
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v4i *>(p)); else *reinterpret_cast<v4i *>(p) = v;
}

// NT: streaming loads / stores.  BLOCKED: a wave's chunks are consecutive (its share of the batch is one contiguous range) instead
// of interleaved with every other wave's.
template <int NOPS, bool NT, bool BLOCKED>
__global__ __launch_bounds__(64 * kWaves) void stream_fma(const int *in, int *out, size_t tasks, double c1, double c2, unsigned long long *clk) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const size_t nw = (size_t)gridDim.x * kWaves, w = (size_t)blockIdx.x * kWaves + wave, per = (tasks + nw - 1) / nw;
  const size_t first = BLOCKED ? w * per : w, stride = BLOCKED ? 1 : nw, end = BLOCKED ? (first + per < tasks ? first + per : tasks) : tasks;
  if (first >= end) return;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  v4i cur[4], nxt[4];
  for (int k = 0; k < 4; ++k) cur[k] = ld<NT>(in + first * kChunk + 4 * lane + 256 * k);
  size_t task = first;
  auto body = [&](auto more_tag) __attribute__((always_inline)) {
    constexpr bool more = decltype(more_tag)::value;
    if (more)
      for (int k = 0; k < 4; ++k) nxt[k] = ld<NT>(in + (task + stride) * kChunk + 4 * lane + 256 * k);
    double a[16];
#pragma unroll
    for (int k = 0; k < 4; ++k) { a[4 * k] = cur[k].x; a[4 * k + 1] = cur[k].y; a[4 * k + 2] = cur[k].z; a[4 * k + 3] = cur[k].w; }
#pragma unroll
    for (int i = 0; i < NOPS / 16; ++i)
#pragma unroll
      for (int k = 0; k < 16; ++k) a[k] = __builtin_fma(a[k], c1, c2);
    v4i o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = v4i{(int)a[4 * k], (int)a[4 * k + 1], (int)a[4 * k + 2], (int)a[4 * k + 3]};
    if (more)
      for (int k = 0; k < 4; ++k) cur[k] = nxt[k];
    for (int k = 0; k < 4; ++k) st<NT>(out + task * kChunk + 4 * lane + 256 * k, o[k]);
  };
  for (; task + stride < end; task += stride) body(std::true_type());
  body(std::false_type());
  if (clk && lane == 0 && w == 0) { clk[0] = __builtin_amdgcn_s_memtime() - t0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

// what a pure WRITE stream sustains (keygen_bcast_fused writes 2l rows per key half and reads next to nothing from HBM): the same
// resident grid, every wave storing its 4 KiB chunks, streaming or normal stores
template <bool NT>
__global__ __launch_bounds__(64 * kWaves) void stream_fill(int *out, size_t tasks, int seed) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const size_t nw = (size_t)gridDim.x * kWaves, w = (size_t)blockIdx.x * kWaves + wave;
  const v4i o = {seed, lane, seed ^ lane, wave};
  for (size_t task = w; task < tasks; task += nw)
    for (int k = 0; k < 4; ++k) st<NT>(out + task * kChunk + 4 * lane + 256 * k, o);
}

template <bool NT>
int run_fill(int *out, size_t rows, int sets, int reps, int grid) {
  const size_t tasks = rows * 256 / kChunk, per = rows * 256;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) stream_fill<NT><<<grid, 64 * kWaves>>>(out + (i % sets) * per, tasks, i);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) stream_fill<NT><<<grid, 64 * kWaves>>>(out + (i % sets) * per, tasks, i);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps, gbs = rows * 1024.0 / us * 1e-3;
  printf("write only, %s stores, %4d workgroups  %9.2f us  %8.1f GB/s  (%4.1f %% of 8 TB/s)\n", NT ? "streaming" : "normal   ", grid, us, gbs, gbs / 80);
  return 0;
}

template <int NOPS, bool NT = true, bool BLOCKED = false>
int run(const int *in, int *out, size_t rows, int sets, int reps, unsigned long long *d_clk, int grid) {
  const size_t tasks = rows * 256 / kChunk, per = rows * 256;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) stream_fma<NOPS, NT, BLOCKED><<<grid, 64 * kWaves>>>(in + (i % sets) * per, out + (i % sets) * per, tasks, 0.999999, 0.25, d_clk);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) stream_fma<NOPS, NT, BLOCKED><<<grid, 64 * kWaves>>>(in + (i % sets) * per, out + (i % sets) * per, tasks, 0.999999, 0.25, d_clk);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long clk[2];
  CHECK(hipMemcpy(clk, d_clk, sizeof clk, hipMemcpyDeviceToHost));
  const double us = ms * 1e3 / reps, gbs = rows * 2048.0 / us * 1e-3, mhz = clk[1] ? 100.0 * clk[0] / clk[1] : 0;
  // issue slots: one wave instruction per 4 cycles and SIMD, 1024 SIMDs; the fp64 work = NOPS + 32 conversions per wave and chunk
  const double slots = us * mhz / 4 * 1024, used = (double)tasks * (NOPS + 32);
  printf("%5d fp64 ops per lane and chunk, %s, %s, %4d workgroups  %9.2f us  %8.1f GB/s  (%4.1f %% of 8 TB/s)  %5.0f MHz  fp64 pipes %4.1f %% busy\n", NOPS, NT ? "streaming" : "normal   ", BLOCKED ? "blocked    " : "interleaved", grid, us, gbs,
         gbs / 80, mhz, 100 * used / slots);
  return 0;
}

int main(int argc, char **argv) {
  const int logr = argc > 1 ? atoi(argv[1]) : 18, reps = argc > 2 ? atoi(argv[2]) : 20;
  const size_t rows = (size_t)1 << logr, per = rows * 256;
  const int sets = (int)std::max<size_t>(2, ((size_t)3 << 30) / (per * 8));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  int *in, *out;
  unsigned long long *d_clk;
  CHECK(hipMalloc(&in, per * 4 * sets)); CHECK(hipMalloc(&out, per * 4 * sets)); CHECK(hipMalloc(&d_clk, 16));
  CHECK(hipMemset(in, 1, per * 4 * sets)); CHECK(hipMemset(out, 0, per * 4 * sets));
  const int grid = prop.multiProcessorCount * 4;
  printf("# %zu rows of 256 int32 in, the same out (%zu MiB each way), %d operand sets rotated (cold), resident grid of %d workgroups x %d waves, %d launches timed\n",
         rows, per * 4 >> 20, sets, grid, kWaves, reps);
  if (run<0>(in, out, rows, sets, reps, d_clk, grid)) return 1;
  if (run<128>(in, out, rows, sets, reps, d_clk, grid)) return 1;
  if (run<256>(in, out, rows, sets, reps, d_clk, grid)) return 1;
  if (run<352>(in, out, rows, sets, reps, d_clk, grid)) return 1;
  if (run<464>(in, out, rows, sets, reps, d_clk, grid)) return 1;
  if (run<560>(in, out, rows, sets, reps, d_clk, grid)) return 1;
  if (run<672>(in, out, rows, sets, reps, d_clk, grid)) return 1;
  // the schedule itself: streaming or normal accesses, interleaved or blocked chunks, 4 / 6 / 8 workgroups per CU (the kernel
  // needs 40 registers: the transforms, at 121 and 40 KiB of LDS, hold 4)
  for (int wg = 4; wg <= 8; wg += 2) {
    const int g = prop.multiProcessorCount * wg;
    if (run<0, true, false>(in, out, rows, sets, reps, d_clk, g)) return 1;
    if (run<0, false, false>(in, out, rows, sets, reps, d_clk, g)) return 1;
    if (run<0, true, true>(in, out, rows, sets, reps, d_clk, g)) return 1;
    if (run<464, true, false>(in, out, rows, sets, reps, d_clk, g)) return 1;
    if (run<464, false, false>(in, out, rows, sets, reps, d_clk, g)) return 1;
    if (run<464, true, true>(in, out, rows, sets, reps, d_clk, g)) return 1;
  }
  for (int wg = 4; wg <= 8; wg += 4) {
    if (run_fill<true>(out, rows, sets, reps, prop.multiProcessorCount * wg)) return 1;
    if (run_fill<false>(out, rows, sets, reps, prop.multiProcessorCount * wg)) return 1;
  }
  return 0;
}
