// Ceilings for the fused scheme kernels that sit below 0.40 of the HBM peak: what does a stand-alone kernel with
// THEIR traffic shape and THEIR instruction count sustain?  One wave per 1 KiB row-task, resident grid, the next task's loads
// issued before this task's arithmetic (the fused kernels' schedule), NOPS dependent-chain-free fp64 FMAs per task on the loaded
// values, then the stores.  Shapes (IN = KiB read from HBM per task, L2 = KiB read from a small table that stays in the L2,
// OUT = KiB written, NOPS = wave instructions per task as counted in the kernels' ISA):
//   verify_fused    IN 1, L2 1, OUT 0, NOPS 204 (132 fp64 + 8 quarter-rate 64-bit multiply-adds counted as 32 + 40 others)
//   polymul_fused   IN 2, L2 0, OUT 1, NOPS 395
//   aggregation     IN 1, L2 0, OUT 0, NOPS 12, with as few tasks as 128 / 256 signers have ((l + 1) rows each)
// and NOPS = 0 of each shape (the pure stream).  Printed: time, GB/s of HBM-side algorithmic bytes, fraction of 8 TB/s, shader clock.
// LDS: the radix-4 passes exchange the four values of every lane through LDS between passes (three round trips per transform at
// degree 256); the "+ N LDS round trips" rows add them.
// usage: shape_ceiling [reps = 20]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int kWaves = 4;

// LDSRT: round trips through LDS per task as the radix-4 passes make them -- every lane writes its first four values (8-byte
// stores, the passes' stride-s pattern stands in as a lane rotation), the wave synchronises, every lane reads four back -- with
// the task's NOPS instructions spread between them
template <int NOPS, int IN, int L2, int OUT, int LDSRT = 0>
__global__ __launch_bounds__(64 * kWaves) void shape(const int *in, const int *tab, int tab_rows, int *out, size_t tasks, double c1, double c2,
                                                     unsigned long long *clk, double *sink) {
  __shared__ double lds[kWaves * 256 + 8];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const size_t nw = (size_t)gridDim.x * kWaves, w = (size_t)blockIdx.x * kWaves + wave;
  if (w >= tasks) return;
  double *region = lds + wave * 256;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  v4i cur[IN + L2], nxt[IN + L2];
  auto fetch = [&](v4i (&dst)[IN + L2], size_t t) {
#pragma unroll
    for (int k = 0; k < IN; ++k) dst[k] = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(in + (t * IN + k) * 256 + 4 * lane));
#pragma unroll
    for (int k = 0; k < L2; ++k) dst[IN + k] = *reinterpret_cast<const v4i *>(tab + ((t + k) % tab_rows) * 256 + 4 * lane);
  };
  fetch(cur, w);
  double mx = 0;
  for (size_t t = w; t < tasks; t += nw) {
    const size_t tn = t + nw < tasks ? t + nw : t;
    fetch(nxt, tn);
    double a[4 * (IN + L2)];
#pragma unroll
    for (int k = 0; k < IN + L2; ++k) { a[4 * k] = cur[k].x; a[4 * k + 1] = cur[k].y; a[4 * k + 2] = cur[k].z; a[4 * k + 3] = cur[k].w; }
    constexpr int NV = 4 * (IN + L2);
    constexpr int SEG = LDSRT + 1, PER = NOPS / NV / SEG;
#pragma unroll
    for (int seg = 0; seg < SEG; ++seg) {
#pragma unroll
      for (int i = 0; i < (seg == SEG - 1 ? NOPS / NV - PER * (SEG - 1) : PER); ++i)
#pragma unroll
        for (int k = 0; k < NV; ++k) a[k] = __builtin_fma(a[k], c1, c2);
      if (seg < LDSRT) {
        const int rot = (lane + 16 * (seg + 1)) & 63;
#pragma unroll
        for (int k = 0; k < 4; ++k) region[64 * k + lane] = a[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = region[64 * k + rot];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      }
    }
#pragma unroll
    for (int k = 0; k < OUT; ++k)
      __builtin_nontemporal_store(v4i{(int)a[4 * k], (int)a[4 * k + 1], (int)a[4 * k + 2], (int)a[4 * k + 3]}, reinterpret_cast<v4i *>(out + (t * OUT + k) * 256 + 4 * lane));
    if (OUT == 0) {
#pragma unroll
      for (int k = 0; k < NV; ++k) mx = __builtin_fmax(mx, a[k]);
    }
#pragma unroll
    for (int k = 0; k < IN + L2; ++k) cur[k] = nxt[k];
  }
  if (OUT == 0 && mx == 1.2345e300) *sink = mx;                 // (keeps the arithmetic alive; never true)
  if (clk && lane == 0 && w == 0) { clk[0] = __builtin_amdgcn_s_memtime() - t0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

template <int NOPS, int IN, int L2, int OUT, int LDSRT = 0>
int run(const char *name, const int *in, const int *tab, int *out, size_t tasks, size_t pool_tasks, int reps, int grid, unsigned long long *d_clk, double *d_sink) {
  const size_t step_in = tasks * IN * 256, step_out = tasks * OUT * 256;
  // operand sets rotated (cold): as many as BOTH pools hold (in: pool_tasks rows, out: pool_tasks / 3 rows)
  size_t sets_ = pool_tasks / (tasks * IN);
  if (OUT) sets_ = std::min(sets_, (pool_tasks / 3) / (tasks * OUT));
  if (sets_ < 1) { printf("%s: %zu tasks do not fit the pools\n", name, tasks); return 1; }
  const int sets = (int)sets_;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int g = (int)std::min<size_t>((size_t)grid, (tasks + kWaves - 1) / kWaves);
  for (int i = 0; i < 3; ++i) shape<NOPS, IN, L2, OUT, LDSRT><<<g, 64 * kWaves>>>(in + (i % sets) * step_in, tab, 83, out + (i % sets) * step_out, tasks, 0.999999, 0.25, d_clk, d_sink);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) shape<NOPS, IN, L2, OUT, LDSRT><<<g, 64 * kWaves>>>(in + (i % sets) * step_in, tab, 83, out + (i % sets) * step_out, tasks, 0.999999, 0.25, d_clk, d_sink);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long clk[2];
  CHECK(hipMemcpy(clk, d_clk, sizeof clk, hipMemcpyDeviceToHost));
  const double us = ms * 1e3 / reps, gbs = tasks * (IN + OUT) * 1024.0 / us * 1e-3, mhz = clk[1] ? 100.0 * clk[0] / clk[1] : 0;
  fflush(stdout);
  printf("%-44s %8zu tasks  %4d instr + %d LDS round trips per task  %9.2f us  %8.1f GB/s  (%4.1f %% of 8 TB/s)  %5.0f MHz  %d sets\n", name, tasks, NOPS, LDSRT, us, gbs, gbs / 80, mhz, sets);
  return 0;
}

int main(int argc, char **argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 20;
  setvbuf(stdout, nullptr, _IOLBF, 0);
  const size_t pool_tasks = (size_t)3 << 20;                   // 3 GiB of input rows, 1 GiB of output rows
  int *in, *out, *tab;
  unsigned long long *d_clk; double *d_sink;
  CHECK(hipMalloc(&in, pool_tasks * 1024)); CHECK(hipMalloc(&out, (pool_tasks / 3) * 1024)); CHECK(hipMalloc(&tab, 84 * 1024));
  CHECK(hipMalloc(&d_clk, 16)); CHECK(hipMalloc(&d_sink, 8));
  CHECK(hipMemset(in, 1, pool_tasks * 1024)); CHECK(hipMemset(tab, 1, 84 * 1024)); CHECK(hipMemset(d_clk, 0, 16));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const size_t l = 83;
#define R(NAME, NOPS, IN, L2, OUT, TASKS, GRID) if (run<NOPS, IN, L2, OUT>(NAME, in, tab, out, (size_t)(TASKS), pool_tasks, reps, GRID, d_clk, d_sink)) return 1;
#define RL(NAME, NOPS, IN, L2, OUT, RT, TASKS, GRID) if (run<NOPS, IN, L2, OUT, RT>(NAME, in, tab, out, (size_t)(TASKS), pool_tasks, reps, GRID, d_clk, d_sink)) return 1;
  for (size_t G : {1024, 8192}) {
    printf("# verify_fused's shape, %zu aggregates of %zu rows (resident grid of %d workgroups x %d waves)\n", G, l, 2 * cus, kWaves);
    R("verify: stream only", 0, 1, 1, 0, G * l, 2 * cus)
    R("verify: 132 fp64", 132, 1, 1, 0, G * l, 2 * cus)
    R("verify: 168 (fp64-split sums)", 168, 1, 1, 0, G * l, 2 * cus)
    R("verify: 204 (as counted)", 204, 1, 1, 0, G * l, 2 * cus)
    R("verify: 204, 4 workgroups per CU", 204, 1, 1, 0, G * l, 4 * cus)
    R("verify: 204, 5 per CU (the kernel's)", 204, 1, 1, 0, G * l, 5 * cus)
    R("verify: 168, 5 per CU", 168, 1, 1, 0, G * l, 5 * cus)
    R("verify: 132, 5 per CU", 132, 1, 1, 0, G * l, 5 * cus)
    R("verify: stream only, 5 per CU", 0, 1, 1, 0, G * l, 5 * cus)
    RL("verify: 168 + the passes' 3 LDS round trips, 5 per CU", 168, 1, 1, 0, 3, G * l, 5 * cus)
    RL("verify: 204 + 3 LDS round trips, 5 per CU", 204, 1, 1, 0, 3, G * l, 5 * cus)
  }
  printf("# polymul_fused's shape, 65536 products (2 rows in, 1 row out)\n");
  R("polymul: stream only", 0, 2, 0, 1, 65536, 2 * cus)
  R("polymul: 324 (3 x 108 fp64)", 324, 2, 0, 1, 65536, 2 * cus)
  R("polymul: 395 (as counted)", 392, 2, 0, 1, 65536, 2 * cus)
  R("polymul: 395, 3 workgroups per CU", 392, 2, 0, 1, 65536, 3 * cus)
  R("polymul: 536 (the loop's own count), 4 per CU", 528, 2, 0, 1, 65536, 4 * cus)
  R("polymul: 536, 3 per CU", 528, 2, 0, 1, 65536, 3 * cus)
  RL("polymul: 395 + 9 LDS round trips, 4 per CU", 392, 2, 0, 1, 9, 65536, 4 * cus)
  RL("polymul: 536 + 9 LDS round trips, 4 per CU", 528, 2, 0, 1, 9, 65536, 4 * cus)
  RL("polymul: 536 + 12 LDS round trips, 4 per CU", 528, 2, 0, 1, 12, 65536, 4 * cus)
  printf("# few-signer aggregation: (l + 1) rows per signer read once, 12 instructions per row, l rows written\n");
  R("aggregate 128 signers: read stream", 12, 1, 0, 0, 128 * (l + 1), 8 * cus)
  R("aggregate 256 signers: read stream", 12, 1, 0, 0, 256 * (l + 1), 8 * cus)
  R("aggregate 1024 signers: read stream", 12, 1, 0, 0, 1024 * (l + 1), 8 * cus)
  return 0;
}
