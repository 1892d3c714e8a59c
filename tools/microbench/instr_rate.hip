// Instruction-rate microbenchmark for gfx950 (MI355X).
// Measures sustained wave64 issue rate of the integer / fp64 VALU instructions the
// modular-arithmetic butterflies are built from, so the NTT kernel's arithmetic core
// is chosen from measurements rather than from datasheet folklore.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/instr_rate instr_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITERS = 2048;   // loop trips
constexpr int CH = 8;         // independent chains per lane

// ---- 32-bit ops: 8 chains, one op each per trip -------------------------------------
#define KERNEL_U32(NAME, ASM)                                                        \
__global__ void NAME(unsigned* out, unsigned s0, unsigned s1) {                      \
  unsigned a[CH];                                                                    \
  _Pragma("unroll") for (int i = 0; i < CH; ++i) a[i] = threadIdx.x * 2654435761u + i + s0; \
  unsigned b = s1 | 1u, c = s0 + 12345u;                                             \
  for (int it = 0; it < ITERS; ++it) {                                               \
    _Pragma("unroll") for (int i = 0; i < CH; ++i) {                                 \
      asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c));                               \
    }                                                                                \
  }                                                                                  \
  unsigned r = 0;                                                                    \
  _Pragma("unroll") for (int i = 0; i < CH; ++i) r ^= a[i];                          \
  if (r == 0x12345678u) out[threadIdx.x] = r;                                        \
}

KERNEL_U32(k_add_u32,      "v_add_u32 %0, %0, %1")
KERNEL_U32(k_mul_lo_u32,   "v_mul_lo_u32 %0, %0, %1")
KERNEL_U32(k_mul_hi_u32,   "v_mul_hi_u32 %0, %0, %1")
KERNEL_U32(k_mul_hi_i32,   "v_mul_hi_i32 %0, %0, %1")
KERNEL_U32(k_mul_u32_u24,  "v_mul_u32_u24 %0, %0, %1")
KERNEL_U32(k_mul_hi_u32_u24,"v_mul_hi_u32_u24 %0, %0, %1")
KERNEL_U32(k_mad_u32_u24,  "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL_U32(k_min_u32,      "v_min_u32 %0, %0, %1")
KERNEL_U32(k_add3_u32,     "v_add3_u32 %0, %0, %1, %2")
KERNEL_U32(k_lshl_add_u32, "v_lshl_add_u32 %0, %0, 3, %1")
KERNEL_U32(k_fma_f32,      "v_fma_f32 %0, %0, %1, %2")
KERNEL_U32(k_mad_u64_u32_lo, "v_mad_u64_u32 v[100:101], vcc, %0, %1, v[100:101]\n\tv_mov_b32 %0, v100")
KERNEL_U32(k_cndmask,      "v_cmp_gt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc")
KERNEL_U32(k_sub_min,      "v_sub_u32 %0, %0, %1\n\tv_min_u32 %0, %0, %2")

// v_mad_u64_u32 proper: 64-bit accumulator chains
__global__ void k_mad_u64_u32(unsigned* out, unsigned s0, unsigned s1) {
  unsigned long long a[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) a[i] = threadIdx.x * 2654435761ull + i + s0;
  unsigned b = s1 | 1u, c = s0 + 12345u;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
    }
  }
  unsigned long long r = 0;
#pragma unroll
  for (int i = 0; i < CH; ++i) r ^= a[i];
  if (r == 0x12345678ull) out[threadIdx.x] = (unsigned)r;
}

// ---- fp64 ops ------------------------------------------------------------------------
#define KERNEL_F64(NAME, ASM)                                                        \
__global__ void NAME(unsigned* out, unsigned s0, unsigned s1) {                      \
  double a[CH];                                                                      \
  _Pragma("unroll") for (int i = 0; i < CH; ++i) a[i] = 1.0 + 1e-9 * (threadIdx.x + i + s0); \
  double b = 1.0 + 1e-12 * s1, c = 1e-30 * s0;                                       \
  for (int it = 0; it < ITERS; ++it) {                                               \
    _Pragma("unroll") for (int i = 0; i < CH; ++i) {                                 \
      asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c));                               \
    }                                                                                \
  }                                                                                  \
  double r = 0;                                                                      \
  _Pragma("unroll") for (int i = 0; i < CH; ++i) r += a[i];                          \
  if (r == 0.12345678) out[threadIdx.x] = 1;                                         \
}
KERNEL_F64(k_fma_f64,   "v_fma_f64 %0, %0, %1, %2")
KERNEL_F64(k_mul_f64,   "v_mul_f64 %0, %0, %1")
KERNEL_F64(k_add_f64,   "v_add_f64 %0, %0, %1")
KERNEL_F64(k_rndne_f64, "v_rndne_f64 %0, %0")
KERNEL_F64(k_floor_f64, "v_floor_f64 %0, %0")

// conversions: i32 -> f64 -> i32 round trip (two instructions per "op")
__global__ void k_cvt_rt(unsigned* out, unsigned s0, unsigned s1) {
  int a[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) a[i] = threadIdx.x + i + s0;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      double t;
      asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(t) : "v"(a[i]));
      asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(a[i]) : "v"(t));
    }
  }
  int r = 0;
#pragma unroll
  for (int i = 0; i < CH; ++i) r ^= a[i];
  if (r == 0x12345678) out[threadIdx.x] = r;
}

// ---- whole butterflies (C++; the compiler schedules) -----------------------------------
constexpr unsigned Q = 2147465729u;

__device__ __forceinline__ unsigned csub(unsigned x) { unsigned y = x - Q; return y < x ? y : x; }  // min_u32 form

// Shoup: x any u32, w < q, wp = floor(w*2^32/q) -> [0,2q)
__device__ __forceinline__ unsigned mul_shoup(unsigned x, unsigned w, unsigned wp) {
  unsigned qe = __umulhi(x, wp);
  return x * w - qe * Q;
}
__global__ void k_bfly_int(unsigned* out, unsigned s0, unsigned s1) {
  unsigned u[CH], v[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) { u[i] = (threadIdx.x * 2654435761u + i + s0) % Q; v[i] = (u[i] * 7u + s1) % Q; }
  unsigned w = (s1 * 3u + 5u) % Q;
  unsigned wp = (unsigned)(((unsigned long long)w << 32) / Q);
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      unsigned r = mul_shoup(v[i], w, wp);
      r = min(r, r - Q);
      unsigned a = u[i] + r; a = min(a, a - Q);
      unsigned b = u[i] - r; b = min(b, b + Q);
      u[i] = a; v[i] = b;
    }
  }
  unsigned r = 0;
#pragma unroll
  for (int i = 0; i < CH; ++i) r ^= u[i] ^ v[i];
  if (r == 0x12345678u) out[threadIdx.x] = r;
}

// Montgomery-with-precomputed (w, w*qinv): signed result; 3 multiplies
__global__ void k_bfly_mont(unsigned* out, unsigned s0, unsigned s1) {
  unsigned u[CH], v[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) { u[i] = (threadIdx.x * 2654435761u + i + s0) % Q; v[i] = (u[i] * 7u + s1) % Q; }
  unsigned w = (s1 * 3u + 5u) % Q;
  unsigned wq = w * 2497427967u;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      unsigned m = v[i] * wq;
      unsigned r = __umulhi(v[i], w) - __umulhi(m, Q);   // in (-q, q) mod 2^32
      r = min(r, r + Q);
      unsigned a = u[i] + r; a = min(a, a - Q);
      unsigned b = u[i] - r; b = min(b, b + Q);
      u[i] = a; v[i] = b;
    }
  }
  unsigned r = 0;
#pragma unroll
  for (int i = 0; i < CH; ++i) r ^= u[i] ^ v[i];
  if (r == 0x12345678u) out[threadIdx.x] = r;
}

// fp64 butterfly: values exact integers in doubles; 6-op mulmod + add + sub
__global__ void k_bfly_f64(unsigned* out, unsigned s0, unsigned s1) {
  double u[CH], v[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) { u[i] = (double)((threadIdx.x * 2654435761u + i + s0) % Q); v[i] = (double)((threadIdx.x * 7u + s1 + i) % Q); }
  const double q = (double)Q, qinv = 1.0 / (double)Q;
  double w = (double)((s1 * 3u + 5u) % Q);
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      double h = v[i] * w;
      double l = __builtin_fma(v[i], w, -h);
      double c = __builtin_rint(h * qinv);
      double d = __builtin_fma(-c, q, h);
      double r = d + l;
      double a = u[i] + r, b = u[i] - r;
      // keep magnitudes bounded for the benchmark (not needed in the real kernel's 8 stages)
      u[i] = b; v[i] = a;
      if ((it & 7) == 7) { double cc = __builtin_rint(v[i] * qinv); v[i] = __builtin_fma(-cc, q, v[i]); cc = __builtin_rint(u[i] * qinv); u[i] = __builtin_fma(-cc, q, u[i]); }
    }
  }
  double r = 0;
#pragma unroll
  for (int i = 0; i < CH; ++i) r += u[i] + v[i];
  if (r == 0.12345678) out[threadIdx.x] = 1;
}

typedef void (*kern_t)(unsigned*, unsigned, unsigned);
struct Entry { const char* name; kern_t k; double ops_per_trip_per_chain; };

int main(int argc, char** argv) {
  int dev = 0; CHECK(hipSetDevice(dev));
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, dev));
  printf("device: %s  CUs=%d  clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  unsigned* out; CHECK(hipMalloc(&out, 4096));
  std::vector<Entry> es = {
    {"v_add_u32", k_add_u32, 1}, {"v_fma_f32", k_fma_f32, 1}, {"v_min_u32", k_min_u32, 1},
    {"v_add3_u32", k_add3_u32, 1}, {"v_lshl_add_u32", k_lshl_add_u32, 1},
    {"v_mul_u32_u24", k_mul_u32_u24, 1}, {"v_mul_hi_u32_u24", k_mul_hi_u32_u24, 1}, {"v_mad_u32_u24", k_mad_u32_u24, 1},
    {"v_mul_lo_u32", k_mul_lo_u32, 1}, {"v_mul_hi_u32", k_mul_hi_u32, 1}, {"v_mul_hi_i32", k_mul_hi_i32, 1},
    {"v_mad_u64_u32", k_mad_u64_u32, 1},
    {"cmp+cndmask (2 instr)", k_cndmask, 1}, {"sub+min (2 instr)", k_sub_min, 1},
    {"v_fma_f64", k_fma_f64, 1}, {"v_mul_f64", k_mul_f64, 1}, {"v_add_f64", k_add_f64, 1},
    {"v_rndne_f64", k_rndne_f64, 1}, {"v_floor_f64", k_floor_f64, 1},
    {"cvt f64<-i32 + i32<-f64 (2 instr)", k_cvt_rt, 1},
    {"butterfly int Shoup", k_bfly_int, 1}, {"butterfly int Mont3", k_bfly_mont, 1}, {"butterfly fp64", k_bfly_f64, 1},
  };
  const int nCU = p.multiProcessorCount;
  for (int wavesPerSimd : {1, 2, 4}) {
    printf("--- %d wave(s) per SIMD (blocks of 256 threads, %d blocks/CU) ---\n", wavesPerSimd, wavesPerSimd);
    for (auto& e : es) {
      int blocks = nCU * wavesPerSimd;
      hipEvent_t t0, t1; CHECK(hipEventCreate(&t0)); CHECK(hipEventCreate(&t1));
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 1u, 3u);  // warm
      CHECK(hipDeviceSynchronize());
      float best = 1e30f;
      for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(t0));
        hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 1u, 3u);
        CHECK(hipEventRecord(t1)); CHECK(hipEventSynchronize(t1));
        float ms; CHECK(hipEventElapsedTime(&ms, t0, t1)); if (ms < best) best = ms;
      }
      double waveops = (double)blocks * 4 /*waves*/ * ITERS * CH;     // wave-level "ops"
      double per_simd_ops = waveops / (nCU * 4.0);
      double ns_per_op = best * 1e6 / per_simd_ops;                   // ns per wave-op per SIMD
      printf("%-36s %8.3f ms  %7.2f ns/waveop/SIMD  (~%.1f cyc @2.4GHz)\n", e.name, best, ns_per_op, ns_per_op * 2.4);
      CHECK(hipEventDestroy(t0)); CHECK(hipEventDestroy(t1));
    }
  }
  return 0;
}
