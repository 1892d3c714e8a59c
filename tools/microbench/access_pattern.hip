// Does the ORDER in which a sliced kernel walks the signers' rows cost bandwidth?  The one-pass signing + aggregation launch
// (aggregate_onepass<.., SIGN>) streams at 4.9 TB/s where the flat sign kernel reaches 5.9: its workgroups each own a few rows
// (3-4 KiB) of EVERY signer of their slice, so a workgroup's consecutive reads are 170 KiB apart.  This stand-alone kernel moves
// the same bytes -- per signer two key halves of L rows read, L rows written, 1 KiB rows -- in three orders:
//   flat      consecutive threads, consecutive addresses (the sign kernel's order)
//   sliced    workgroup = AR rows of every signer of its slice; its 8 waves take the slice's signers round-robin (the launch today)
//   banded    workgroup = 8 x AR consecutive rows; wave w takes rows [w * AR, (w + 1) * AR) of the SAME signer: 24-32 KiB contiguous
//             per signer and workgroup
// and prints the rate of each, cold (operand sets rotate).  usage: access_pattern [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int D4 = 64;                     // int4 columns per row (degree 256)
constexpr int WAVES = 8;

__device__ __forceinline__ int4 mix(int4 a, int4 b) { return make_int4(a.x * 3 + b.x, a.y * 5 + b.y, a.z * 7 + b.z, a.w * 9 + b.w); }

__global__ void k_flat(const int4 *sk, int4 *sig, size_t n_sig, int L) {
  const size_t per = (size_t)L * D4, total = n_sig * per, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const size_t b = i / per, rem = i % per;
    sig[i] = mix(sk[b * 2 * per + rem], sk[(b * 2 + 1) * per + rem]);
  }
}

// workgroup (cb, slice): AR rows of every signer of the slice; waves take signers round-robin, 2 signers in flight per wave
template <int AR>
__global__ __launch_bounds__(64 * WAVES) void k_sliced(const int4 *sk, int4 *sig, size_t n_sig, int L, int ncb, int nsl, int4 *sink) {
  const int cb = blockIdx.x % ncb, sb = blockIdx.x / ncb, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t per = (size_t)L * D4, base = n_sig / nsl, i0 = sb * base, i1 = i0 + base;
  size_t col[AR];
  for (int r = 0; r < AR; ++r) { size_t c = (size_t)cb * 64 * AR + r * 64 + lane; col[r] = c < per ? c : per - 1; }
  int4 acc = make_int4(0, 0, 0, 0);
  for (size_t i = i0 + wave; i < i1; i += 2 * WAVES) {
    int4 a[2][AR], b[2][AR];
    for (int s = 0; s < 2; ++s) { const size_t is = i + s * WAVES; if (is < i1) for (int r = 0; r < AR; ++r) { a[s][r] = sk[is * 2 * per + col[r]]; b[s][r] = sk[(is * 2 + 1) * per + col[r]]; } }
    for (int s = 0; s < 2; ++s) { const size_t is = i + s * WAVES; if (is < i1) for (int r = 0; r < AR; ++r) { const int4 v = mix(a[s][r], b[s][r]); sig[is * per + col[r]] = v; acc = mix(acc, v); } }
  }
  if (acc.x == 0x7fffffff) sink[0] = acc;       // keeps the sums alive
}

// workgroup (band, slice): WAVES x AR consecutive rows; wave w owns rows [w * AR ..) of every signer of the slice, signers in order
template <int AR>
__global__ __launch_bounds__(64 * WAVES) void k_banded(const int4 *sk, int4 *sig, size_t n_sig, int L, int nband, int nsl, int4 *sink) {
  const int band = blockIdx.x % nband, sb = blockIdx.x / nband, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t per = (size_t)L * D4, base = n_sig / nsl, i0 = sb * base, i1 = i0 + base;
  size_t col[AR];
  for (int r = 0; r < AR; ++r) { size_t c = ((size_t)band * WAVES + wave) * 64 * AR + r * 64 + lane; col[r] = c < per ? c : per - 1; }
  int4 acc = make_int4(0, 0, 0, 0);
  for (size_t i = i0; i < i1; i += 2) {
    int4 a[2][AR], b[2][AR];
    for (int s = 0; s < 2; ++s) { const size_t is = i + s; if (is < i1) for (int r = 0; r < AR; ++r) { a[s][r] = sk[is * 2 * per + col[r]]; b[s][r] = sk[(is * 2 + 1) * per + col[r]]; } }
    for (int s = 0; s < 2; ++s) { const size_t is = i + s; if (is < i1) for (int r = 0; r < AR; ++r) { const int4 v = mix(a[s][r], b[s][r]); sig[is * per + col[r]] = v; acc = mix(acc, v); } }
  }
  if (acc.x == 0x7fffffff) sink[0] = acc;
}

int main(int argc, char **argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 60;
  const int L = 84;                                  // 83 rows rounded up: 28 blocks of 3, 21 of 4
  const size_t n_sig = 1024, per = (size_t)L * D4;
  const size_t in_bytes = n_sig * 2 * per * 16, out_bytes = n_sig * per * 16, moved = in_bytes + out_bytes;
  const int NSETS = 6;                               // 6 x 258 MB of inputs: nothing is cache-resident
  int4 *in[NSETS], *out[NSETS], *sink;
  for (int k = 0; k < NSETS; ++k) { CHECK(hipMalloc(&in[k], in_bytes)); CHECK(hipMalloc(&out[k], out_bytes)); CHECK(hipMemset(in[k], k + 1, in_bytes)); }
  CHECK(hipMalloc(&sink, 64));
  hipEvent_t t0, t1; CHECK(hipEventCreate(&t0)); CHECK(hipEventCreate(&t1));
  struct Case { const char *name; int kind, ar, nsl; };
  const Case cases[] = {{"flat (grid-stride, 2048 x 256 threads)", 0, 0, 0},
                        {"sliced AR=4: 21 column blocks x 12 slices = 252 workgroups", 1, 4, 12},
                        {"sliced AR=3: 28 column blocks x  9 slices = 252 workgroups", 1, 3, 9},
                        {"sliced AR=3: 28 column blocks x  8 slices = 224 workgroups", 1, 3, 8},
                        {"banded AR=3: 4 bands of 24 KiB x 64 slices = 256 workgroups (7 waves used of 8)", 2, 3, 64},
                        {"banded AR=4: 3 bands of 32 KiB x 85 slices = 255 workgroups (7 waves used of 8)", 2, 4, 85}};
  for (const Case &c : cases) {
    float best = 1e30f;
    for (int pass = 0; pass < 3; ++pass) {
      for (int w = 0; w < 2; ++w) {
        if (w) CHECK(hipEventRecord(t0));
        for (int r = 0; r < (w ? reps : 10); ++r) {
          const int4 *i4 = in[r % NSETS]; int4 *o4 = out[r % NSETS];
          if (c.kind == 0) hipLaunchKernelGGL(k_flat, dim3(2048), dim3(256), 0, 0, i4, o4, n_sig, L);
          else if (c.kind == 1 && c.ar == 4) hipLaunchKernelGGL(k_sliced<4>, dim3(21 * c.nsl), dim3(512), 0, 0, i4, o4, n_sig, L, 21, c.nsl, sink);
          else if (c.kind == 1) hipLaunchKernelGGL(k_sliced<3>, dim3(28 * c.nsl), dim3(512), 0, 0, i4, o4, n_sig, L, 28, c.nsl, sink);
          else if (c.ar == 3) hipLaunchKernelGGL(k_banded<3>, dim3(4 * c.nsl), dim3(512), 0, 0, i4, o4, n_sig, L, 4, c.nsl, sink);
          else hipLaunchKernelGGL(k_banded<4>, dim3(3 * c.nsl), dim3(512), 0, 0, i4, o4, n_sig, L, 3, c.nsl, sink);
        }
        if (w) { CHECK(hipEventRecord(t1)); CHECK(hipEventSynchronize(t1)); } else CHECK(hipDeviceSynchronize());
      }
      float ms; CHECK(hipEventElapsedTime(&ms, t0, t1));
      if (ms / reps < best) best = ms / reps;
    }
    printf("%-86s %7.2f us  %6.1f GB/s  (%4.1f %% of 8 TB/s)\n", c.name, best * 1e3, moved / (best * 1e-3) / 1e9, moved / (best * 1e-3) / 8e12 * 100);
  }
  return 0;
}
